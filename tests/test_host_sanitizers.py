"""Host-side passes of the sparse block under the sanitizers (CPU only; GPU AddressSanitizer is not available on the pool):
tests/host/row_patterns_harness.cpp compiles prost_amd/csrc/host/{common,linop}.cpp into one translation unit and drives
BuildRowPatterns (the recognition of stencils written out as sparse matrices, threaded) and the multi-core csr2csc."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "row_patterns_harness.cpp")
EXPECT = ["stencil 126600 rows: ok=1 patterns=3 bad=0", "stencil 151368 rows: ok=1 patterns=3 bad=0", "stencil 176364 rows: ok=1 patterns=3 bad=0",
          "unstructured: ok=0", "csr2csc 4498600 entries: equal=1"]


@pytest.mark.parametrize("name,flags", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]), ("tsan", ["-fsanitize=thread"])])
def test_row_patterns_and_csr2csc_under_sanitizers(tmp_path, name, flags):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / ("harness_" + name))
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread"] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "prost_amd", "csrc", "host"),
                                                                       SRC, "-o", exe, "-Wl,--unresolved-symbols=ignore-all"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(SRC))
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.strip().splitlines() == EXPECT, r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
