"""The enqueue order of a column-sharded slab with the residual-driven rule on the device, on a MOCKED transport (CPU only).

Round 6 lets slabs on the RCCL transport run device-resident batches (BackendPDHG::SetExchangeHook): the halo exchange is enqueued from
inside the batch, so a whole solver_iterate_sharded call is one sequence of stream-ordered work with ONE host wait per batch.  No
multi-GPU node was available to any round so far; the first real 8-GPU run should exercise a schedule that has been checked.
tests/host/slab_schedule_harness.cpp compiles the solver's host sources against a recording mock of the kernel C ABI and prints
what a rank enqueues; this test parses that log:

  * per residual iteration:  iteration kernel (+ partial sums) -> all-reduce of the four sums -> rule kernel -> next launch
  * the halo exchange every halo - 2 iterations, counted over ALL iterations (also the first two, which run the host loop), never
    inside a two-iteration launch
  * one host wait per batch on the device-side transport; on the host-callback transport (gloo) the rule stays on the host
    (a wait per residual iteration) and the exchanges still come every halo - 2 iterations
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "slab_schedule_harness.cpp")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("slab") / "slab_schedule_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "prost_amd", "csrc", "host"), SRC, "-o", exe,
           "-Wl,--unresolved-symbols=ignore-all"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(SRC))
    assert b.returncode == 0, b.stderr[-3000:]
    return exe


def _run(exe, iters, residual_iter, halo, pairs=1, host_transport=False):
    env = dict(os.environ)
    env.pop("MOCK_HOST_TRANSPORT", None)
    if host_transport:
        env["MOCK_HOST_TRANSPORT"] = "1"
    r = subprocess.run([exe, str(iters), str(residual_iter), str(halo), str(pairs)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = r.stdout.strip().splitlines()
    head = re.match(r"path (\S+) device_rules (\d)", lines[0])
    tail = re.match(r"iterations (\d+) since_exchange (\d+) exchanges (\d+)", lines[1])
    return head.group(1), int(head.group(2)), int(tail.group(1)), int(tail.group(2)), int(tail.group(3)), lines[2:]


def _launch_iterations(line):
    """iterations a launch line covers: ([k ...], carries_sums) or None for a non-launch line"""
    m = re.match(r"iteration x(\d)(?: k=([\d,]+))?", line)
    if not m:
        return None
    ks = [int(v) for v in m.group(2).split(",")] if m.group(2) else None
    return int(m.group(1)), ks, "+sums" in line


@pytest.mark.parametrize("iters,residual_iter,halo,pairs", [(26, 3, 8, 1), (40, 1, 8, 1), (61, 10, 5, 1), (33, 4, 8, 0), (250, 7, 10, 1), (12, 5, 3, 1)])
def test_device_resident_slab_batches_enqueue_exchange_allreduce_rule_and_launches_in_order(harness, iters, residual_iter, halo, pairs):
    path, dev, done, since, exchanges, log = _run(harness, iters, residual_iter, halo, pairs)
    assert path == "pdhg:fused-grad2d" and dev == 1 and done == iters
    period = halo - 2
    count, since_ex, seen_begin, waits_after_begin, batches = 0, 0, False, 0, 0
    i = 0
    while i < len(log):
        line = log[i]
        if line.startswith("rule_begin"):
            seen_begin = True; batches += 1
        elif line == "HALO EXCHANGE":
            assert since_ex == period, (i, since_ex)                    # exactly when it is due, over ALL iterations since the last one
            since_ex = 0
        elif line.startswith("HOST WAIT"):
            if seen_begin:
                waits_after_begin += 1
        else:
            launch = _launch_iterations(line)
            if launch is not None:
                width, ks, sums = launch
                if ks is not None:
                    assert ks == list(range(count, count + width)), (line, count)      # consecutive, nothing skipped or repeated
                    if sums:
                        # the sums belong to the launch's LAST iteration, a residual iteration; behind the launch: all-reduce, then the rule kernel
                        assert ks[-1] % residual_iter == 0, line
                        assert log[i + 1] == "all-reduce of 4 sums" and log[i + 2] == "rule kernel k=%d" % ks[-1], log[i:i + 3]
                        assert "+rule" not in line                                      # with a communicator the kernel itself never applies the rule
                    else:
                        assert not any(k % residual_iter == 0 for k in ks), line          # no residual iteration without its sums
                count += width; since_ex += width
                assert since_ex <= period, (line, since_ex)                               # no launch across an exchange
        i += 1
    assert count == iters and since_ex == since
    assert exchanges == (iters - 1) // period
    # one host wait per batch of up to 240 iterations (the first two iterations run the host loop: their waits come before the first batch)
    assert batches == max(1, -(-(iters - 2) // 240)) and waits_after_begin == batches, (batches, waits_after_begin)


def test_host_callback_transport_keeps_the_rule_on_the_host(harness):
    """gloo / host-callback transport: the exchange needs the host, so slabs keep the host loop (a wait per residual iteration)"""
    path, dev, done, since, exchanges, log = _run(harness, 30, 3, 8, 1, host_transport=True)
    assert path == "pdhg:fused-grad2d" and dev == 0 and done == 30
    assert not any(l.startswith("rule_begin") or l.startswith("rule kernel") for l in log)
    assert exchanges == 29 // 6 and sum(1 for l in log if l == "HALO EXCHANGE") == exchanges
    residual_launches = sum(1 for l in log if "+sums" in l)
    assert residual_launches == 10 and sum(1 for l in log if l.startswith("HOST WAIT")) >= residual_launches
    count = since_ex = 0
    for l in log:
        if l == "HALO EXCHANGE":
            assert since_ex == 6
            since_ex = 0
        else:
            launch = _launch_iterations(l)
            if launch:
                count += launch[0]; since_ex += launch[0]
                assert since_ex <= 6
    assert count == 30
