"""CPU tests of bench.py's bookkeeping (no GPU): the byte model per kernel name.

Every name `BackendPDHG::KernelTimes` can emit (prost_amd/csrc/host/backend_pdhg.cpp, the `names[]` table) must map to the
compulsory values per pixel / voxel DESIGN.md section 3 states for that launch.  Round 4 matched kernel kinds by the substring
"dual", which also sits in "resi-dual-s": every `+residuals` launch was priced as a dual pass.
"""
import os
import re

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, volume) -> values per pixel / voxel and launch
EXPECTED = {
    ("fused_primal2d_kernel", False): 5, ("fused_dual2d_kernel", False): 6,
    ("fused_primal3d_kernel", True): 6, ("fused_dual3d_kernel", True): 8,
    ("fused_iter2d_kernel", False): 7, ("fused_iter2d_mc_kernel", False): 7, ("fused_iter3d_kernel", True): 9,
    ("fused_iter2d_kernel+residuals", False): 9, ("fused_iter3d_kernel+residuals", True): 13,
    ("fused_iter2d_x2_kernel", False): 7, ("fused_iter2d_mc_x2_kernel", False): 7, ("fused_iter3d_x2_kernel", True): 9,
    ("fused_iter2d_x2_kernel+mid", False): 10,
    ("fused_iter2d_x2_kernel+residuals", False): 7, ("fused_iter2d_mc_x2_kernel+residuals", False): 7,
    ("fused_iter3d_x2_kernel+residuals", True): 9,
    ("fused_iter2d_x2_kernel+mid+residuals", False): 10,
    # K iterations per launch (tolerance-class arithmetic): the same seven values whatever K
    ("fused_iter2d_xk_kernel<2>", False): 7, ("fused_iter2d_xk_kernel<3>", False): 7, ("fused_iter2d_xk_kernel<4>", False): 7,
    ("fused_iter2d_xk_kernel<2>+residuals", False): 7, ("fused_iter2d_xk_kernel<3>+residuals", False): 7, ("fused_iter2d_xk_kernel<4>+residuals", False): 7,
}


@pytest.mark.parametrize("key", sorted(EXPECTED))
def test_compulsory_floats_of_every_kernel_name(key):
    name, volume = key
    assert bench.compulsory_floats(name, volume) == EXPECTED[key]


def test_every_name_the_backend_can_emit_is_priced():
    """the names[] table of KernelTimes, read from the source: nothing the backend reports is left without a byte figure, and no
    `+residuals` name is taken for a dual pass"""
    src = open(os.path.join(ROOT, "prost_amd", "csrc", "host", "backend_pdhg.cpp")).read()
    block = src[src.index("const char* names[kKernelKinds]"):]
    block = block[:block.index("};")]
    names = set(re.findall(r'"(fused_[a-z0-9_+<>]+)"', block))
    assert len(names) >= 21, names
    for name in names:
        volume = "3d" in name
        assert (name, volume) in EXPECTED, "tests/test_bench_harness.py does not know %s" % name
        assert bench.compulsory_floats(name, volume) == EXPECTED[(name, volume)], name
        family, suffix = bench.kernel_kind(name)
        assert family in ("primal", "dual", "iter", "iter_x2", "iter_xk")
        if "residuals" in name:
            assert family not in ("primal", "dual"), name


def test_no_rate_above_the_hbm_peak_unless_its_name_says_equivalent():
    """round-5 review: a top-level `achieved_hbm_GBps` of 13 218 read as impossible.  The two-pass-equivalent figures carry 'equiv' in
    their names; everything else called GB/s, and every roofline fraction, is checked against the peak before the line is printed"""
    ok = {"n_gpus": 1, "value": 19000.0, "two_pass_equiv_GBps": 14026.0, "roofline": {"achieved": 4600.0, "achieved_hbm_traffic": 5500.0, "frac": 0.575, "frac_hbm_traffic": 0.69,
                                                                                "algorithmic_equiv_GBps": 14400.0, "algorithmic_equiv_frac": 1.8, "peak": 8000.0},
          "fmad": {"two_pass_equiv_GBps": 23000.0}, "roofline_fmad": {"achieved": 4500.0, "frac": 0.56}}
    assert bench.validate_line(ok) is ok
    for bad in ({"achieved_hbm_GBps": 13218.0}, {"roofline": {"achieved": 8100.0}}, {"roofline": {"frac": 1.01}}, {"roofline_fmad": {"achieved_hbm_traffic": 9000.0}},
                {"fmad": {"some_GBps": 9000.0}}):
        with pytest.raises(ValueError):
            bench.validate_line(dict(ok, **bad))
    assert bench.validate_line({"n_gpus": 8, "some_GBps": 40000.0})            # whole-job rates scale with the GPUs
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"achieved_hbm_GBps"' not in src and "validate_line(out)" in src


def test_unknown_names_are_not_priced():
    for name in ("cg_step_xr2_kernel", "fused_dual2d_kernel+residuals", "residuals", "fused_iter2d_kernel+mid", "something_dual", "fused_iter2d_xk_kernel<4>+mid",
                 "fused_iter2d_xk_kernel<x>"):
        assert bench.compulsory_floats(name, False) is None


def test_c4_byte_model():
    """C4 (ADMM): every kernel name BackendADMM::KernelTimes emits is priced, and the whole-iteration figure is the sum over the
    stages outside the solve and the CG rounds of the path that ran"""
    src = open(os.path.join(ROOT, "prost_amd", "csrc", "host", "backend_admm.cpp")).read()
    for table in ("names4", "names2"):
        block = src[src.index("static const char* const %s[4]" % table):]
        block = block[:block.index("};")]
        for name in re.findall(r'"([a-zA-Z0-9_<>]+)"', block):
            assert bench.c4_kernel_bytes(name, 1024 * 1024) > 0, name
    px = 1024 * 1024
    outer = sum(v * 4 + i * 4 for v, i in bench.C4_OUTER_VALUES.values()) * px
    assert bench.c4_iteration_bytes("admm:pixel-op", 10, px) == outer + (44 * 10 - 4) * 4 * px
    assert bench.c4_iteration_bytes("admm:fused-op", 10, px) == outer + 10 * (55 * 4 + 7 * 4) * px
    assert bench.c4_iteration_bytes("admm:generic", 10, px) is None
    assert bench.c4_iteration_bytes("admm:pixel-op", 10, px, 8) > bench.c4_iteration_bytes("admm:pixel-op", 10, px, 4)
