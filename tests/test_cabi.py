"""C-ABI shape checks that need no GPU: both shared libraries load, export every symbol their
headers declare, and the host-only commands (problem setup, factories, error paths) behave like
the reference's gateway."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import _capi, _hip, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prost_[a-z0-9_]+)\s*\(", text)) - {"prost_interm_cb", "prost_stop_cb"})


def test_kernel_library_exports_every_declared_symbol():
    names = declared_functions("prost_hip.h")
    assert len(names) > 70
    L = _hip.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert L.prost_hip_abi_version() == 10
    assert L.prost_hip_reduce_workspace_bytes() >= 4096


def test_host_library_exports_every_declared_symbol():
    names = declared_functions("prost_c.h")
    L = _capi.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_unknown_command_and_factory_errors():
    with pytest.raises(_capi.ProstError, match="Unknown command 'frobnicate'"):
        _capi.command("frobnicate")
    u, q = prost.variable(4), prost.variable(8)
    prob = prost.min_max_problem([u], [q])
    prob.add_dual_pair(u, q, prost.block.gradient2d(2, 2, 1))
    prob.data["prox_g"] = [["no_such_prox", 0, 4, True, []]]
    with pytest.raises(_capi.ProstError, match="Creating prox with ID 'no_such_prox' failed. Reason: Name not registered in ProxFactory"):
        prost.problem_info(prob)
    prob.data["prox_g"] = [["zero", 0, 4, True, []]]
    prob.data["linop"] = [["no_such_block", 0, 0, []]]
    with pytest.raises(_capi.ProstError, match="Name not registered in BlockFactory"):
        prost.problem_info(prob)
    prob.data["linop"] = [["gradient2d", 0, 0]]
    with pytest.raises(_capi.ProstError, match="Invalid block description"):
        prost.problem_info(prob)


def test_problem_setup_errors_match_reference_messages():
    u, q = prost.variable(6), prost.variable(12)
    prob = prost.min_max_problem([u], [q])
    prob.add_dual_pair(u, q, prost.block.gradient2d(3, 2, 1))
    prob.data["prox_g"] = [["zero", 0, 4, True, []], ["zero", 2, 4, True, []]]
    prob.data["prox_fstar"] = [["zero", 0, 12, True, []]]
    with pytest.raises(_capi.ProstError, match=r"prox_g \(CheckDomainProx\): Prox operators are overlapping: \[0, 3\] and \[2, 5\]"):
        prost.problem_info(prob)
    prob.data["prox_g"] = [["zero", 0, 9, True, []]]
    with pytest.raises(_capi.ProstError, match="Last prox operator ends after the domain"):
        prost.problem_info(prob)
    prob.data["prox_g"] = [["zero", 0, 6, True, []]]
    prob.data["prox_f"] = [["zero", 0, 12, True, []]]
    with pytest.raises(_capi.ProstError, match="Proximal operator for f AND fstar specified"):
        prost.problem_info(prob)
    prob.data["prox_f"] = []
    prob.data["linop"].append(["zero", 3, 2, [4, 2]])
    with pytest.raises(_capi.ProstError, match="Blocks are overlapping inside the linear operator"):
        prost.problem_info(prob)


@pytest.mark.parametrize("precision", ["single", "double"])
def test_problem_info_matches_oracle_setup(precision):
    """zero-prox filling, Pock-Chambolle preconditioners with the carried value, averaging over
    non-diagstep groups (problem.cu:196-323, :503-536) -- host code, compared with the oracle"""
    import scipy.sparse as sp
    prost.set_precision(precision)
    try:
        dt = np.float32 if precision == "single" else np.float64
        rng = np.random.default_rng(0)
        nx, ny = 6, 5
        n = nx * ny
        u, w = prost.variable(n), prost.variable(7)
        q, r = prost.variable(2 * n), prost.variable(9)
        K = sp.random(9, 7, density=0.3, random_state=1, format="csc")
        K[3, :] = 0            # an all-zero row: inherits the previous value (problem.cu:267-273)
        prob = prost.min_max_problem([u, w], [q, r])
        prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, 1))
        prob.add_dual_pair(w, r, prost.block.sparse(K))
        prob.add_dual_pair(u, r, prost.block.diags(9, n, [0.5, -2.0], [0, 3]))
        prob.add_function(u, prost.function.sum_1d("square", 1, rng.random(n), 10))
        prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
        for alpha in (1.0, 0.5):
            prob.set_scaling_alpha(alpha)
            info = prost.problem_info(prob)
            P = oracle.Problem(prob.data, prob.nrows, prob.ncols, dt)
            P.initialize()
            sl, sr = P.scaling()
            assert np.array_equal(np.asarray(info["scaling_left"]), sl)
            assert np.array_equal(np.asarray(info["scaling_right"]), sr)
        pg = np.asarray(info["prox_g"]).reshape(-1, 3) if np.ndim(info["prox_g"]) == 1 else np.asarray(info["prox_g"]).T
        assert sorted(pg[:, 0].tolist()) == [0, n] and sorted(pg[:, 1].tolist()) == [7, n]     # zero prox appended for w
        assert info["nrows"] == 2 * n + 9 and info["ncols"] == n + 7
    finally:
        prost.set_precision("double")


def test_problem_setup_on_several_host_threads_matches_oracle():
    """The preconditioner sweeps run on up to 8 host threads above ~10^6 entries (ParallelFor): a 1300 x 1100 image
    (1.43 M pixels, 2.86 M rows) plus a sparse block with long runs of empty rows / columns, so that the value the
    sequential sweep carries over empty rows (problem.cu:262-287) has to cross sub-range boundaries."""
    import scipy.sparse as sp
    prost.set_precision("single")
    try:
        nx, ny = 1300, 1100
        n = nx * ny
        kr, kc = 2_400_000, 1_300_000
        rows = np.array([5, 17, 1_200_000, 1_200_001, 2_399_990]); cols = np.array([3, 600_000, 600_001, 1_299_999, 7])
        K = sp.csc_matrix((np.array([0.5, -2.0, 3.0, 0.25, 1.5]), (rows, cols)), shape=(kr, kc))
        u, w = prost.variable(n), prost.variable(kc)
        q, r = prost.variable(2 * n), prost.variable(kr)
        prob = prost.min_max_problem([u, w], [q, r])
        prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, 1))
        prob.add_dual_pair(w, r, prost.block.sparse(K))
        prob.add_function(u, prost.function.sum_1d("square", 1, 0.5, 10))
        prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
        prob.set_scaling_alpha(1.0)
        info = prost.problem_info(prob)
        P = oracle.Problem(prob.data, prob.nrows, prob.ncols, np.float32)
        P.initialize()
        sl, sr = P.scaling()
        assert np.array_equal(np.asarray(info["scaling_left"]), sl)
        assert np.array_equal(np.asarray(info["scaling_right"]), sr)
        assert len(np.unique(sl)) >= 4 and len(np.unique(sr)) >= 4          # carried values differ along the sweep
    finally:
        prost.set_precision("double")


def test_precision_switch_and_default():
    assert prost.get_precision() == "double"          # reference default: typedef double real (config.hpp:7)
    prost.set_precision("single")
    assert prost.get_precision() == "single"
    prost.set_precision("double")
    with pytest.raises(_capi.ProstError):
        prost.set_precision("half")


def test_value_round_trip_through_the_c_tree():
    L = _capi.lib()
    keep = []
    v = _capi.to_value({"a": [1, "two", np.arange(3.0)], "m": np.arange(6.0).reshape(2, 3)}, keep)
    try:
        a = L.prost_value_field(v, b"a")
        assert L.prost_value_kind(a) == _capi.VALUE_CELL and L.prost_value_count(a) == 3
        assert _capi.from_value(a) [1] == "two" and np.array_equal(_capi.from_value(a)[2], np.arange(3.0))
        assert np.array_equal(_capi.from_value(L.prost_value_field(v, b"m")), np.arange(6.0).reshape(2, 3))
        assert not L.prost_value_field(v, b"missing")
    finally:
        L.prost_value_free(v)


def test_product_fails_loudly_without_a_gpu():
    """no CPU fallback: on a machine without an MI355X every compute command raises"""
    if _hip.device_count() > 0:
        pytest.skip("a GPU is present")
    prob, u, q, f = synthetic.rof_problem(8, 8)
    with pytest.raises(_capi.ProstError, match="no MI355X|no CPU fallback|Invalid HIP device"):
        prost.solve(prob, prost.backend.pdhg(), prost.options(max_iters=2, verbose=False))
    with pytest.raises(_capi.ProstError):
        prost.eval_prox(prost.function.sum_1d("abs"), np.ones(4), 1.0, np.ones(4))
    with pytest.raises(_hip.HipError):
        _hip.require_device()


def test_glibc_rand_stream_is_generated_chunk_parallel_and_stays_exact():
    """Problem::normest starts from (T)rand() / (T)RAND_MAX of a fresh process (problem.cu:441-444).  The host library generates
    that stream on several threads by jump-ahead of the additive-feedback recurrence (GlibcRand::fill_unit); it must equal the
    sequential stream -- the oracle's GlibcRand, pinned against the real std::rand() by tests/test_oracle_pinning.py -- for
    lengths below, at and across the chunk boundaries, and after a prefix drawn one by one."""
    import oracle
    from prost_amd import _capi
    prost.set_precision("single")
    try:
        for n, skip in ((1, 0), (30, 0), (31, 5), (1000, 0), ((1 << 21) + 7, 0), (5 * (1 << 20) + 3, 17)):
            got = np.asarray(_capi.command("glibc_rand_unit", [n, skip], nlhs=1)[0]).reshape(-1)
            ref = oracle.glibc_rand(1, n + skip)[skip:].astype(np.float32) / np.float32(2147483647)
            assert got.shape == (n,) and np.array_equal(got.astype(np.float32), ref), (n, skip, int((got.astype(np.float32) != ref).sum()))
        # consecutive calls continue the stream (Problem::normest draws the start vector piece by piece into pinned staging buffers)
        for n, skip, piece in ((100, 0, 7), (5 * (1 << 20) + 3, 3, (1 << 21) + 5), (3 * (1 << 20), 0, 1 << 20), (40, 2, 31)):
            got = np.asarray(_capi.command("glibc_rand_unit", [n, skip, piece], nlhs=1)[0]).reshape(-1)
            ref = oracle.glibc_rand(1, n + skip)[skip:].astype(np.float32) / np.float32(2147483647)
            assert np.array_equal(got.astype(np.float32), ref), (n, skip, piece, int((got.astype(np.float32) != ref).sum()))
        prost.set_precision("double")
        n = 3 * (1 << 20) + 11
        got = np.asarray(_capi.command("glibc_rand_unit", [n], nlhs=1)[0]).reshape(-1)
        assert np.array_equal(got, oracle.glibc_rand(1, n).astype(np.float64) / 2147483647.0)
    finally:
        prost.set_precision("double")


@pytest.mark.parametrize("precision", ["single", "double"])
def test_constant_preconditioners_of_a_single_stencil_block_match_the_oracle_sweep(precision):
    """A problem whose operator is ONE gradient block takes the constant-preconditioner path (Block::uniform_sums,
    Prox::average_uniform: no sweep, no per-entry averaging); the vectors it stands for must be what the oracle's sweep and
    group averaging (problem.cu:262-287, :503-536) produce: gray / RGB gradient2d with sum_norm2 over 2 / 6 components,
    gradient3d with 3, every alpha, and a conjugated (Moreau-wrapped) f"""
    from prost_amd import synthetic
    prost.set_precision(precision)
    try:
        dt = np.float32 if precision == "single" else np.float64
        cases = []
        for L in (1, 3):
            cases.append(synthetic.rof_problem(7, 5, L, seed=3)[0])
        cases.append(synthetic.tv3d_problem(6, 5, 4, seed=3)[0])
        u, q = prost.variable(7 * 5), prost.variable(2 * 7 * 5)
        p = prost.min_max_problem([u], [q])
        p.add_dual_pair(u, q, prost.block.gradient2d(7, 5, 1))
        p.add_function(u, prost.function.sum_1d("square", 1, np.linspace(0, 1, 35), 10))
        p.add_function(q, prost.function.conjugate(prost.function.sum_norm2(2, False, "abs", 1, 0, 1)))
        cases.append(p)
        for prob in cases:
            for alpha in (1.0, 0.5, 2.0, 0.0):
                prob.set_scaling_alpha(alpha)
                info = prost.problem_info(prob)
                P = oracle.Problem(prob.data, prob.nrows, prob.ncols, dt)
                P.initialize()
                sl, sr = P.scaling()
                assert np.array_equal(np.asarray(info["scaling_left"]), sl), alpha
                assert np.array_equal(np.asarray(info["scaling_right"]), sr), alpha
                assert np.unique(sl).size == 1 and np.unique(sr).size == 1
    finally:
        prost.set_precision("double")


def test_support_predicates_and_launch_geometry_of_the_double_iteration_kernels():
    """Host-only entry points of the kernel library (no launch): which problem shapes the two-iterations-per-launch kernels take,
    and the column-chunk lengths their launchers pick -- rounds of workgroups x column steps minimal (3-D: 2048 x 2048 x 64 ->
    17 strips x 5 plane groups x 3 chunks = 255 workgroups on 256 compute units, 683 columns each), 24 columns at most for the
    multi-channel kernel, one partial per workgroup within the reduction workspace for the residual variants."""
    from prost_amd import _hip as hip
    L = hip.lib()

    def desc(is3d, nx, ny, nl, g_fn="square", f_fn="ind_leq0"):
        d = hip.FusedDesc(); d.is3d = is3d; d.nx, d.ny, d.L = nx, ny, nl
        d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID[f_fn]
        for i, (g, f) in enumerate(zip([1, 0.3, 10, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0])):
            d.g_coeff_val[i] = g; d.f_coeff_val[i] = f
        d.T_val, d.S_val = (1 / 6.0, 0.5) if is3d else (0.25, 0.5)
        return d

    c3 = desc(1, 2048, 2048, 64)
    for dt in (0, 1):
        assert L.prost_hip_fused_iteration3d_x2_supported(C.byref(c3), dt) == 1
        assert L.prost_hip_fused_iteration3d_x2_supported(C.byref(desc(1, 64, 63, 8)), dt) == 1          # any height
        assert L.prost_hip_fused_iteration3d_x2_supported(C.byref(desc(1, 64, 64, 8, g_fn="huber")), dt) == 0
        assert L.prost_hip_fused_iteration3d_x2_supported(C.byref(desc(0, 64, 64, 3)), dt) == 0          # not gradient3d
        assert L.prost_hip_fused_iteration_mc_x2_supported(C.byref(desc(0, 64, 63, 3)), dt) == 1
        assert L.prost_hip_fused_iteration_mc_x2_supported(C.byref(desc(0, 64, 64, 1)), dt) == 0          # gray: prost_hip_fused_iteration2
        assert L.prost_hip_fused_iteration_mc_x2_supported(C.byref(desc(0, 64, 64, 5)), dt) == 0
        assert L.prost_hip_fused_iteration_mc_x2_supported(C.byref(desc(0, 64, 64, 3, f_fn="huber")), dt) == 0
    # without a device the launcher assumes the 256 compute units of an MI355X
    assert L.prost_hip_fused_iteration3d_x2_chunk_cols(C.byref(c3), 0, 0) == 683
    assert L.prost_hip_fused_iteration3d_x2_chunk_cols(C.byref(c3), 0, 1) == 683
    assert L.prost_hip_fused_iteration3d_x2_chunk_cols(C.byref(desc(1, 64, 64, 8, g_fn="huber")), 0, 0) == 0
    assert 20 <= L.prost_hip_fused_iteration_mc_x2_chunk_cols(C.byref(desc(0, 4096, 4096, 3)), 0, 0) <= 24
    assert L.prost_hip_fused_iteration_mc_x2_chunk_cols(C.byref(desc(0, 512, 512, 3)), 0, 0) == 2
    assert 1 <= L.prost_hip_fused_iteration_mc_x2_chunk_cols(C.byref(desc(0, 700, 464, 3)), 0, 1) <= 24
    # profitable: tiny (launch-bound) and large (throughput-bound) images, not the latency-bound middle
    assert [L.prost_hip_fused_iteration_mc_x2_profitable(C.byref(desc(0, n, n, 3)), 0) for n in (256, 512, 1024, 4096)] == [1, 0, 1, 1]
    # residual variants: one partial per workgroup must fit the reduction workspace (4096 workgroups).  4096^2 in fp64 has 34
    # strips of 124 rows, 24-column chunks would make 5814 workgroups: the chunk grows to the shortest length that fits
    for n, dt in ((4096, 1), (4096, 0), (8192, 1), (16384, 1)):
        rows = 62 * (4 if dt == 0 else 2)
        c = L.prost_hip_fused_iteration_mc_x2_chunk_cols(C.byref(desc(0, n, n, 3)), dt, 1)
        assert c >= 1 and -(-n // rows) * -(-n // c) <= 4096, (n, dt, c)
    assert L.prost_hip_fused_iteration_mc_x2_chunk_cols(C.byref(desc(0, 4096, 4096, 3)), 1, 1) == 35
    # gradient3d two-pass fallback: (row blocks of 256) x planes partials per residual launch even with ONE column chunk -- volumes
    # beyond the workspace are not taken by the fused path at all (they run the generic kernels) instead of looping for a chunk
    # length that does not exist
    assert L.prost_hip_fused_supported(C.byref(desc(1, 64, 4096, 512)), 0) == 1
    assert L.prost_hip_fused_supported(C.byref(desc(1, 64, 4096, 513)), 0) == 0
    assert L.prost_hip_fused_supported(C.byref(desc(1, 64, 4096, 2049)), 0) == 0
    assert L.prost_hip_fused_iteration3d_x2_supported(C.byref(desc(1, 64, 4096, 2049)), 0) == 0


def test_host_csr2csc_parallel_and_sequential_forms_agree_with_scipy():
    """prost::csr2csc (host library; reference src/common.cu:55-82) straight through its C++ symbol: from 4 M entries on it counts and
    scatters row sub-ranges on all host cores where they touch narrow column ranges (banded matrices) and runs the reference's
    sequential counting sort otherwise -- arrays identical to scipy's sorted CSC in both forms.  (Matrices built from COO triplets:
    scipy.sparse.random with a legacy seed permutes rows x columns positions, 13 TiB at these shapes.)"""
    import ctypes as C
    import scipy.sparse as sp
    from reference_matrices import spmat_gradient2d
    # (loaded with local symbol scope: the real reference build of tests/test_oracle_pinning.py defines the same C++ names)
    lib = C.CDLL(os.path.join(ROOT, "prost_amd", "lib", "libprost.so"))
    f = getattr(lib, "_ZN5prost7csr2cscIdEEviiiPKT_PKiS5_PS1_PiS7_")
    f.restype = None
    rng = np.random.default_rng(1)

    def check(A):
        A = sp.csr_matrix(A); A.sort_indices()
        n, m = A.shape
        val, ind, ptr = (np.ascontiguousarray(A.data, dtype=np.float64), np.ascontiguousarray(A.indices, dtype=np.int32), np.ascontiguousarray(A.indptr, dtype=np.int32))
        oval, oind, optr = np.zeros(A.nnz), np.zeros(A.nnz, dtype=np.int32), np.zeros(m + 1, dtype=np.int32)
        f(C.c_int(n), C.c_int(m), C.c_int(A.nnz), *[a.ctypes.data_as(C.c_void_p) for a in (val, ind, ptr, oval, oind, optr)])
        B = sp.csc_matrix(A); B.sort_indices()
        assert np.array_equal(optr, B.indptr) and np.array_equal(oind, B.indices) and np.array_equal(oval, B.data)

    band = sp.csr_matrix(spmat_gradient2d(1100, 1100, 1))
    band.data = band.data * rng.uniform(0.5, 1.5, band.nnz)
    check(band); check(band.T)                                                   # parallel form (narrow column ranges)
    nz = 4_500_000
    check(sp.coo_matrix((rng.uniform(-1, 1, nz), (rng.integers(0, 1_500_000, nz), rng.integers(0, 1_200_000, nz))), shape=(1_500_000, 1_200_000)))   # sequential form
    check(sp.coo_matrix((rng.uniform(-1, 1, nz), (rng.integers(0, 30, nz) * 40000 + rng.integers(0, 3, nz), rng.integers(0, 900_000, nz))), shape=(1_300_000, 900_000)))  # empty rows, dense stripes
    check(sp.random(300, 200, density=0.05, random_state=1))                     # small: sequential
