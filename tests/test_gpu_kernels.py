"""GPU parity tests of the kernel C ABI (include/prost_hip.h) against the CPU oracle.

Bar: bit-exact for every kernel whose arithmetic is +,-,*,/,sqrt (both sides are compiled without
FMA contraction and HIP's fp32 divide/sqrt are correctly rounded); <= 4 ulp where libm
transcendentals (pow/sin/acos/cos) are involved (lq, epi_quad); reductions (different summation
order) rel 1e-6.
"""
import ctypes as C

import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64]


_KEEP = []


def dev(hip, a):
    """host -> device; the buffer is kept alive until the end of the test (kernels are async)."""
    d = hip.DeviceArray.from_host(a)
    _KEEP.append(d)
    return d


@pytest.fixture(autouse=True)
def _release_device_buffers(request):
    yield
    if "hip" in request.fixturenames:
        request.getfixturevalue("hip").sync()
    for d in _KEEP:
        d.free()
    del _KEEP[:]


def ulp_diff(a, b):
    a = np.asarray(a); b = np.asarray(b)
    eps = np.finfo(a.dtype).eps
    scale = np.maximum(np.maximum(np.abs(a), np.abs(b)), np.finfo(a.dtype).tiny)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / (scale.astype(np.float64) * eps)
    d[np.isnan(a) & np.isnan(b)] = 0
    return np.nanmax(d) if d.size else 0.0


GRAD_SHAPES = [(7, 5, 1, False), (7, 5, 3, False), (7, 5, 3, True), (1, 9, 1, False), (9, 1, 1, False),
               (33, 130, 2, False), (307, 229, 8, False), (64, 300, 2, True),
               # label-first runs of ny L values: the 16-bytes-per-lane kernels of round 6 (ny L a multiple of 4 / 2), more than one strip,
               # L larger than a lane's 4 values, one column, one row
               (9, 8, 3, True), (70, 1030, 4, True), (5, 12, 7, True), (1, 16, 1, True), (6, 1, 4, True), (3, 2, 2, True)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", GRAD_SHAPES)
@pytest.mark.parametrize("d3", [False, True])
def test_gradient_fwd_adj(hip, dtype, shape, d3):
    nx, ny, L, lf = shape
    rng = np.random.default_rng(1)
    n = nx * ny * L
    k = 3 if d3 else 2
    x = rng.standard_normal(n).astype(dtype)
    y = rng.standard_normal(k * n).astype(dtype)
    base_r = rng.standard_normal(k * n).astype(dtype)
    base_c = rng.standard_normal(n).astype(dtype)
    og = oracle.grad3d if d3 else oracle.grad2d
    name = "grad3d" if d3 else "grad2d"
    for acc in (0, 1):
        ref_f = og(x, nx, ny, L, lf, adjoint=False, acc=base_r.copy() if acc else None)
        ref_a = og(y, nx, ny, L, lf, adjoint=True, acc=base_c.copy() if acc else None)
        r = dev(hip, base_r); c = dev(hip, base_c)
        hip.check(hip.fn(name + "_fwd", dtype)(r.ptr, dev(hip, x).ptr, hip.sz(nx), hip.sz(ny), hip.sz(L), int(lf), acc, None))
        hip.check(hip.fn(name + "_adj", dtype)(c.ptr, dev(hip, y).ptr, hip.sz(nx), hip.sz(ny), hip.sz(L), int(lf), acc, None))
        assert np.array_equal(r.to_host(), ref_f)
        assert np.array_equal(c.to_host(), ref_a)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(5912, 1131), (300, 2000), (64, 64)])
def test_diags(hip, dtype, shape):
    nrows, ncols = shape
    rng = np.random.default_rng(2)
    nd = 29
    perm = rng.permutation(nrows + ncols - 2)[:nd]
    ofs, fac = oracle.diags_sort(perm - nrows + 1, rng.random(nd), dtype)
    x = rng.standard_normal(ncols).astype(dtype)
    y = rng.standard_normal(nrows).astype(dtype)
    d_ofs, d_fac = dev(hip, ofs), dev(hip, fac)
    r = hip.DeviceArray.zeros(nrows, dtype)
    hip.check(hip.fn("diags_fwd", dtype)(r.ptr, dev(hip, x).ptr, hip.sz(nrows), hip.sz(ncols), hip.sz(nd), d_ofs.ptr, d_fac.ptr, None))
    assert np.array_equal(r.to_host(), oracle.diags(x, nrows, ncols, ofs, fac))
    for quirk in (0, 1):
        c = hip.DeviceArray.zeros(ncols, dtype)
        hip.check(hip.fn("diags_adj", dtype)(c.ptr, dev(hip, y).ptr, hip.sz(nrows), hip.sz(ncols), hip.sz(nd), d_ofs.ptr, d_fac.ptr, quirk, None))
        assert np.array_equal(c.to_host(), oracle.diags(y, nrows, ncols, ofs, fac, adjoint=True, ref_grid_quirk=bool(quirk)))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("density", [0.01, 0.2])
def test_csr_spmv(hip, dtype, density):
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    A = sp.random(700, 900, density=density, format="csr", random_state=4, dtype=np.float64).astype(dtype)
    A.sort_indices()
    x = rng.standard_normal(900).astype(dtype)
    base = rng.standard_normal(700).astype(dtype)
    ref = oracle.csr_spmv_acc(base.copy(), x, A.data, A.indptr, A.indices)
    r = dev(hip, base)
    hip.check(hip.fn("csr_spmv_acc", dtype)(r.ptr, dev(hip, x).ptr, hip.sz(700), hip.sz(A.nnz), dev(hip, A.data).ptr,
                                            dev(hip, A.indptr.astype(np.int32)).ptr, dev(hip, A.indices.astype(np.int32)).ptr, None))
    got = r.to_host()
    if A.nnz / 700 <= 6:
        assert np.array_equal(got, ref)           # row-per-lane path sums in index order
    else:
        tol = 1e-5 if dtype == np.float32 else 1e-13
        assert np.allclose(got, ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("acc", [0, 1])
def test_pattern_spmv_with_anchored_tables_at_the_c_abi(hip, dtype, acc):
    """prost_hip_pattern_spmv_anchored (round 5): a matrix between two geometries -- here the FULL 2-D convolution of a 41 x 23 image with a
    3 x 4 kernel, rows padded with a few empty ones -- as one 16-bit pattern number + one int32 anchor (the row's first column) per row and a
    table of (column - anchor, value) sequences built HERE with numpy; the product equals the row-by-row CSR sum of the oracle bit for bit,
    accumulating and not, for an output that is 16-byte aligned (4 / 2 rows per lane) and one that is not (one row per lane)."""
    import ctypes as C
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    ny, nx, kernel = 41, 23, rng.uniform(-1, 1, (3, 4)).astype(dtype)
    kernel[1, 2] = 0
    ky, kx = kernel.shape
    ny2, nx2 = ny + ky - 1, nx + kx - 1
    yy, xx = np.mgrid[0:ny, 0:nx]
    src = (yy + xx * ny).reshape(-1)
    rows, cols, vals = [], [], []
    for j in range(kx):
        for i in range(ky):
            if kernel[i, j] != 0:
                rows.append(((yy + i) + (xx + j) * ny2).reshape(-1)); cols.append(src); vals.append(np.full(src.size, kernel[i, j]))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ny2 * nx2 + 5, ny * nx)).astype(dtype)   # 5 empty rows at the end
    A.sort_indices()
    m, n = A.shape
    ptr, ind, val = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data
    # the table: patterns numbered in order of first appearance
    table, ids, anchor = {}, np.zeros((m + 3) // 4 * 4, np.uint16), np.zeros((m + 3) // 4 * 4, np.int32)
    pptr, rel, pval = [0], [], []
    for r in range(m):
        b, e = ptr[r], ptr[r + 1]
        a0 = int(ind[b]) if e > b else (int(anchor[r - 1]) + 1 if r else 0)
        key = (tuple((ind[b:e] - a0).tolist()), val[b:e].tobytes())
        if key not in table:
            table[key] = len(table)
            rel.extend((ind[b:e] - a0).tolist()); pval.extend(val[b:e].tolist()); pptr.append(len(rel))
        ids[r] = table[key]; anchor[r] = a0
    assert len(table) < 200 and len(rel) < 1024
    x = rng.standard_normal(n).astype(dtype)
    base = rng.standard_normal(m).astype(dtype)
    ref = oracle.csr_spmv_acc(base.copy() if acc else np.zeros(m, dtype), x, A.data, A.indptr, A.indices)
    d_ids, d_anchor = dev(hip, ids), dev(hip, anchor)
    d_pptr, d_rel, d_pval, d_x = dev(hip, np.asarray(pptr, np.int32)), dev(hip, np.asarray(rel, np.int32)), dev(hip, np.asarray(pval, dtype)), dev(hip, x)
    for shift in (0, 1):                       # 1: the output starts one element into its buffer -- not 16-byte aligned
        buf = np.zeros(m + 1, dtype); buf[shift:shift + m] = base
        r = dev(hip, buf)
        out = C.c_void_p(r.ptr.value + shift * np.dtype(dtype).itemsize)
        hip.check(hip.fn("pattern_spmv_anchored", dtype)(out, d_x.ptr, hip.sz(m), d_ids.ptr, d_anchor.ptr, d_pptr.ptr, d_rel.ptr, d_pval.ptr, len(pptr) - 1, len(rel), acc, None))
        got = r.to_host()[shift:shift + m]
        assert np.array_equal(got, ref), (shift, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("op", [0, 1])
@pytest.mark.parametrize("fn", oracle.FUNCTIONS)
def test_prox_elem(hip, dtype, op, fn):
    rng = np.random.default_rng(5)
    count = 1000
    for dim, il, inv, vec in [(1 if op == 0 else 2, False, False, True), (1 if op == 0 else 7, True, True, True),
                              (1 if op == 0 else 3, False, True, False)]:
        arg = rng.uniform(-3, 3, count * dim).astype(dtype)
        if op == 1:   # zero-norm elements
            idx0 = np.arange(dim) if il else np.arange(dim) * count
            arg[idx0] = 0
        td = rng.uniform(0.1, 2, count * dim).astype(dtype)
        alpha = 0.5 if fn == "lq" else 0.7
        if vec:
            coeffs = [rng.uniform(0.5, 2, count), rng.uniform(-1, 1, count), rng.uniform(0.1, 2, count),
                      rng.uniform(-1, 1, count), rng.uniform(0, 1, count), alpha, 1.3]
            coeffs[0][:3] = 0; coeffs[2][3:6] = 0        # a == 0 / c == 0 branch (elem_operation_1d.hpp:42-44)
        else:
            coeffs = [1.0, 0.25, 2.0, 0.0, 0.0, alpha, 1.3]
        ref = oracle.prox_elem(op, fn, arg, td, 0.8, count, dim, il, coeffs, inv)
        ptrs, vals, keep = hip.coeff_args(coeffs, dtype, count)
        res = hip.DeviceArray.zeros(count * dim, dtype)
        hip.check(hip.fn("prox_elem", dtype)(op, hip.FN_ID[fn], res.ptr, dev(hip, arg).ptr, dev(hip, td).ptr, hip.dbl(0.8), int(inv),
                                             hip.sz(count), hip.sz(dim), int(il), ptrs, vals, None))
        got = res.to_host()
        if fn == "lq":
            # Newton iteration on pow(): stops at |delta| <= 1e-5 (f32) / 1e-11 (f64), function_1d.hpp:173-191,
            # so device and host libm agree only to about that accuracy
            tol = 5e-5 if dtype == np.float32 else 1e-10
            assert np.allclose(got, ref, rtol=tol, atol=tol, equal_nan=True), ulp_diff(got, ref)
        else:
            assert np.array_equal(got, ref, equal_nan=True), (fn, op, dim, il, inv, ulp_diff(got, ref))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("op,fn", [(0, "square"), (0, "abs"), (0, "huber"), (1, "ind_leq0"), (1, "abs"), (1, "l0")])
@pytest.mark.parametrize("moreau", [0, 1])
def test_prox_elem_with_argument_source_equals_argument_pass_plus_prox(hip, dtype, op, fn, moreau):
    """prost_hip_prox_elem_arg (argument formed in the load path of the prox kernel) against the separate argument
    pass + prost_hip_prox_elem / _moreau on the same operands: identical bits for both PDHG argument modes, the
    vector kernels (dims 1-4 in registers, 7 in two passes), the interleaved / unaligned element-per-lane kernel."""
    rng = np.random.default_rng(9)
    L_ = hip.lib()
    for count, dim, il, off in ((1000, 1 if op == 0 else 2, False, 0), (1000, 1 if op == 0 else 7, False, 0), (996, 1 if op == 0 else 3, True, 0),
                                (1001, 1 if op == 0 else 4, False, 0), (1000, 1 if op == 0 else 3, False, 1)):
        n = count * dim
        coeffs = [rng.uniform(0.5, 2, count), rng.uniform(-1, 1, count), rng.uniform(0.1, 2, count), 0.1, 0.2, 0.7, 1.3]
        ptrs, vals, keep = hip.coeff_args(coeffs, dtype, count)
        v0, v1, v2, v3 = (hip.DeviceArray.from_host(rng.uniform(lo, hi, n + off).astype(dtype)) for lo, hi in ((-2, 2), (0.2, 1.5), (-1, 1), (-1, 1)))
        td = hip.DeviceArray.from_host(rng.uniform(0.2, 1.5, n + off).astype(dtype))
        esz = np.dtype(dtype).itemsize
        at = lambda d: C.c_void_p(d.ptr.value + off * esz)          # off = 1: pointers that are not 16-byte aligned
        for mode in (1, 2):
            temp = hip.DeviceArray.zeros(n + off, dtype); ref = hip.DeviceArray.zeros(n + off, dtype); got = hip.DeviceArray.zeros(n + off, dtype)
            if mode == 1:
                hip.check(hip.fn("pdhg_primal_arg", dtype)(at(temp), at(v0), at(v1), at(v2), hip.dbl(0.37), hip.sz(n), None))
            else:
                hip.check(hip.fn("pdhg_dual_arg", dtype)(at(temp), at(v0), at(v1), at(v2), at(v3), hip.dbl(0.37), hip.dbl(0.8), hip.sz(n), None))
            plain = hip.fn("prox_elem_moreau" if moreau else "prox_elem", dtype)
            hip.check(plain(op, hip.FN_ID[fn], at(ref), at(temp), at(td), hip.dbl(0.9), 0, hip.sz(count), hip.sz(dim), int(il), ptrs, vals, None))
            spec = hip.ArgSpec(); spec.mode = mode
            for k, d in enumerate((v0, v1, v2, v3)):
                spec.v[k] = d.ptr.value + off * esz
            spec.s[0], spec.s[1] = 0.37, 0.8
            hip.check(hip.fn("prox_elem_arg", dtype)(op, hip.FN_ID[fn], moreau, at(got), C.byref(spec), at(td), hip.dbl(0.9), 0, hip.sz(count), hip.sz(dim), int(il),
                                                     ptrs, vals, None))
            assert np.array_equal(got.to_host()[off:], ref.to_host()[off:], equal_nan=True), (count, dim, il, off, mode)
    # the result must not alias the first operand
    spec = hip.ArgSpec(); spec.mode = 1
    for k in range(3):
        spec.v[k] = v0.ptr.value
    assert hip.fn("prox_elem_arg", dtype)(0, hip.FN_ID["abs"], 0, v0.ptr, C.byref(spec), td.ptr, hip.dbl(1.0), 0, hip.sz(8), hip.sz(1), 0, ptrs, vals, None) != 0
    assert b"alias" in L_.prost_hip_last_error()


def _cgls_reference(A, b, x, sig, tau, shift, tol, maxit, dtype):
    """cgls::Solve (cgls.hpp:222-371) on S^(1/2) A T^(1/2) in numpy: vectors in `dtype`, scalars in double."""
    T = dtype
    sq_s, sq_t = np.sqrt(sig).astype(T), np.sqrt(tau).astype(T)
    ATr = A.T.tocsr()

    def gemv(op, alpha, xv, beta, yv):            # GemvPrecondK (backend_admm.cu:199-272), functor by functor
        if op == "n":
            t = (sq_t * xv).astype(T); yv = ((T(beta) / (T(alpha) * sq_s)) * yv).astype(T)
            yv = (yv + (A @ t).astype(T)).astype(T)
            return (T(alpha) * sq_s * yv).astype(T)
        t = (sq_s * xv).astype(T); yv = ((T(beta) / (T(alpha) * sq_t)) * yv).astype(T)
        yv = (yv + (ATr @ t).astype(T)).astype(T)
        return (T(alpha) * sq_t * yv).astype(T)
    Aop = lambda v: gemv("n", 1, v, 0, np.zeros(A.shape[0], T))
    # the norms of cgls.hpp are order-independent sums in the library (reduce.hpp): math.fsum is the exactly rounded sum of the same terms
    nrm = lambda v: math.sqrt(math.fsum((v.astype(np.float64) ** 2).tolist()))
    x = x.copy(); r = b.copy()
    if nrm(x) > 0:
        r = gemv("n", -1, x, 1, r)
    s = gemv("t", 1, r, -shift, x.copy())
    p = s.copy()
    norms0 = norms = nrm(s); gamma = norms0 ** 2; normx = xmax = nrm(x)
    k = 0
    if norms < np.finfo(T).eps:
        return x, 0
    while k < maxit:
        q = Aop(p)
        dlt = nrm(q) ** 2 + shift * nrm(p) ** 2
        alpha = T(gamma / dlt)
        x = (alpha * p + x).astype(T); r = (T(-gamma / dlt) * q + r).astype(T)
        s = gemv("t", 1, r, -shift, x.copy())
        norms = nrm(s); gamma1 = gamma; gamma = norms ** 2
        p = (T(gamma / gamma1) * p + s).astype(T)
        normx = nrm(x); xmax = max(xmax, normx)
        if norms <= norms0 * tol or normx * tol >= 1:
            break
        k += 1
    return x, k


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("m,n,tol,zero_start", [(300, 200, 1e-3, False), (301, 203, 1e-10, True), (64, 1000, 0.2, False)])
def test_cgls_stages_at_the_c_abi(hip, dtype, m, n, tol, zero_start):
    """The device-resident CGLS driven stage by stage through prost_hip_cgls_stage_* with a CSR operator
    (prost_hip_csr_spmv*), against a numpy restatement of cgls.hpp: same iteration count, same x to round-off;
    rounds queued after the stopping test must leave x untouched."""
    import scipy.sparse as sp
    rng = np.random.default_rng(21)
    L_ = hip.lib()
    A = sp.random(m, n, density=0.05, random_state=4, format="csr", dtype=np.float64).astype(dtype); A.sort_indices()
    At = A.T.tocsr(); At.sort_indices()
    sig = rng.uniform(0.3, 2, m).astype(dtype); tau = rng.uniform(0.3, 2, n).astype(dtype)
    b = rng.standard_normal(m).astype(dtype)
    x0 = np.zeros(n, dtype) if zero_start else rng.standard_normal(n).astype(dtype)
    # CG amplifies round-off differences (here: the association order of the norm reductions) from iteration to iteration:
    # fp32 agrees to 2e-4 for 8 iterations (5e-3 after 12), fp64 to 1e-10 for 12
    maxit = 8 if dtype == np.float32 else 12
    x_ref, k_ref = _cgls_reference(A, b, x0, sig, tau, 1.0, tol, maxit, dtype)
    dev_ = lambda a: hip.DeviceArray.from_host(np.ascontiguousarray(a))
    dA = [dev_(A.data), dev_(A.indptr.astype(np.int32)), dev_(A.indices.astype(np.int32))]
    dAt = [dev_(At.data), dev_(At.indptr.astype(np.int32)), dev_(At.indices.astype(np.int32))]
    vec = {k: hip.DeviceArray.zeros(sz, dtype) for k, sz in (("p", n), ("q", m), ("r", m), ("s", n), ("t", max(m, n)))}
    db, dx, dsig, dtau = dev_(b), dev_(x0), dev_(sig), dev_(tau)
    state = hip.DeviceArray.zeros(L_.prost_hip_cgls_state_bytes() // 8 + 1, np.float64)
    ws = hip.DeviceArray(L_.prost_hip_cgls_workspace_bytes() // 8, np.float64)
    d = hip.CglsDesc()
    d.state, d.workspace, d.b, d.x = state.ptr.value, ws.ptr.value, db.ptr.value, dx.ptr.value
    d.p, d.q, d.r, d.s, d.t = (vec[k].ptr.value for k in "pqrst")
    d.sigma, d.tau, d.m, d.n, d.shift, d.tol, d.host_done, d.epoch = dsig.ptr.value, dtau.ptr.value, m, n, 1.0, tol, None, 1
    stage = lambda which: hip.check(hip.fn("cgls_stage", dtype)(which, C.byref(d), None))
    K = lambda res, rhs, acc: hip.check(hip.fn("csr_spmv_acc" if acc else "csr_spmv", dtype)(res.ptr, rhs.ptr, hip.sz(m), hip.sz(A.nnz), dA[0].ptr, dA[1].ptr, dA[2].ptr, None))
    Kt = lambda res, rhs: hip.check(hip.fn("csr_spmv_acc", dtype)(res.ptr, rhs.ptr, hip.sz(n), hip.sz(A.nnz), dAt[0].ptr, dAt[1].ptr, dAt[2].ptr, None))
    INIT_X, INIT_R, INIT_R2, INIT_S, STEP_Q, STEP_XR, STEP_S, STEP_P = range(8)
    stage(INIT_X); stage(INIT_R); K(vec["r"], vec["t"], True); stage(INIT_R2); Kt(vec["s"], vec["t"]); stage(INIT_S)
    for _ in range(maxit):
        K(vec["q"], vec["t"], False); stage(STEP_Q); stage(STEP_XR); Kt(vec["s"], vec["t"]); stage(STEP_S); stage(STEP_P)
    res = hip.CglsResult()
    hip.check(L_.prost_hip_cgls_result(state.ptr, C.byref(res), None))
    assert res.iterations == k_ref, (res.iterations, k_ref)
    assert res.converged == (1 if k_ref < maxit else 0)
    rtol = 2e-4 if dtype == np.float32 else 1e-10
    err = float(np.abs(dx.to_host() - x_ref).max()) / max(1.0, float(np.abs(x_ref).max()))
    assert err <= rtol, (err, res.iterations)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", ["c4", "side_by_side_3d", "side_by_side_3d_vec", "uncovered_rows"])
def test_cgls_fused_rounds_at_the_c_abi(hip, dtype, shape):
    """prost_hip_cgls_round_*: a CG round in four launches -- the operator (CSR and gradient blocks, evaluated by the thread that
    owns the output element, blocks in order) applied inside the kernels, alpha / beta / the stopping test formed by the
    consuming kernels, one scalar record per round -- against (a) the numpy restatement of cgls.hpp on the assembled scipy
    matrix and (b) the staged rounds of prost_hip_cgls_stage_* with the separate operator kernels: same iteration counts,
    same x to round-off (only the grouping of the partial sums differs).
    c4: K = [W; grad2d] as in BASELINE config 4; side_by_side_3d(_vec): K = [A B; grad3d 0]; uncovered_rows: rows no block writes."""
    import scipy.sparse as sp
    from reference_matrices import spmat_gradient2d, spmat_gradient3d
    rng = np.random.default_rng(5)
    L_ = hip.lib()
    dev_ = lambda a: hip.DeviceArray.from_host(np.ascontiguousarray(a))
    keep, blocks = [], []

    def csr_block(M, row, col):
        M = sp.csr_matrix(M).astype(dtype); M.sort_indices()
        Mt = sp.csr_matrix(M.T).astype(dtype); Mt.sort_indices()
        arrs = [dev_(M.data), dev_(M.indptr.astype(np.int32)), dev_(M.indices.astype(np.int32)),
                dev_(Mt.data), dev_(Mt.indptr.astype(np.int32)), dev_(Mt.indices.astype(np.int32))]
        keep.append(arrs)
        blocks.append(dict(kind=hip.OP_CSR, row=row, col=col, nrows=M.shape[0], ncols=M.shape[1], M=M, arrs=arrs, dims=(0, 0, 0)))

    def grad_block(nx, ny, L, d3, row, col):
        G = (spmat_gradient3d if d3 else spmat_gradient2d)(nx, ny, L)
        blocks.append(dict(kind=hip.OP_GRAD3D if d3 else hip.OP_GRAD2D, row=row, col=col, nrows=G.shape[0], ncols=G.shape[1], M=sp.csr_matrix(G), arrs=None, dims=(nx, ny, L)))

    if shape == "c4":
        nx, ny = 24, 36
        npx = nx * ny
        W = sp.hstack([sp.diags(rng.uniform(-0.5, 0.5, npx)), sp.diags(rng.uniform(-0.5, 0.5, npx))])
        csr_block(W, 0, 0); grad_block(nx, ny, 2, False, npx, 0)
        m, n = npx + 4 * npx, 2 * npx
    elif shape.startswith("side_by_side_3d"):
        nx, ny, L = (6, 8, 4) if shape.endswith("_vec") else (9, 14, 5)      # _vec: every boundary on a multiple of 4 -> 16 bytes of rows per lane
        npx = nx * ny * L
        csr_block(sp.random(300, npx, density=0.004, random_state=1), 0, 0)
        csr_block(sp.random(300, 120, density=0.03, random_state=2), 0, npx)
        grad_block(nx, ny, L, True, 300, 0)
        m, n = 300 + 3 * npx, npx + 120
    else:
        nx, ny = 10, 12
        npx = nx * ny
        csr_block(sp.random(50, npx, density=0.02, random_state=3), 7, 0)
        grad_block(nx, ny, 1, False, 64, 0)
        m, n = 64 + 2 * npx + 5, npx
    K = sp.lil_matrix((m, n))
    for b in blocks:
        K[b["row"]:b["row"] + b["nrows"], b["col"]:b["col"] + b["ncols"]] = b["M"]
    K = sp.csr_matrix(K).astype(dtype)
    op = hip.FusedOp(); op.nblocks = len(blocks)
    for i, b in enumerate(blocks):
        o = op.block[i]
        o.kind, o.row, o.col, o.nrows, o.ncols = b["kind"], b["row"], b["col"], b["nrows"], b["ncols"]
        o.nx, o.ny, o.L = b["dims"]
        if b["arrs"]:
            o.val, o.ptr, o.ind, o.val_t, o.ptr_t, o.ind_t = (a.ptr.value for a in b["arrs"])
    assert L_.prost_hip_fused_op_supported(C.byref(op), C.c_uint64(m), C.c_uint64(n)) == 1
    last_row = max(b["row"] + b["nrows"] for b in blocks)
    assert L_.prost_hip_fused_op_supported(C.byref(op), C.c_uint64(last_row - 1), C.c_uint64(n)) == 0   # a block sticks out

    sig = rng.uniform(0.3, 2, m).astype(dtype); tau = rng.uniform(0.3, 2, n).astype(dtype)
    bvec = rng.standard_normal(m).astype(dtype)
    x0 = rng.standard_normal(n).astype(dtype)
    maxit, tol = (8 if dtype == np.float32 else 12), 1e-3
    x_ref, k_ref = _cgls_reference(K, bvec, x0, sig, tau, 1.0, tol, maxit, dtype)
    Kt = sp.csr_matrix(K.T); Kt.sort_indices(); K.sort_indices()
    dK = [dev_(K.data), dev_(K.indptr.astype(np.int32)), dev_(K.indices.astype(np.int32))]
    dKt = [dev_(Kt.data), dev_(Kt.indptr.astype(np.int32)), dev_(Kt.indices.astype(np.int32))]
    INIT_X, INIT_R, INIT_R2, INIT_S, STEP_Q, STEP_XR, STEP_S, STEP_P = range(8)
    results = {}
    for mode in ("fused", "staged"):
        vec = {k: hip.DeviceArray.zeros(sz, dtype) for k, sz in (("p", n), ("q", m), ("r", m), ("s", n), ("t", max(m, n)))}
        db, dx, dsig, dtau = dev_(bvec), dev_(x0), dev_(sig), dev_(tau)
        rec = L_.prost_hip_cgls_state_bytes()
        state = hip.DeviceArray.zeros((maxit + 6) * rec // 8 + 1, np.float64)
        ws = hip.DeviceArray(L_.prost_hip_cgls_workspace_bytes() // 8, np.float64)
        d = hip.CglsDesc()
        d.state, d.workspace, d.b, d.x = state.ptr.value, ws.ptr.value, db.ptr.value, dx.ptr.value
        d.p, d.q, d.r, d.s, d.t = (vec[k].ptr.value for k in "pqrst")
        d.sigma, d.tau, d.m, d.n, d.shift, d.tol, d.host_done, d.epoch = dsig.ptr.value, dtau.ptr.value, m, n, 1.0, tol, None, 1
        stage = lambda which: hip.check(hip.fn("cgls_stage", dtype)(which, C.byref(d), None))
        Kf = lambda res, rhs, acc: hip.check(hip.fn("csr_spmv_acc" if acc else "csr_spmv", dtype)(res.ptr, rhs.ptr, hip.sz(m), hip.sz(K.nnz), dK[0].ptr, dK[1].ptr, dK[2].ptr, None))
        Ka = lambda res, rhs: hip.check(hip.fn("csr_spmv_acc", dtype)(res.ptr, rhs.ptr, hip.sz(n), hip.sz(K.nnz), dKt[0].ptr, dKt[1].ptr, dKt[2].ptr, None))
        rounds = maxit + 3               # rounds queued after the stopping test only hand the record on
        res = hip.CglsResult()
        if mode == "fused":
            hip.check(hip.fn("cgls_init_fused", dtype)(C.byref(d), C.byref(op), None))
        else:
            stage(INIT_X); stage(INIT_R); Kf(vec["r"], vec["t"], True); stage(INIT_R2); Ka(vec["s"], vec["t"]); stage(INIT_S)
        if mode == "fused":
            for j in range(rounds):
                hip.check(hip.fn("cgls_round", dtype)(C.byref(d), C.byref(op), j, None))
                if j + 1 == maxit:
                    hip.check(L_.prost_hip_cgls_result_at(state.ptr, maxit, C.byref(res), None))
                    x_at_maxit = dx.to_host()
            last = hip.CglsResult()
            hip.check(L_.prost_hip_cgls_result_at(state.ptr, rounds, C.byref(last), None))
            if res.converged:            # the extra rounds changed nothing
                assert (last.iterations, last.converged) == (res.iterations, res.converged) and np.array_equal(dx.to_host(), x_at_maxit)
            results[mode] = (x_at_maxit, res.iterations, res.converged)
        else:
            for j in range(maxit):
                Kf(vec["q"], vec["t"], False); stage(STEP_Q); stage(STEP_XR); Ka(vec["s"], vec["t"]); stage(STEP_S); stage(STEP_P)
            hip.check(L_.prost_hip_cgls_result(state.ptr, C.byref(res), None))
            results[mode] = (dx.to_host(), res.iterations, res.converged)
    rtol = 2e-4 if dtype == np.float32 else 1e-10
    scale = max(1.0, float(np.abs(x_ref).max()))
    for mode, (x, its, conv) in results.items():
        assert its == k_ref, (mode, its, k_ref)
        assert conv == (1 if k_ref < maxit else 0), mode
        assert float(np.abs(x - x_ref).max()) / scale <= rtol, (mode, float(np.abs(x - x_ref).max()) / scale)
    # fused against staged: the same arithmetic per element, different grouping of the partial sums only
    assert float(np.abs(results["fused"][0] - results["staged"][0]).max()) / scale <= (2e-5 if dtype == np.float32 else 1e-12)


def test_cgls_converges_to_the_damped_least_squares_solution(hip):
    """What cgls::Solve (cgls.hpp:222-371) computes on GemvPrecondK (backend_admm.cu:199-272) is the minimiser of
    |A' x - b|^2 + shift |x|^2 with A' = sqrt(Sigma) A sqrt(Tau): the device-resident CG (prost_hip_cgls_stage_*), run to a tight
    tolerance in fp64, must reach the solution scipy computes -- lsqr with damp = sqrt(shift) and the normal equations solved
    directly.  Pins the stage kernels to the mathematical definition where no reference vector exists."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    dtype = np.float64
    rng = np.random.default_rng(33)
    L_ = hip.lib()
    m, n, shift = 400, 260, 1.0
    A = sp.random(m, n, density=0.04, random_state=9, format="csr", dtype=np.float64); A.sort_indices()
    At = A.T.tocsr(); At.sort_indices()
    sig = rng.uniform(0.3, 2, m); tau = rng.uniform(0.3, 2, n)
    b = rng.standard_normal(m)
    Abar = sp.diags(np.sqrt(sig)) @ A @ sp.diags(np.sqrt(tau))
    x_direct = spla.spsolve((Abar.T @ Abar + shift * sp.identity(n)).tocsc(), Abar.T @ b)
    x_lsqr = spla.lsqr(Abar, b, damp=np.sqrt(shift), atol=1e-14, btol=1e-14, iter_lim=5000)[0]
    assert np.allclose(x_lsqr, x_direct, rtol=1e-8, atol=1e-10)
    dev_ = lambda a: hip.DeviceArray.from_host(np.ascontiguousarray(a))
    dA = [dev_(A.data), dev_(A.indptr.astype(np.int32)), dev_(A.indices.astype(np.int32))]
    dAt = [dev_(At.data), dev_(At.indptr.astype(np.int32)), dev_(At.indices.astype(np.int32))]
    vec = {k: hip.DeviceArray.zeros(sz, dtype) for k, sz in (("p", n), ("q", m), ("r", m), ("s", n), ("t", max(m, n)))}
    db, dx, dsig, dtau = dev_(b), dev_(np.zeros(n)), dev_(sig), dev_(tau)
    state = hip.DeviceArray.zeros(L_.prost_hip_cgls_state_bytes() // 8 + 1, np.float64)
    ws = hip.DeviceArray(L_.prost_hip_cgls_workspace_bytes() // 8, np.float64)
    d = hip.CglsDesc()
    d.state, d.workspace, d.b, d.x = state.ptr.value, ws.ptr.value, db.ptr.value, dx.ptr.value
    d.p, d.q, d.r, d.s, d.t = (vec[k].ptr.value for k in "pqrst")
    d.sigma, d.tau, d.m, d.n, d.shift, d.tol, d.host_done, d.epoch = dsig.ptr.value, dtau.ptr.value, m, n, shift, 1e-13, None, 1
    stage = lambda which: hip.check(hip.fn("cgls_stage", dtype)(which, C.byref(d), None))
    K = lambda res, rhs, acc: hip.check(hip.fn("csr_spmv_acc" if acc else "csr_spmv", dtype)(res.ptr, rhs.ptr, hip.sz(m), hip.sz(A.nnz), dA[0].ptr, dA[1].ptr, dA[2].ptr, None))
    Kt = lambda res, rhs: hip.check(hip.fn("csr_spmv_acc", dtype)(res.ptr, rhs.ptr, hip.sz(n), hip.sz(A.nnz), dAt[0].ptr, dAt[1].ptr, dAt[2].ptr, None))
    INIT_X, INIT_R, INIT_R2, INIT_S, STEP_Q, STEP_XR, STEP_S, STEP_P = range(8)
    stage(INIT_X); stage(INIT_R); K(vec["r"], vec["t"], True); stage(INIT_R2); Kt(vec["s"], vec["t"]); stage(INIT_S)
    for _ in range(400):
        K(vec["q"], vec["t"], False); stage(STEP_Q); stage(STEP_XR); Kt(vec["s"], vec["t"]); stage(STEP_S); stage(STEP_P)
    res = hip.CglsResult()
    hip.check(L_.prost_hip_cgls_result(state.ptr, C.byref(res), None))
    assert res.converged == 1 and res.iterations < 400
    x = dx.to_host()
    assert np.allclose(x, x_direct, rtol=1e-8, atol=1e-10), float(np.abs(x - x_direct).max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_epi_quad(hip, dtype):
    rng = np.random.default_rng(6)
    count, dim = 2000, 4
    arg = rng.uniform(-2, 2, count * dim).astype(dtype)
    a = rng.uniform(0.5, 2, count); b = rng.uniform(-1, 1, count * (dim - 1)); c = rng.uniform(-1, 1, count)
    b[: count] = 0
    for av, cv in ((a, c), (1.3, 0.2)):
        ref = oracle.prox_epi_quad(arg, count, dim, av, b, cv)
        av_ = np.atleast_1d(av); cv_ = np.atleast_1d(cv)
        da = dev(hip, av_.astype(dtype)) if av_.size > 1 else None
        dc = dev(hip, cv_.astype(dtype)) if cv_.size > 1 else None
        res = hip.DeviceArray.zeros(count * dim, dtype)
        hip.check(hip.fn("prox_epi_quad", dtype)(res.ptr, dev(hip, arg).ptr, hip.sz(count), hip.sz(dim), da.ptr if da else None,
                                                 hip.dbl(av_[0]), dev(hip, b.astype(dtype)).ptr, dc.ptr if dc else None, hip.dbl(cv_[0]), None))
        tol = 2e-5 if dtype == np.float32 else 1e-11
        assert np.allclose(res.to_host(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_moreau_and_pdhg_elementwise(hip, dtype):
    rng = np.random.default_rng(7)
    n = 5000
    v = [rng.standard_normal(n).astype(dtype) for _ in range(5)]
    td = rng.uniform(0.1, 2, n).astype(dtype)
    tau, sigma, theta = dtype(0.37), dtype(1.9), dtype(0.8)
    d = [dev(hip, a) for a in v]; dtd = dev(hip, td)
    for inv in (0, 1):
        out = hip.DeviceArray.zeros(n, dtype)
        hip.check(hip.fn("moreau_prescale", dtype)(out.ptr, d[0].ptr, dtd.ptr, hip.dbl(tau), inv, hip.sz(n), None))
        ref = v[0] * (tau * td) if inv else v[0] / (tau * td)
        assert np.array_equal(out.to_host(), ref.astype(dtype))
        res = dev(hip, v[1])
        hip.check(hip.fn("moreau_postscale", dtype)(res.ptr, d[0].ptr, dtd.ptr, hip.dbl(tau), inv, hip.sz(n), None))
        ref = v[0] - v[1] / (tau * td) if inv else v[0] - tau * td * v[1]
        assert np.array_equal(res.to_host(), ref.astype(dtype))
    out = hip.DeviceArray.zeros(n, dtype)
    hip.check(hip.fn("pdhg_primal_arg", dtype)(out.ptr, d[0].ptr, dtd.ptr, d[1].ptr, hip.dbl(tau), hip.sz(n), None))
    assert np.array_equal(out.to_host(), (v[0] - tau * td * v[1]).astype(dtype))
    hip.check(hip.fn("pdhg_dual_arg", dtype)(out.ptr, d[0].ptr, dtd.ptr, d[1].ptr, d[2].ptr, hip.dbl(sigma), hip.dbl(theta), hip.sz(n), None))
    assert np.array_equal(out.to_host(), (v[0] + sigma * td * ((1 + theta) * v[1] - theta * v[2])).astype(dtype))
    # residual reductions: terms in T, sums in double
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    out2 = hip.DeviceArray.zeros(2, np.float64)
    hip.check(hip.fn("pdhg_residual_primal", dtype)(out2.ptr, d[0].ptr, d[1].ptr, dtd.ptr, d[2].ptr, d[3].ptr, hip.dbl(sigma), hip.dbl(theta), hip.sz(n), ws.ptr, None))
    sq = np.sqrt(td)
    z_hat = ((v[0] - v[1]) / (sigma * sq) + sq * ((1 + theta) * v[3] - theta * v[2])).astype(dtype)
    diff = (z_hat - sq * v[3]).astype(dtype)
    ref = np.array([np.sum((diff * diff).astype(np.float64)), np.sum((z_hat * z_hat).astype(np.float64))])
    assert np.allclose(out2.to_host(), ref, rtol=1e-12)
    hip.check(hip.fn("pdhg_residual_dual", dtype)(out2.ptr, d[0].ptr, d[1].ptr, dtd.ptr, d[2].ptr, d[3].ptr, hip.dbl(tau), hip.sz(n), ws.ptr, None))
    w_hat = ((v[0] - v[1]) / (tau * sq) - sq * v[2]).astype(dtype)
    diff = (w_hat + sq * v[3]).astype(dtype)
    ref = np.array([np.sum((diff * diff).astype(np.float64)), np.sum((w_hat * w_hat).astype(np.float64))])
    assert np.allclose(out2.to_host(), ref, rtol=1e-12)
    hip.check(hip.fn("nrm2", dtype)(out2.ptr, d[0].ptr, hip.sz(n), ws.ptr, None))
    assert np.isclose(out2.to_host()[0], np.sqrt(np.sum(v[0].astype(np.float64) ** 2)), rtol=1e-13)


def _fused_desc(hip, dtype, nx, ny, L, g_fn, g_coeffs, f_fn, f_coeffs, Tval, Sval):
    d = hip.FusedDesc()
    d.is3d = 0; d.nx, d.ny, d.L = nx, ny, L
    d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID[f_fn]
    gp, gv, k1 = hip.coeff_args(g_coeffs, dtype, nx * ny * L)
    fp, fv, k2 = hip.coeff_args(f_coeffs, dtype, nx * ny)
    for i in range(7):
        d.g_coeff_ptr[i] = gp[i]; d.g_coeff_val[i] = gv[i]
        d.f_coeff_ptr[i] = fp[i]; d.f_coeff_val[i] = fv[i]
    d.T_val, d.S_val = Tval, Sval
    return d, (k1, k2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(16, 12, 1), (40, 1028, 1), (33, 130, 2), (9, 7, 3), (5, 2052, 4), (64, 64, 1)])
@pytest.mark.parametrize("fns", [("square", "ind_leq0"), ("abs", "huber")])
def test_fused_passes_match_unfused_oracle(hip, dtype, shape, fns):
    """One fused primal + dual pass == the reference's unfused sequence evaluated by the oracle
    (backend_pdhg.cu:313-370), including the residual sums of :392-431."""
    nx, ny, L = shape
    g_fn, f_fn = fns
    rng = np.random.default_rng(8)
    n, m = nx * ny * L, 2 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    y_prev = rng.uniform(-1, 1, m).astype(dtype); x_old = rng.uniform(0, 1, n).astype(dtype)
    f = rng.uniform(0, 1, n)
    tau, sigma, theta = dtype(0.9), dtype(1.1), dtype(0.85)
    Tval, Sval = dtype(0.25), dtype(0.5)
    g_coeffs = [1.0, f, 10.0, 0.0, 0.0, 0.3, 0.0]
    f_coeffs = [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0]
    desc, keep = _fused_desc(hip, dtype, nx, ny, L, g_fn, g_coeffs, f_fn, f_coeffs, Tval, Sval)
    assert hip.lib().prost_hip_fused_supported(C.byref(desc), 0) == 1
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    Td, Sd = np.full(n, Tval, dtype), np.full(m, Sval, dtype)
    for use_kty, use_prev in ((1, 1), (0, 0), (1, 0)):
        # ---- oracle, unfused ----
        kty = oracle.grad2d(y, nx, ny, L, adjoint=True) if use_kty else np.zeros(n, dtype)
        ktyp = oracle.grad2d(y_prev, nx, ny, L, adjoint=True) if use_prev else np.zeros(n, dtype)
        temp = (x - tau * Td * kty).astype(dtype)
        x_ref = oracle.prox_elem(0, g_fn, temp, Td, tau, n, 1, False, g_coeffs)
        sq = np.sqrt(Td)
        w_hat = ((x - x_ref) / (tau * sq) - sq * ktyp).astype(dtype); diff = (w_hat + sq * kty).astype(dtype)
        dres_ref = np.array([np.sum((diff * diff).astype(np.float64)), np.sum((w_hat * w_hat).astype(np.float64))])
        # ---- fused primal ----
        x_new = hip.DeviceArray.zeros(n, dtype); out2 = hip.DeviceArray.zeros(2, np.float64)
        hip.check(hip.fn("fused_primal", dtype)(C.byref(desc), x_new.ptr, dev(hip, x).ptr, dev(hip, y).ptr, dev(hip, y_prev).ptr,
                                                hip.dbl(tau), use_kty, use_prev, out2.ptr, ws.ptr, None))
        assert np.array_equal(x_new.to_host(), x_ref)
        assert np.allclose(out2.to_host(), dres_ref, rtol=1e-11)
        x_new2 = hip.DeviceArray.zeros(n, dtype)
        hip.check(hip.fn("fused_primal", dtype)(C.byref(desc), x_new2.ptr, dev(hip, x).ptr, dev(hip, y).ptr, None,
                                                hip.dbl(tau), use_kty, use_prev, None, None, None))
        assert np.array_equal(x_new2.to_host(), x_ref)
    for use_kxp in (1, 0):
        kx = oracle.grad2d(x, nx, ny, L)
        kxp = oracle.grad2d(x_old, nx, ny, L) if use_kxp else np.zeros(m, dtype)
        temp = (y + sigma * Sd * ((1 + theta) * kx - theta * kxp)).astype(dtype)
        y_ref = oracle.prox_elem(1, f_fn, temp, Sd, sigma, nx * ny, 2 * L, False, f_coeffs)
        sq = np.sqrt(Sd)
        z_hat = ((y - y_ref) / (sigma * sq) + sq * ((1 + theta) * kx - theta * kxp)).astype(dtype); diff = (z_hat - sq * kx).astype(dtype)
        pres_ref = np.array([np.sum((diff * diff).astype(np.float64)), np.sum((z_hat * z_hat).astype(np.float64))])
        y_new = hip.DeviceArray.zeros(m, dtype); out2 = hip.DeviceArray.zeros(2, np.float64)
        hip.check(hip.fn("fused_dual", dtype)(C.byref(desc), y_new.ptr, dev(hip, y).ptr, dev(hip, x).ptr, dev(hip, x_old).ptr,
                                              hip.dbl(sigma), hip.dbl(theta), use_kxp, out2.ptr, ws.ptr, None))
        assert np.array_equal(y_new.to_host(), y_ref)
        assert np.allclose(out2.to_host(), pres_ref, rtol=1e-11)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(16, 12, 1), (40, 1028, 1), (33, 256, 2), (70, 252, 1), (5, 2052, 2), (64, 64, 1), (2, 4, 1),
                                   (16, 13, 1), (40, 1030, 1), (33, 255, 2), (70, 250, 1), (9, 501, 1), (3, 5, 2)])
@pytest.mark.parametrize("fns", [("square", "ind_leq0"), ("abs", "ind_leq0"), ("abs", "huber")])
def test_single_kernel_iteration_equals_two_passes(hip, dtype, shape, fns):
    """prost_hip_fused_iteration (7 floats/pixel) == fused_primal + fused_dual (11 floats/pixel), bit for bit,
    for every column-chunk size (halo columns) and the iteration-0 flags"""
    nx, ny, L = shape
    g_fn, f_fn = fns
    rng = np.random.default_rng(9)
    n, m = nx * ny * L, 2 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    tau, sigma, theta = 0.9, 1.1, 0.85
    for g_coeffs in ([1.0, f, 10.0, 0.0, 0.0, 0.3, 0.0], [1.0, 0.5, 10.0, 0.0, 0.0, 0.3, 0.0], [f + 0.5, f, 10.0, f * 0.1, 0.0, 0.3, 0.0]):
        desc, keep = _fused_desc(hip, dtype, nx, ny, L, g_fn, g_coeffs, f_fn, [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0], 0.25, 0.5)
        assert hip.lib().prost_hip_fused_iteration_supported(C.byref(desc), 0 if dtype == np.float32 else 1)
        dx, dy = dev(hip, x), dev(hip, y)
        dyp = dev(hip, rng.uniform(-1, 1, m).astype(dtype))
        ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
        for use_kty, use_kxp, use_ktyp in ((1, 1, 1), (0, 0, 0), (1, 0, 0)):
            x_ref = hip.DeviceArray.zeros(n, dtype); y_ref = hip.DeviceArray.zeros(m, dtype)
            rd = hip.DeviceArray.zeros(2, np.float64); rp = hip.DeviceArray.zeros(2, np.float64)
            hip.check(hip.fn("fused_primal", dtype)(C.byref(desc), x_ref.ptr, dx.ptr, dy.ptr, dyp.ptr, hip.dbl(tau), use_kty, use_ktyp, rd.ptr, ws.ptr, None))
            hip.check(hip.fn("fused_dual", dtype)(C.byref(desc), y_ref.ptr, dy.ptr, x_ref.ptr, dx.ptr, hip.dbl(sigma), hip.dbl(theta), use_kxp, rp.ptr, ws.ptr, None))
            res_ref = np.concatenate([rp.to_host(), rd.to_host()])
            for cols in (0, 1, 3, 8, 1000):
                x_new = hip.DeviceArray.zeros(n, dtype); y_new = hip.DeviceArray.zeros(m, dtype)
                hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), x_new.ptr, y_new.ptr, dx.ptr, dy.ptr, None, hip.dbl(tau), hip.dbl(sigma),
                                                           hip.dbl(theta), use_kty, use_kxp, 0, cols, None, None, None))
                assert np.array_equal(x_new.to_host(), x_ref.to_host()), (cols, use_kty)
                assert np.array_equal(y_new.to_host(), y_ref.to_host()), (cols, use_kxp)
                # residual variant: same iterates + the four residual sums of the two-pass kernels
                x_new2 = hip.DeviceArray.zeros(n, dtype); y_new2 = hip.DeviceArray.zeros(m, dtype); r4 = hip.DeviceArray.zeros(4, np.float64)
                hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), x_new2.ptr, y_new2.ptr, dx.ptr, dy.ptr, dyp.ptr, hip.dbl(tau), hip.dbl(sigma),
                                                           hip.dbl(theta), use_kty, use_kxp, use_ktyp, cols, r4.ptr, ws.ptr, None))
                assert np.array_equal(x_new2.to_host(), x_ref.to_host()) and np.array_equal(y_new2.to_host(), y_ref.to_host())
                assert np.allclose(r4.to_host(), res_ref, rtol=1e-11, atol=1e-300), (cols, r4.to_host(), res_ref)
                for d_ in (x_new, y_new, x_new2, y_new2, r4):
                    d_.free()
            for d_ in (x_ref, y_ref, rd, rp):
                d_.free()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(16, 12), (40, 1028), (33, 256), (70, 248), (70, 252), (5, 2052), (64, 64), (4, 4), (7, 496), (131, 500),
                                   (16, 13), (40, 1030), (33, 255), (70, 249), (9, 501), (6, 7), (50, 247)])
@pytest.mark.parametrize("fns", [("square", "ind_leq0"), ("abs", "ind_leq0"), ("abs", "huber")])
def test_double_iteration_kernel_equals_two_single_launches(hip, dtype, shape, fns):
    """prost_hip_fused_iteration2 (two PDHG iterations, intermediate iterate kept in registers) ==
    two prost_hip_fused_iteration launches with the step sizes of iteration k and k+1, bit for bit,
    for every column-chunk size (3-column pipeline warm-up, halo lanes at strip borders)"""
    nx, ny = shape
    g_fn, f_fn = fns
    rng = np.random.default_rng(11)
    n, m = nx * ny, 2 * nx * ny
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    tau = (C.c_double * 2)(0.9, 0.7); sigma = (C.c_double * 2)(1.1, 1.4); theta = (C.c_double * 2)(0.85, 0.8)
    for g_coeffs in ([1.0, f, 10.0, 0.0, 0.0, 0.3, 0.0], [1.0, 0.5, 10.0, 0.0, 0.0, 0.3, 0.0], [f + 0.5, f, 10.0, f * 0.1, 0.0, 0.3, 0.0]):
        desc, keep = _fused_desc(hip, dtype, nx, ny, 1, g_fn, g_coeffs, f_fn, [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0], 0.25, 0.5)
        assert hip.lib().prost_hip_fused_iteration2_supported(C.byref(desc), 0 if dtype == np.float32 else 1)
        dx, dy = dev(hip, x), dev(hip, y)
        x1 = hip.DeviceArray.zeros(n, dtype); y1 = hip.DeviceArray.zeros(m, dtype)
        x_ref = hip.DeviceArray.zeros(n, dtype); y_ref = hip.DeviceArray.zeros(m, dtype)
        hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), x1.ptr, y1.ptr, dx.ptr, dy.ptr, None, hip.dbl(tau[0]), hip.dbl(sigma[0]),
                                                   hip.dbl(theta[0]), 1, 1, 0, 0, None, None, None))
        hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), x_ref.ptr, y_ref.ptr, x1.ptr, y1.ptr, None, hip.dbl(tau[1]), hip.dbl(sigma[1]),
                                                   hip.dbl(theta[1]), 1, 1, 0, 0, None, None, None))
        # residual sums of the SECOND iteration as the single-iteration kernel reports them
        ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
        r4s = hip.DeviceArray.zeros(4, np.float64); xs = hip.DeviceArray.zeros(n, dtype); ys = hip.DeviceArray.zeros(m, dtype)
        hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), xs.ptr, ys.ptr, x1.ptr, y1.ptr, dy.ptr, hip.dbl(tau[1]), hip.dbl(sigma[1]),
                                                   hip.dbl(theta[1]), 1, 1, 1, 0, r4s.ptr, ws.ptr, None))
        res_ref = r4s.to_host()
        xr, yr, x1r, y1r = x_ref.to_host(), y_ref.to_host(), x1.to_host(), y1.to_host()
        for cols in (0, 1, 2, 3, 5, 8, 1000):
            for mid, res in ((False, False), (True, False), (False, True), (True, True)):
                x2 = hip.DeviceArray.from_host(np.full(n, 7.0, dtype)); y2 = hip.DeviceArray.from_host(np.full(m, 7.0, dtype))
                xm = hip.DeviceArray.from_host(np.full(n, 7.0, dtype)); ym = hip.DeviceArray.from_host(np.full(m, 7.0, dtype))
                r4 = hip.DeviceArray.zeros(4, np.float64)
                hip.check(hip.fn("fused_iteration2", dtype)(C.byref(desc), x2.ptr, y2.ptr, dx.ptr, dy.ptr, xm.ptr if mid else None, ym.ptr if mid else None,
                                                            tau, sigma, theta, cols, r4.ptr if res else None, ws.ptr if res else None, None))
                assert np.array_equal(x2.to_host(), xr), (cols, mid, res, np.flatnonzero(x2.to_host() != xr)[:8])
                assert np.array_equal(y2.to_host(), yr), (cols, mid, res, np.flatnonzero(y2.to_host() != yr)[:8])
                if mid:
                    assert np.array_equal(xm.to_host(), x1r) and np.array_equal(ym.to_host(), y1r), (cols, mid, res)
                if res:
                    # the straight-line fp32 instances form the residual TERMS with fused multiply-adds (tolerance-compared sums,
                    # kernels_fused_iter2.hip); every other instance evaluates the reference's expressions and only the order differs
                    # (a sum that is pure round-off -- 1e-15 beside sums of order 1 -- is compared on the scale of the largest one)
                    fast = fns[1] == "ind_leq0"
                    eps = 1e-9 if dtype == np.float32 else 1e-15
                    assert np.allclose(r4.to_host(), res_ref, rtol=2e-6 if fast and dtype == np.float32 else 1e-11,
                                       atol=eps * np.abs(res_ref).max() if fast else 1e-300), (cols, r4.to_host(), res_ref)
                for d_ in (x2, y2, xm, ym, r4):
                    d_.free()
        for d_ in (x1, y1, x_ref, y_ref, xs, ys, r4s):
            d_.free()


@pytest.mark.parametrize("dtype", DTYPES)
def test_double_iteration_kernel_special_values(hip, dtype):
    """The pair kernel's short sqrt / division forms (device_math.hpp: double-reciprocal division,
    un-scaled sqrt refinement) only run for norms in [2^-96, 2^126]; zero norms, subnormal / tiny /
    huge magnitudes and signed zeros take the general expansions.  Every regime must reproduce the
    single-iteration kernel bit for bit."""
    nx, ny = 24, 1028
    n, m = nx * ny, 2 * nx * ny
    rng = np.random.default_rng(23)
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    # flat regions (K x == 0) with y == 0 -> zero norms; tiny, subnormal and huge duals; signed zeros
    x[: n // 6] = 0.5
    y[: n // 6] = 0.0; y[n: n + n // 6] = 0.0
    y[n // 6: n // 5] *= 1e-30; y[n + n // 6: n + n // 5] *= 1e-30
    y[n // 5: n // 4] *= 1e-42; y[n + n // 5: n + n // 4] = 0.0
    y[n // 4: n // 3] *= 3e19; y[n + n // 4: n + n // 3] *= 3e19
    y[n // 3: n // 3 + 500] = -0.0
    x[n // 2: n // 2 + 300] = 1e-39; x[n // 2 + 300: n // 2 + 600] = -0.0
    tau = (C.c_double * 2)(0.9, 0.7); sigma = (C.c_double * 2)(1.1, 1.4); theta = (C.c_double * 2)(0.85, 0.8)
    desc, keep = _fused_desc(hip, dtype, nx, ny, 1, "square", [1.0, f, 10.0, 0.0, 0.0, 0.3, 0.0], "ind_leq0", [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0], 0.25, 0.5)
    dx, dy = dev(hip, x), dev(hip, y)
    x1 = hip.DeviceArray.zeros(n, dtype); y1 = hip.DeviceArray.zeros(m, dtype)
    x_ref = hip.DeviceArray.zeros(n, dtype); y_ref = hip.DeviceArray.zeros(m, dtype)
    hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), x1.ptr, y1.ptr, dx.ptr, dy.ptr, None, hip.dbl(tau[0]), hip.dbl(sigma[0]), hip.dbl(theta[0]), 1, 1, 0, 0, None, None, None))
    hip.check(hip.fn("fused_iteration", dtype)(C.byref(desc), x_ref.ptr, y_ref.ptr, x1.ptr, y1.ptr, None, hip.dbl(tau[1]), hip.dbl(sigma[1]), hip.dbl(theta[1]), 1, 1, 0, 0, None, None, None))
    for cols in (0, 5):
        x2 = hip.DeviceArray.zeros(n, dtype); y2 = hip.DeviceArray.zeros(m, dtype); xm = hip.DeviceArray.zeros(n, dtype); ym = hip.DeviceArray.zeros(m, dtype)
        hip.check(hip.fn("fused_iteration2", dtype)(C.byref(desc), x2.ptr, y2.ptr, dx.ptr, dy.ptr, xm.ptr, ym.ptr, tau, sigma, theta, cols, None, None, None))
        for got, want, name in ((xm, x1, "x_mid"), (ym, y1, "y_mid"), (x2, x_ref, "x"), (y2, y_ref, "y")):
            g, w = got.to_host(), want.to_host()
            assert np.array_equal(g, w, equal_nan=True), (name, cols, np.flatnonzero(g != w)[:8])
            assert np.array_equal(np.signbit(g), np.signbit(w)), (name, cols, "sign of zero")


def test_short_division_and_sqrt_forms(hip):
    """device_math.hpp's short correctly rounded forms (used by the two-iterations kernel) against the
    compiler's IEEE expansions on 2^31 pseudo-random cases per form, evaluated on the device (round 4: the fp64 quotient with a
    shared refined reciprocal and the fp64 square root without range scaling as well -- slot 7)."""
    bad = hip.DeviceArray.zeros(8, np.uint64)
    total = np.zeros(8, dtype=np.uint64)
    for seed in range(8):
        hip.check(hip.lib().prost_hip_selftest_math(bad.ptr, C.c_uint64(1 << 28), C.c_uint64(seed), None))
        total += bad.to_host()
    names = ("division", "sqrt", "exact_division", "subtraction", "control", "division_plus_zero", "min0", "fp64_division_and_sqrt")
    assert total[:4].tolist() == [0, 0, 0, 0] and total[5:].tolist() == [0, 0, 0], dict(zip(names, total.tolist()))
    assert total[4] > 1000, "control: the harness must see the approximate reciprocal fail"
    print("selftest control mismatches:", int(total[4]), "of", 8 << 28)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(6, 8, 3), (21, 1028, 3), (33, 64, 4), (3, 256, 4), (40, 508, 3), (9, 321, 3), (14, 63, 4), (5, 130, 3), (7, 1, 3), (11, 2, 4)])
@pytest.mark.parametrize("fns", [("square", "ind_leq0"), ("abs", "huber")])
@pytest.mark.parametrize("vector_b", [True, False])
def test_multichannel_single_kernel_equals_two_passes(hip, dtype, shape, fns, vector_b):
    """prost_hip_fused_iteration_mc (3 / 4 channels on the wavefronts of one workgroup, the norm over the 2 L gradient
    components assembled in LDS in the reference's component order) against the two-pass kernels: same bits for
    x_new and y_new for every chunk width and flag combination."""
    nx, ny, L = shape
    g_fn, f_fn = fns
    rng = np.random.default_rng(7)
    n, m = nx * ny * L, 2 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    tau, sigma, theta = dtype(0.9), dtype(1.1), dtype(0.85)
    g_coeffs = [1.0, f if vector_b else 0.4, 10.0, 0.0, 0.0, 0.3, 0.0]
    f_coeffs = [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = nx, ny, L
    d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID[f_fn]
    gp, gv, k1 = hip.coeff_args(g_coeffs, dtype, n)
    fp, fv, k2 = hip.coeff_args(f_coeffs, dtype, n)
    for i in range(7):
        d.g_coeff_ptr[i] = gp[i]; d.g_coeff_val[i] = gv[i]; d.f_coeff_ptr[i] = fp[i]; d.f_coeff_val[i] = fv[i]
    d.T_val, d.S_val = 0.25, 0.5
    dt = 0 if dtype == np.float32 else 1
    # (heights that are not a whole number of 16-byte row groups -- 321, 63, 130 in double, 1, 2 -- run the one-row-per-lane instance)
    assert hip.lib().prost_hip_fused_iteration_mc_supported(C.byref(d), dt) == 1
    d1 = hip.FusedDesc(); d1.is3d = 0; d1.nx, d1.ny, d1.L = nx, ny, 1
    assert hip.lib().prost_hip_fused_iteration_mc_supported(C.byref(d1), dt) == 0          # L = 1, 2: kernels_fused_iter.hip
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    dx, dy = hip.DeviceArray.from_host(x), hip.DeviceArray.from_host(y)
    dyp = hip.DeviceArray.from_host(rng.uniform(-1, 1, m).astype(dtype))
    for use_kty, use_kxp, use_ktyp in ((1, 1, 1), (0, 0, 0), (1, 0, 0), (1, 1, 0)):
        x_ref = hip.DeviceArray.zeros(n, dtype); y_ref = hip.DeviceArray.zeros(m, dtype)
        rd = hip.DeviceArray.zeros(2, np.float64); rp = hip.DeviceArray.zeros(2, np.float64)
        hip.check(hip.fn("fused_primal", dtype)(C.byref(d), x_ref.ptr, dx.ptr, dy.ptr, dyp.ptr, hip.dbl(tau), use_kty, use_ktyp, rd.ptr, ws.ptr, None))
        hip.check(hip.fn("fused_dual", dtype)(C.byref(d), y_ref.ptr, dy.ptr, x_ref.ptr, dx.ptr, hip.dbl(sigma), hip.dbl(theta), use_kxp, rp.ptr, ws.ptr, None))
        res_ref = np.concatenate([rp.to_host(), rd.to_host()])      # {primal diff^2, primal var^2, dual diff^2, dual var^2}
        for cols in (0, 1, 2, 5, 64):
            for res in (False, True):
                x_new = hip.DeviceArray.zeros(n, dtype); y_new = hip.DeviceArray.zeros(m, dtype); r4 = hip.DeviceArray.zeros(4, np.float64)
                hip.check(hip.fn("fused_iteration_mc", dtype)(C.byref(d), x_new.ptr, y_new.ptr, dx.ptr, dy.ptr, dyp.ptr if res else None, hip.dbl(tau), hip.dbl(sigma),
                                                              hip.dbl(theta), use_kty, use_kxp, use_ktyp, cols, r4.ptr if res else None, ws.ptr if res else None, None))
                assert np.array_equal(x_new.to_host(), x_ref.to_host()), (cols, use_kty, res)
                assert np.array_equal(y_new.to_host(), y_ref.to_host()), (cols, use_kty, use_kxp, res)
                if res:
                    assert np.allclose(r4.to_host(), res_ref, rtol=1e-11, atol=1e-300), (cols, r4.to_host(), res_ref)
    hip.sync()


class _NormestDesc(C.Structure):
    _fields_ = [("workspace", C.c_void_p), ("x", C.c_void_p), ("x_temp", C.c_void_p), ("ax", C.c_void_p), ("sigma", C.c_void_p), ("tau", C.c_void_p),
                ("m", C.c_uint64), ("n", C.c_uint64), ("norm_x", C.c_double), ("out", C.c_void_p), ("norm_x_from", C.c_void_p)]


class _NormestGradDesc(C.Structure):
    _fields_ = [("is3d", C.c_int), ("nx", C.c_uint64), ("ny", C.c_uint64), ("L", C.c_uint64), ("x_in", C.c_void_p), ("x_out", C.c_void_p),
                ("tau", C.c_double), ("sigma", C.c_double), ("norm_x_from", C.c_void_p), ("out", C.c_void_p), ("workspace", C.c_void_p)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(9, 16, 1, False), (33, 250, 3, False), (7, 1028, 2, False), (12, 16, 5, True), (5, 251, 9, True), (40, 512, 3, True), (1, 8, 1, True), (6, 1, 4, False)])
def test_normest_gradient_round_equals_the_staged_round(hip, dtype, shape):
    """prost_hip_normest_grad_round (the whole power-iteration round of Problem::normest in one stencil kernel, for a single
    gradient block under constant preconditioners) against the staged round NORMEST_A, K, NORMEST_B, K^T, NORMEST_C it replaces:
    same bits in the new x, the two norms to summation order -- first round (no divide) and a later one (x / |x|)."""
    nx, ny, L, d3 = shape
    n = nx * ny * L
    m = (3 if d3 else 2) * n
    rng = np.random.default_rng(8)
    tau, sigma = dtype(1.0 / (6 if d3 else 4)), dtype(0.5)
    lib = hip.lib()
    ws = hip.DeviceArray(lib.prost_hip_cgls_workspace_bytes() // 8, np.float64)
    name = "grad3d" if d3 else "grad2d"
    for norm_prev in (0.0, 37.25):
        x0 = rng.uniform(0, 1, n).astype(dtype)
        # staged
        x = dev(hip, x0); xt = hip.DeviceArray.zeros(n, dtype); ax = hip.DeviceArray.zeros(m, dtype)
        sig = dev(hip, np.full(m, sigma, dtype)); ta = dev(hip, np.full(n, tau, dtype))
        out = hip.DeviceArray.zeros(2, np.float64); nf = dev(hip, np.array([norm_prev]))
        d = _NormestDesc(ws.ptr.value, x.ptr.value, xt.ptr.value, ax.ptr.value, sig.ptr.value, ta.ptr.value, m, n, 0.0, out.ptr.value, nf.ptr.value)
        stage = hip.fn("normest_stage", dtype)
        hip.check(stage(0, C.byref(d), None))
        hip.check(hip.fn(name + "_fwd", dtype)(ax.ptr, xt.ptr, hip.sz(nx), hip.sz(ny), hip.sz(L), 0, 0, None))
        hip.check(stage(1, C.byref(d), None))
        hip.check(hip.fn(name + "_adj", dtype)(xt.ptr, ax.ptr, hip.sz(nx), hip.sz(ny), hip.sz(L), 0, 0, None))
        hip.check(stage(2, C.byref(d), None))
        x_ref, out_ref = x.to_host(), out.to_host()
        # one kernel
        xi = dev(hip, x0); xo = hip.DeviceArray.zeros(n, dtype); out2 = hip.DeviceArray.zeros(2, np.float64)
        g = _NormestGradDesc(1 if d3 else 0, nx, ny, L, xi.ptr.value, xo.ptr.value, float(tau), float(sigma), nf.ptr.value, out2.ptr.value, ws.ptr.value)
        hip.check(hip.fn("normest_grad_round", dtype)(C.byref(g), None))
        assert np.array_equal(xo.to_host(), x_ref), (norm_prev, np.flatnonzero(xo.to_host() != x_ref)[:8])
        assert np.allclose(out2.to_host(), out_ref, rtol=1e-12, atol=0), (out2.to_host(), out_ref)
        assert np.all(out_ref > 0)
    hip.sync()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(6, 8, 3), (5, 12, 4), (20, 1028, 3), (33, 64, 2), (4, 256, 4), (40, 508, 3), (64, 252, 2), (9, 248, 4), (130, 16, 3), (11, 126, 3), (7, 63, 3), (6, 125, 2)])
@pytest.mark.parametrize("vector_b", [True, False])
@pytest.mark.parametrize("radius,g_fn", [(1.0, "square"), (1e-7, "square"), (1.0, "abs")])
def test_double_multichannel_iteration_equals_two_iterations_of_the_two_pass_kernels(hip, dtype, shape, vector_b, radius, g_fn):
    """prost_hip_fused_iteration_mc_x2 (two iterations per launch, one channel per wavefront, the squares of both dual steps
    meeting in LDS for the norm over the 2 L components of a pixel) against two iterations of the two-pass kernels, which are
    pinned to the oracle (test_fused_passes_match_unfused_oracle): same bits for x^(k+2) and all 2 L components of y^(k+2), for
    every chunk width (1: every column a chunk border; 200: one chunk), 2 / 3 / 4 channels, strip layouts (252 = one strip + 4
    rows, 1028 rows: five strips; 63 / 125 / 126 rows: one row per lane, two halo lanes on either side) and step sizes that change between the two iterations (alg2); with residual sums: the four sums
    of the second iteration against the two-pass kernels' for that iteration."""
    nx, ny, L = shape
    dt = 0 if dtype == np.float32 else 1
    rng = np.random.default_rng(13)
    n, m = nx * ny * L, 2 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    g_coeffs = [1.0, f if vector_b else 0.4, 10.0 if g_fn == "square" else 0.6, 0.0, 0.0, 0.3, 0.0]
    f_coeffs = [1.0, radius, 1.0, 0.0, 0.0, 0.3, 0.0]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = nx, ny, L
    d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID["ind_leq0"]
    gp, gv, k1 = hip.coeff_args(g_coeffs, dtype, n)
    fp, fv, k2 = hip.coeff_args(f_coeffs, dtype, n)
    for i in range(7):
        d.g_coeff_ptr[i] = gp[i]; d.g_coeff_val[i] = gv[i]; d.f_coeff_ptr[i] = fp[i]; d.f_coeff_val[i] = fv[i]
    d.T_val, d.S_val = 0.25, 0.5
    assert hip.lib().prost_hip_fused_iteration_mc_x2_supported(C.byref(d), dt) == 1
    tau, sigma, theta = [0.9, 0.61], [1.1, 1.63], [0.85, 0.67]
    dx, dy = dev(hip, x), dev(hip, y)
    x1 = hip.DeviceArray.zeros(n, dtype); y1 = hip.DeviceArray.zeros(m, dtype)
    x2 = hip.DeviceArray.zeros(n, dtype); y2 = hip.DeviceArray.zeros(m, dtype)
    P, D = hip.fn("fused_primal", dtype), hip.fn("fused_dual", dtype)
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    hip.check(P(C.byref(d), x1.ptr, dx.ptr, dy.ptr, None, hip.dbl(tau[0]), 1, 0, None, ws.ptr, None))
    hip.check(D(C.byref(d), y1.ptr, dy.ptr, x1.ptr, dx.ptr, hip.dbl(sigma[0]), hip.dbl(theta[0]), 1, None, ws.ptr, None))
    rd = hip.DeviceArray.zeros(2, np.float64); rp = hip.DeviceArray.zeros(2, np.float64)      # the second iteration as a residual iteration (y_prev = y^k)
    hip.check(P(C.byref(d), x2.ptr, x1.ptr, y1.ptr, dy.ptr, hip.dbl(tau[1]), 1, 1, rd.ptr, ws.ptr, None))
    hip.check(D(C.byref(d), y2.ptr, y1.ptr, x2.ptr, x1.ptr, hip.dbl(sigma[1]), hip.dbl(theta[1]), 1, rp.ptr, ws.ptr, None))
    res_ref = np.concatenate([rp.to_host(), rd.to_host()])      # {primal diff^2, primal var^2, dual diff^2, dual var^2}
    x_ref, y_ref = x2.to_host(), y2.to_host()
    arr = lambda v: (C.c_double * 2)(*v)
    pad, sentinel = 256, dtype(-123456.75)
    for cols, res in ((0, False), (0, True), (1, True), (2, False), (5, True), (7, False), (200, True)):
        bx = dev(hip, np.full(n + 2 * pad, sentinel, dtype)); by = dev(hip, np.full(m + 2 * pad, sentinel, dtype))      # canaries around the outputs
        esz = np.dtype(dtype).itemsize
        xo = C.c_void_p(bx.ptr.value + pad * esz); yo = C.c_void_p(by.ptr.value + pad * esz)
        r4 = hip.DeviceArray.zeros(4, np.float64)
        hip.check(hip.fn("fused_iteration_mc_x2", dtype)(C.byref(d), xo, yo, dx.ptr, dy.ptr, arr(tau), arr(sigma), arr(theta), cols,
                                                                r4.ptr if res else None, ws.ptr if res else None, None))
        if res:      # the same terms as the two-pass kernels, summed in double in another order
            assert np.allclose(r4.to_host(), res_ref, rtol=1e-11, atol=1e-300), (cols, r4.to_host(), res_ref)
        hx, hy = bx.to_host(), by.to_host()
        assert np.all(hx[:pad] == sentinel) and np.all(hx[pad + n:] == sentinel) and np.all(hy[:pad] == sentinel) and np.all(hy[pad + m:] == sentinel), cols
        assert np.array_equal(hx[pad:pad + n], x_ref), (cols, np.flatnonzero(hx[pad:pad + n] != x_ref)[:8])
        for k in range(2 * L):
            got = hy[pad + k * nx * ny: pad + (k + 1) * nx * ny]
            assert np.array_equal(got, y_ref[k * nx * ny:(k + 1) * nx * ny]), (cols, k, np.flatnonzero(got != y_ref[k * nx * ny:(k + 1) * nx * ny])[:8])
    hip.sync()
