"""Fused gradient3d passes (BASELINE config 3: volumetric TV) against the CPU oracle: kernel level
(one pass == the reference's unfused sequence, bit-exact) and solver level (PDHG iterates)."""
import ctypes as C

import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import synthetic

pytestmark = pytest.mark.gpu
DTYPES = [np.float32, np.float64]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(6, 8, 1), (5, 12, 4), (20, 1028, 3), (9, 7, 2), (33, 64, 5)])
@pytest.mark.parametrize("fns", [("square", "ind_leq0"), ("abs", "huber")])
def test_fused3d_passes_match_unfused_oracle(hip, dtype, shape, fns):
    nx, ny, L = shape
    g_fn, f_fn = fns
    rng = np.random.default_rng(3)
    n, m = nx * ny * L, 3 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    y_prev = rng.uniform(-1, 1, m).astype(dtype); x_old = rng.uniform(0, 1, n).astype(dtype)
    f = rng.uniform(0, 1, n)
    tau, sigma, theta = dtype(0.9), dtype(1.1), dtype(0.85)
    Tval, Sval = dtype(1.0 / 6.0), dtype(0.5)
    g_coeffs = [1.0, f, 10.0, 0.0, 0.0, 0.3, 0.0]
    f_coeffs = [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0]
    d = hip.FusedDesc(); d.is3d = 1; d.nx, d.ny, d.L = nx, ny, L
    d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID[f_fn]
    gp, gv, k1 = hip.coeff_args(g_coeffs, dtype, n)
    fp, fv, k2 = hip.coeff_args(f_coeffs, dtype, n)
    for i in range(7):
        d.g_coeff_ptr[i] = gp[i]; d.g_coeff_val[i] = gv[i]; d.f_coeff_ptr[i] = fp[i]; d.f_coeff_val[i] = fv[i]
    d.T_val, d.S_val = float(Tval), float(Sval)
    assert hip.lib().prost_hip_fused_supported(C.byref(d), 0) == 1
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    Td, Sd = np.full(n, Tval, dtype), np.full(m, Sval, dtype)
    dx, dy, dyp, dxo = (hip.DeviceArray.from_host(v) for v in (x, y, y_prev, x_old))
    for use_kty, use_prev in ((1, 1), (0, 0), (1, 0)):
        kty = oracle.grad3d(y, nx, ny, L, adjoint=True) if use_kty else np.zeros(n, dtype)
        ktyp = oracle.grad3d(y_prev, nx, ny, L, adjoint=True) if use_prev else np.zeros(n, dtype)
        temp = (x - tau * Td * kty).astype(dtype)
        x_ref = oracle.prox_elem(0, g_fn, temp, Td, tau, n, 1, False, g_coeffs)
        sq = np.sqrt(Td)
        w_hat = ((x - x_ref) / (tau * sq) - sq * ktyp).astype(dtype); diff = (w_hat + sq * kty).astype(dtype)
        dres = np.array([np.sum((diff * diff).astype(np.float64)), np.sum((w_hat * w_hat).astype(np.float64))])
        x_new = hip.DeviceArray.zeros(n, dtype); out2 = hip.DeviceArray.zeros(2, np.float64)
        hip.check(hip.fn("fused_primal", dtype)(C.byref(d), x_new.ptr, dx.ptr, dy.ptr, dyp.ptr, hip.dbl(tau), use_kty, use_prev, out2.ptr, ws.ptr, None))
        assert np.array_equal(x_new.to_host(), x_ref)
        assert np.allclose(out2.to_host(), dres, rtol=1e-11)
    for use_kxp in (1, 0):
        kx = oracle.grad3d(x, nx, ny, L)
        kxp = oracle.grad3d(x_old, nx, ny, L) if use_kxp else np.zeros(m, dtype)
        temp = (y + sigma * Sd * ((1 + theta) * kx - theta * kxp)).astype(dtype)
        y_ref = oracle.prox_elem(1, f_fn, temp, Sd, sigma, n, 3, False, f_coeffs)
        sq = np.sqrt(Sd)
        z_hat = ((y - y_ref) / (sigma * sq) + sq * ((1 + theta) * kx - theta * kxp)).astype(dtype); diff = (z_hat - sq * kx).astype(dtype)
        pres = np.array([np.sum((diff * diff).astype(np.float64)), np.sum((z_hat * z_hat).astype(np.float64))])
        y_new = hip.DeviceArray.zeros(m, dtype); out2 = hip.DeviceArray.zeros(2, np.float64)
        hip.check(hip.fn("fused_dual", dtype)(C.byref(d), y_new.ptr, dy.ptr, dx.ptr, dxo.ptr, hip.dbl(sigma), hip.dbl(theta), use_kxp, out2.ptr, ws.ptr, None))
        assert np.array_equal(y_new.to_host(), y_ref)
        assert np.allclose(out2.to_host(), pres, rtol=1e-11)
    hip.sync()


@pytest.mark.parametrize("precision,dtype", [("single", np.float32), ("double", np.float64)])
@pytest.mark.parametrize("step", ["alg2", "boyd"])
def test_tv3d_pdhg_iterates_match_oracle(hip, precision, dtype, step):
    prost.set_gpu(0)
    prost.set_precision(precision)
    try:
        for (nx, ny, L) in ((12, 16, 5), (7, 1028, 2)):
            prob, u, q, f = synthetic.tv3d_problem(nx, ny, L, seed=2)
            for fused, single in ((True, True), (True, False), (False, False)):
                # fused + single: one kernel per non-residual iteration (x_new of plane l+1 recomputed), two passes on residual
                # iterations; fused only: two passes always; neither: the generic nine-vector path
                b = prost.backend.pdhg(stepsize=step, residual_iter=3, alg2_gamma=0.5)
                b[1]["allow_fused"] = fused
                b[1]["allow_single_kernel"] = single
                o = prost.options(max_iters=40, num_cback_calls=0, verbose=False)
                s = prost.Solver(prob, b, o); s.iterate(40); st = s.state(); s.destroy()
                assert st["path"] == ("pdhg:fused-grad3d" if fused else "pdhg:generic")
                bo = prost.backend.pdhg(stepsize=step, residual_iter=3, alg2_gamma=0.5)
                so = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, dtype); so.initialize(); so.iterate(40)
                ost = so.state()
                for v in "xyzw":
                    assert np.array_equal(st[v], ost[v]), (v, fused, float(np.abs(st[v] - ost[v]).max()))
    finally:
        prost.set_precision("double")


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
@pytest.mark.parametrize("step", ["alg1", "alg2", "boyd", "goldstein"])
@pytest.mark.parametrize("residual_iter,data_term", [(1, "square"), (3, "square"), (4, "square"), (5, "square"), (10, "square"), (10, "abs"), (3, "abs")])
def test_tv3d_pair_schedule_is_invisible(hip, prec, dtype, step, residual_iter, data_term):
    """Volumetric TV (fp32 and fp64) with two iterations per launch (prost_hip_fused_iteration3d_x2) wherever neither k nor k+2 is a
    residual iteration (k+1 may be one: the kernel forms its sums): the state after ANY number of iterations -- x, y, the constraint variables z, w (which need the
    previous iterate, rebuilt by one single launch after a pair), residuals, step sizes -- is bit-identical to the path
    that launches every iteration separately, and the iterates equal the oracle's."""
    prost.set_gpu(0)
    prost.set_precision(prec)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    try:
        for (nx, ny, L) in ((12, 16, 5), (9, 250, 14), (6, 128, 30), (7, 67, 6)):
            prob, u, q, f = synthetic.tv3d_problem(nx, ny, L, seed=2, data_term=data_term, lmb=10.0 if data_term == "square" else 0.7)
            for iters in (2, 3, 4, 5, 9, 10, 11, 23):
                states = []
                for pair in (True, False):
                    b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.5)
                    b[1]["allow_pair_kernel"] = pair
                    s = prost.Solver(prob, b, o)
                    info = s.iterate(iters, time_kernels=True, sample_every=1)
                    names = list(info["kernels"])
                    st = s.state()
                    s.iterate(7)              # a second batch: the pairing restarts from another offset
                    st2 = s.state()
                    s.destroy()
                    assert st["path"] == "pdhg:fused-grad3d"
                    # a pair starts at k >= 2 unless k or k + 2 is a residual iteration
                    x2 = [k for k in names if k.startswith("fused_iter3d_x2_kernel")]
                    if residual_iter >= 3 and iters >= 8:
                        assert bool(x2) == pair, (names, pair)
                    elif not pair:
                        assert not x2
                    states.append((st, st2))
                for a_, b_ in zip(states[0], states[1]):
                    for v in "xyzw":
                        assert np.array_equal(a_[v], b_[v]), (nx, ny, L, iters, v)
                    for v in ("tau", "sigma", "theta", "iteration"):
                        assert a_[v] == b_[v], (nx, ny, L, iters, v, a_[v], b_[v])
                    for v in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm"):       # same terms, another summation order
                        assert np.isclose(a_[v], b_[v], rtol=1e-9, atol=0), (nx, ny, L, iters, v, a_[v], b_[v])
            bo = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.5)
            so = oracle.Solver(prob.data, prob.nrows, prob.ncols, bo, o, dtype); so.initialize(); so.iterate(23)
            ost = so.state()
            s = prost.Solver(prob, bo, o); s.iterate(23); st = s.state(); s.destroy()
            for v in "xyzw":
                assert np.array_equal(st[v], ost[v]), (nx, ny, L, v)
    finally:
        prost.set_precision("double")


def test_tv3d_large_fused_equals_generic(hip):
    """512 x 512 x 32 (8.4 M voxels, the C3 shape scaled to what the host can read back): the fused gradient3d
    passes and the generic nine-vector path are two independent kernel paths -- identical bits after 12 iterations."""
    prost.set_gpu(0)
    prost.set_precision("single")
    try:
        prob, u, q, f = synthetic.tv3d_problem(512, 512, 32, seed=4)
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
        st = {}
        for fused in (True, False):
            b = prost.backend.pdhg(stepsize="alg2", residual_iter=4, alg2_gamma=0.5)
            b[1]["allow_fused"] = fused           # True: one kernel per non-residual iteration, two passes on residual iterations
            s = prost.Solver(prob, b, o); s.iterate(12); st[fused] = s.state(); s.destroy()
        assert st[True]["path"] == "pdhg:fused-grad3d" and st[False]["path"] == "pdhg:generic"
        for v in "xyzw":
            assert np.array_equal(st[True][v], st[False][v]), v
        assert np.isfinite(st[True]["x"]).all()
        assert np.isclose(st[True]["primal_res"], st[False]["primal_res"], rtol=1e-5)
    finally:
        prost.set_precision("double")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(6, 8, 1), (5, 12, 4), (20, 1028, 3), (33, 64, 5), (3, 256, 2), (40, 508, 3), (7, 16, 9), (5, 24, 15)])
@pytest.mark.parametrize("fns", [("square", "ind_leq0"), ("abs", "huber")])
@pytest.mark.parametrize("vector_b", [True, False])
def test_single_kernel_3d_iteration_equals_two_passes(hip, dtype, shape, fns, vector_b):
    """prost_hip_fused_iteration3d (x_new of plane l+1 recomputed in registers) against the two-pass kernels, which are
    pinned to the oracle above: same bits for x_new and all three components of y_new, for every chunk width,
    flag combination, plane count (incl. L = 1: no upper plane) and strip layout (1028 rows: five wavefront strips)."""
    nx, ny, L = shape
    g_fn, f_fn = fns
    rng = np.random.default_rng(5)
    n, m = nx * ny * L, 3 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    tau, sigma, theta = dtype(0.9), dtype(1.1), dtype(0.85)
    g_coeffs = [1.0, f if vector_b else 0.4, 10.0, 0.0, 0.0, 0.3, 0.0]
    f_coeffs = [1.0, 1.0, 1.0, 0.0, 0.0, 0.3, 0.0]
    d = hip.FusedDesc(); d.is3d = 1; d.nx, d.ny, d.L = nx, ny, L
    d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID[f_fn]
    gp, gv, k1 = hip.coeff_args(g_coeffs, dtype, n)
    fp, fv, k2 = hip.coeff_args(f_coeffs, dtype, n)
    for i in range(7):
        d.g_coeff_ptr[i] = gp[i]; d.g_coeff_val[i] = gv[i]; d.f_coeff_ptr[i] = fp[i]; d.f_coeff_val[i] = fv[i]
    d.T_val, d.S_val = 1.0 / 6.0, 0.5
    dt = 0 if dtype == np.float32 else 1
    vecw = 4 if dtype == np.float32 else 2
    assert hip.lib().prost_hip_fused_iteration3d_supported(C.byref(d), dt) == (1 if ny % vecw == 0 else 0)
    if ny % vecw:
        return
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
                hip.check(hip.fn("fused_iteration3d", dtype)(C.byref(d), x_new.ptr, y_new.ptr, dx.ptr, dy.ptr, dyp.ptr if res else None, hip.dbl(tau), hip.dbl(sigma),
                                                             hip.dbl(theta), use_kty, use_kxp, use_ktyp, cols, r4.ptr if res else None, ws.ptr if res else None, None))
                assert np.array_equal(x_new.to_host(), x_ref.to_host()), (cols, use_kty, res)
                assert np.array_equal(y_new.to_host(), y_ref.to_host()), (cols, use_kty, use_kxp, res)
                if res:
                    assert np.allclose(r4.to_host(), res_ref, rtol=1e-11, atol=1e-300), (cols, r4.to_host(), res_ref)
            # planes across the wavefronts of a workgroup (x_new exchanged through LDS, one helper wavefront per group)
            if L >= 2:
                assert hip.lib().prost_hip_fused_iteration3d_pw_supported(C.byref(d), dt) == 1
                for waves in (4, 8):
                    x_new = hip.DeviceArray.zeros(n, dtype); y_new = hip.DeviceArray.zeros(m, dtype)
                    hip.check(hip.fn("fused_iteration3d_pw", dtype)(C.byref(d), x_new.ptr, y_new.ptr, dx.ptr, dy.ptr, hip.dbl(tau), hip.dbl(sigma), hip.dbl(theta),
                                                                    use_kty, use_kxp, cols, waves, None))
                    assert np.array_equal(x_new.to_host(), x_ref.to_host()), ("pw", waves, cols, use_kty)
                    assert np.array_equal(y_new.to_host(), y_ref.to_host()), ("pw", waves, cols, use_kty, use_kxp)
    hip.sync()


@pytest.mark.parametrize("shape", [(6, 8, 1), (5, 12, 4), (20, 1028, 3), (33, 64, 5), (4, 256, 2), (40, 508, 6), (7, 16, 9), (5, 24, 15), (64, 252, 11), (9, 248, 7), (9, 250, 14), (5, 6, 29), (12, 130, 13), (70, 126, 27), (9, 63, 6), (8, 125, 14)])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("vector_b", [True, False])
@pytest.mark.parametrize("radius,g_fn", [(1.0, "square"), (1e-7, "square"), (1.0, "abs")])
def test_double_3d_iteration_equals_two_single_launches(hip, dtype, shape, vector_b, radius, g_fn):
    """prost_hip_fused_iteration3d_x2 (two iterations per launch, planes across wavefronts, stages meeting in LDS) against two
    iterations of the two-pass kernels, which are pinned to the oracle above: same bits for
    x^(k+2) and all three components of y^(k+2), for every chunk width (1: every column a chunk border; 64: one chunk), plane
    counts below / equal to / above the 13 planes a workgroup owns (helper planes outside the volume, several plane groups),
    strip layouts (126 = one strip + 2 rows, 1028 rows: nine strips) and step sizes that change between the two iterations (alg2);
    with residual sums: the four sums of the second iteration against the two-pass kernels' for that iteration."""
    nx, ny, L = shape
    dt = 0 if dtype == np.float32 else 1
    rng = np.random.default_rng(11)
    n, m = nx * ny * L, 3 * nx * ny * L
    x = rng.uniform(0, 1, n).astype(dtype); y = rng.uniform(-1, 1, m).astype(dtype)
    f = rng.uniform(0, 1, n)
    g_coeffs = [1.0, f if vector_b else 0.4, 10.0, 0.0, 0.0, 0.3, 0.0]
    f_coeffs = [1.0, radius, 1.0, 0.0, 0.0, 0.3, 0.0]
    d = hip.FusedDesc(); d.is3d = 1; d.nx, d.ny, d.L = nx, ny, L
    d.g_fn = hip.FN_ID[g_fn]; d.f_fn = hip.FN_ID["ind_leq0"]          # abs: the TV-L1 data term (soft threshold around b)
    gp, gv, k1 = hip.coeff_args(g_coeffs, dtype, n)
    fp, fv, k2 = hip.coeff_args(f_coeffs, dtype, n)
    for i in range(7):
        d.g_coeff_ptr[i] = gp[i]; d.g_coeff_val[i] = gv[i]; d.f_coeff_ptr[i] = fp[i]; d.f_coeff_val[i] = fv[i]
    d.T_val, d.S_val = 1.0 / 6.0, 0.5
    assert hip.lib().prost_hip_fused_iteration3d_x2_supported(C.byref(d), dt) == 1     # any height: 2 floats per lane where it is even, else 1 row per lane
    tau, sigma, theta = [0.9, 0.61], [1.1, 1.63], [0.85, 0.67]
    dx, dy = hip.DeviceArray.from_host(x), hip.DeviceArray.from_host(y)
    x1 = hip.DeviceArray.zeros(n, dtype); y1 = hip.DeviceArray.zeros(m, dtype)
    x2 = hip.DeviceArray.zeros(n, dtype); y2 = hip.DeviceArray.zeros(m, dtype)
    P, D = hip.fn("fused_primal", dtype), hip.fn("fused_dual", dtype)          # the two-pass kernels (any height), pinned to the oracle above
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    hip.check(P(C.byref(d), x1.ptr, dx.ptr, dy.ptr, None, hip.dbl(tau[0]), 1, 0, None, ws.ptr, None))
    hip.check(D(C.byref(d), y1.ptr, dy.ptr, x1.ptr, dx.ptr, hip.dbl(sigma[0]), hip.dbl(theta[0]), 1, None, ws.ptr, None))
    rd = hip.DeviceArray.zeros(2, np.float64); rp = hip.DeviceArray.zeros(2, np.float64)      # the second iteration as a residual iteration (y_prev = y^k)
    hip.check(P(C.byref(d), x2.ptr, x1.ptr, y1.ptr, dy.ptr, hip.dbl(tau[1]), 1, 1, rd.ptr, ws.ptr, None))
    hip.check(D(C.byref(d), y2.ptr, y1.ptr, x2.ptr, x1.ptr, hip.dbl(sigma[1]), hip.dbl(theta[1]), 1, rp.ptr, ws.ptr, None))
    res_ref = np.concatenate([rp.to_host(), rd.to_host()])      # {primal diff^2, primal var^2, dual diff^2, dual var^2}
    x_ref, y_ref = x2.to_host(), y2.to_host()
    arr = lambda v: (C.c_double * 2)(*v)
    for cols in (0, 1, 2, 5, 7, 64):
        for res in (False, True):
            xo = hip.DeviceArray.zeros(n, dtype); yo = hip.DeviceArray.zeros(m, dtype); r4 = hip.DeviceArray.zeros(4, np.float64)
            hip.check(hip.fn("fused_iteration3d_x2", dtype)(C.byref(d), xo.ptr, yo.ptr, dx.ptr, dy.ptr, arr(tau), arr(sigma), arr(theta), cols,
                                                                   r4.ptr if res else None, ws.ptr if res else None, None))
            hx, hy = xo.to_host(), yo.to_host()
            assert np.array_equal(hx, x_ref), (cols, res, np.flatnonzero(hx != x_ref)[:8])
            for k in range(3):
                assert np.array_equal(hy[k * n:(k + 1) * n], y_ref[k * n:(k + 1) * n]), (cols, res, k, np.flatnonzero(hy[k * n:(k + 1) * n] != y_ref[k * n:(k + 1) * n])[:8])
            if res:          # the same terms as the two-pass kernels, summed in double in another order
                assert np.allclose(r4.to_host(), res_ref, rtol=1e-11, atol=1e-300), (cols, r4.to_host(), res_ref)
    hip.sync()


@pytest.mark.parametrize("kernel", ["fused_iteration3d", "fused_iteration3d_pw", "fused_iteration3d_x2", "fused_iteration_mc"])
def test_single_kernel_iterations_write_only_their_outputs(hip, kernel):
    """Canary words around the output vectors of the one-kernel iterations (halo lanes, halo columns, helper and idle
    wavefronts must not store anything): shapes whose last row strip, last column chunk and last plane group are
    partial."""
    dtype = np.float32
    rng = np.random.default_rng(17)
    pad, sentinel = 256, np.float32(-123456.75)
    for (nx, ny, L) in ((11, 260, 5), (7, 1028, 2), (19, 16, 4), (5, 508, 3)):
        is3d = kernel != "fused_iteration_mc"
        if not is3d and L not in (3, 4):
            continue
        comps = 3 if is3d else 2
        n, m = nx * ny * L, comps * nx * ny * L
        d = hip.FusedDesc(); d.is3d = 1 if is3d else 0; d.nx, d.ny, d.L = nx, ny, L
        d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
        f = hip.DeviceArray.from_host(rng.uniform(0, 1, n).astype(dtype))
        gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
        for i in range(7):
            d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
        d.g_coeff_ptr[1] = f.ptr.value
        d.T_val, d.S_val = (1.0 / 6.0 if is3d else 0.25), 0.5
        dx = hip.DeviceArray.from_host(rng.uniform(0, 1, n).astype(dtype)); dy = hip.DeviceArray.from_host(rng.uniform(-1, 1, m).astype(dtype))
        for cols in (0, 3):
            bx = hip.DeviceArray.from_host(np.full(n + 2 * pad, sentinel, dtype)); by = hip.DeviceArray.from_host(np.full(m + 2 * pad, sentinel, dtype))
            xo = C.c_void_p(bx.ptr.value + pad * 4); yo = C.c_void_p(by.ptr.value + pad * 4)
            fn = hip.fn(kernel, dtype)
            if kernel == "fused_iteration3d":
                hip.check(fn(C.byref(d), xo, yo, dx.ptr, dy.ptr, None, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 1, cols, None, None, None))
            elif kernel == "fused_iteration3d_pw":
                hip.check(fn(C.byref(d), xo, yo, dx.ptr, dy.ptr, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, cols, 0, None))
            elif kernel == "fused_iteration3d_x2":
                two = lambda a, b: (C.c_double * 2)(a, b)
                hip.check(fn(C.byref(d), xo, yo, dx.ptr, dy.ptr, two(0.3, 0.25), two(1.0, 1.2), two(0.9, 0.8), cols, None, None, None))
            else:
                hip.check(fn(C.byref(d), xo, yo, dx.ptr, dy.ptr, None, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 1, cols, None, None, None))
            hx, hy = bx.to_host(), by.to_host()
            for h, k in ((hx, n), (hy, m)):
                assert np.all(h[:pad] == sentinel) and np.all(h[pad + k:] == sentinel), (kernel, nx, ny, L, cols)
                assert np.all(h[pad:pad + k] != sentinel)          # and every output element was written
