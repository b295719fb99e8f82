"""Pins the CPU oracle (oracle/prost_oracle.cpp) against the reference itself.

 * golden fixtures under tests/golden/ were produced by the REAL reference code (oracle/_ref,
   compiled from /root/reference where it lies; generator: tests/golden/make_golden.py);
 * when oracle/_ref/libprost_ref.so is present (build container, or shipped to the GPU box) the
   oracle is additionally compared live on fresh random inputs.
Bar: bit-exact, except Function1DLq whose Newton/pow path depends on the libm pow overload the
host compiler picks for the reference's unqualified pow() (documented in DESIGN.md).
"""
import os

import numpy as np
import pytest

import oracle
import prost_amd as prost
from oracle import ref
from prost_amd import synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = [np.float32, np.float64]
STEPS = ["alg1", "alg2", "goldstein", "boyd"]


def close_lq(a, b, dt):
    tol = 5e-5 if dt == np.float32 else 1e-9
    fin = np.isfinite(a) & np.isfinite(b)
    # degenerate coefficients (a == 0 -> division by zero) give inf or nan depending on the libm path
    return np.array_equal(np.isfinite(a), np.isfinite(b)) and np.allclose(a[fin], b[fin], rtol=tol, atol=tol)


def test_elementwise_golden():
    g = np.load(os.path.join(GOLD, "elementwise.npz"))
    count = 32
    checked = 0
    for dt in DTYPES:
        for op, dims in ((0, (1,)), (1, (1, 2, 3, 7))):
            for dim in dims:
                for il in (False, True):
                    for inv in (False, True):
                        key = "%s_op%d_dim%d_il%d_inv%d" % (np.dtype(dt).name, op, dim, il, inv)
                        arg, td = g[key + "_arg"], g[key + "_td"]
                        coeffs = [g[key + "_c%d" % i] for i in range(5)]
                        for fn in oracle.FUNCTIONS:
                            alpha = 0.5 if fn == "lq" else 0.7
                            got = oracle.prox_elem(op, fn, arg, td, 0.8, count, dim, il, coeffs + [alpha, 1.3], inv)
                            exp = g[key + "_" + fn]
                            if fn == "lq":
                                assert close_lq(got, exp, dt), (key, fn)
                            else:
                                assert np.array_equal(got, exp, equal_nan=True), (key, fn)
                            checked += 1
    assert checked == 2 * 20 * 14      # 2 dtypes x (1 + 4) dims x 2 layouts x 2 invert flags x 14 functions


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("step", STEPS)
@pytest.mark.parametrize("res_iter", [1, 10])
def test_pdhg_iterates_golden(dtype, step, res_iter):
    g = np.load(os.path.join(GOLD, "pdhg_rof_16x12x2.npz"))
    prob, u, q, _ = synthetic.rof_problem(16, 12, 2, f=g["f"])
    prob.finalize()
    b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
    o = prost.options(max_iters=50, num_cback_calls=0, verbose=False)
    name = np.dtype(dtype).name
    for k in (1, 2, 10, 50):
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, dtype)
        s.initialize()
        s.iterate(k)
        st, sc = s.state(), s.scalars()
        key = "%s_%s_r%d_k%d" % (name, step, res_iter, k)
        for v in "xyzw":
            assert np.array_equal(st[v].astype(dtype), g[key + "_" + v]), (key, v)
        exp = g[key + "_scal"]
        got = np.array([sc[n] for n in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual")])
        # sums of squares: rocThrust's host reduce and the oracle's fold differ in the last bits, and
        # primal_var_norm is pure cancellation noise (z_hat ~ 0 for the ball projection) -> atol
        assert np.allclose(got, exp, rtol=2e-6 if dtype == np.float32 else 1e-14, atol=1e-5 if dtype == np.float32 else 1e-13), key
    sl, sr = s.problem.scaling()
    assert np.array_equal(sl, g[name + "_scaling_left"]) and np.array_equal(sr, g[name + "_scaling_right"])
    assert np.isclose(s.problem.normest(), g[name + "_normest"][0], rtol=1e-6)


@pytest.mark.parametrize("dtype,step", [(np.float32, "alg1"), (np.float32, "alg2"), (np.float32, "goldstein"), (np.float32, "boyd"),
                                        (np.float64, "alg2"), (np.float64, "boyd")])
@pytest.mark.parametrize("res_iter", [1, 10])
def test_pdhg_iterates_golden_64x64(dtype, step, res_iter):
    """64 x 64 gray-value ROF, residual_iter in {1, 10}: oracle == the REAL reference's backend_pdhg.cu
    (tests/golden/pdhg_rof_64x64.npz, made by make_golden.py from oracle/_ref), x and y bit for bit after 2, 10 and 50 iterations"""
    g = np.load(os.path.join(GOLD, "pdhg_rof_64x64.npz"))
    prob, u, q, _ = synthetic.rof_problem(64, 64, 1, f=g["f"].astype(np.float64))
    prob.finalize()
    b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
    o = prost.options(max_iters=50, num_cback_calls=0, verbose=False)
    name = np.dtype(dtype).name
    for k in (2, 10, 50):
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, dtype)
        s.initialize()
        s.iterate(k)
        st, sc = s.state(), s.scalars()
        key = "%s_%s_r%d_k%d" % (name, step, res_iter, k)
        for v in "xy":
            assert np.array_equal(st[v].astype(dtype), g[key + "_" + v]), (key, v)
        exp = g[key + "_scal"]
        got = np.array([sc[n] for n in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual")])
        assert np.allclose(got, exp, rtol=5e-6 if dtype == np.float32 else 1e-13, atol=1e-4 if dtype == np.float32 else 1e-12), (key, got, exp)


def test_misc_golden():
    g = np.load(os.path.join(GOLD, "misc.npz"))
    v, ri, cs = oracle.csr2csc(23, 31, g["csr_val"], g["csr_ind"], g["csr_ptr"])
    assert np.array_equal(v, g["csc_val"]) and np.array_equal(ri, g["csc_ind"]) and np.array_equal(cs, g["csc_ptr"])
    assert np.array_equal(oracle.linspace(0, 999, 10), g["linspace_0_999_10"])
    assert np.array_equal(oracle.linspace(0, 9999, 250), g["linspace_0_9999_250"])
    assert np.array_equal(oracle.glibc_rand(1, 64).astype(np.int64), g["glibc_rand_seed1"])
    # helper::ProjectEpiQuadNd through the oracle's epi_quad kernel with b = 0, c = 0
    for dt in DTYPES:
        name = np.dtype(dt).name
        x0, y0, al = g["epi_x0"].astype(dt), g["epi_y0"].astype(dt), g["epi_alpha"].astype(dt)
        dimx, count = x0.shape
        arg = np.concatenate([x0.reshape(-1), y0])
        got = oracle.prox_epi_quad(arg, count, dimx + 1, al, np.zeros(count * dimx), 0.0)
        tol = 1e-5 if dt == np.float32 else 1e-12
        assert np.allclose(got[:dimx * count].reshape(dimx, count), g["epi_x_" + name], rtol=tol, atol=tol)
        assert np.allclose(got[dimx * count:], g["epi_y_" + name], rtol=tol, atol=tol)


needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("dtype", DTYPES)
def test_live_elementwise_vs_reference(dtype):
    rng = np.random.default_rng(11)
    count = 301
    for op in (0, 1):
        for fn in oracle.FUNCTIONS:
            for il in (False, True):
                for inv in (False, True):
                    dim = 1 if op == 0 else 5
                    arg = rng.uniform(-4, 4, count * dim).astype(dtype)
                    td = rng.uniform(0.05, 3, count * dim).astype(dtype)
                    coeffs = [rng.uniform(0.2, 3, count), rng.uniform(-2, 2, count), rng.uniform(0.1, 3, count),
                              rng.uniform(-1, 1, count), rng.uniform(0, 2, count), 0.5 if fn == "lq" else 0.9, 0.6]
                    a = oracle.prox_elem(op, fn, arg, td, 1.7, count, dim, il, coeffs, inv)
                    b = ref.prox_elem(op, fn, arg, td, 1.7, count, dim, il, coeffs, inv)
                    if fn == "lq":
                        assert close_lq(a, b, dtype)
                    else:
                        assert np.array_equal(a, b, equal_nan=True), (fn, op, il, inv)


@needs_ref
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("step", STEPS)
def test_live_pdhg_vs_reference_backend(dtype, step):
    """reference BackendPDHG + Problem + LinearOperator + ProxMoreau, leaf kernels via plugins"""
    for shape, moreau in (((24, 20, 1), False), ((9, 14, 3), True)):
        nx, ny, L = shape
        prob, u, q, f = synthetic.rof_problem(nx, ny, L, seed=3)
        if moreau:   # hand prox_f (conjugate) instead of prox_fstar: the backend wraps it (backend_pdhg.cu:252-266)
            prob.data["prox_gstar"] = [prost.function.conjugate(lambda i, c, p=p: p)(0, 0) for p in prob.data["prox_g"]]
            prob.data["prox_g"] = []
        b = prost.backend.pdhg(stepsize=step, residual_iter=4, alg2_gamma=0.3)
        x0 = np.linspace(0, 1, prob.ncols); y0 = np.linspace(-0.5, 0.5, prob.nrows)
        o = prost.options(max_iters=30, num_cback_calls=0, verbose=False, x0=x0, y0=y0)
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, dtype)
        s.initialize(); s.iterate(30)
        st = s.state()
        r = ref.RefProblem(prob.data, prob.nrows, prob.ncols, dtype).pdhg(b[1], o, 30)
        for v in "xyzw":
            assert np.array_equal(st[v], r[v]), (shape, v, np.abs(st[v] - r[v]).max())


@needs_ref
def test_live_solve_dual_vs_reference():
    prob, u, q, f = synthetic.rof_problem(12, 10, 1, seed=4)
    b = prost.backend.pdhg(stepsize="alg1", residual_iter=2)
    o = prost.options(max_iters=20, num_cback_calls=0, verbose=False, solve_dual=True)
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float32)
    s.initialize(); s.iterate(20)
    st = s.state()
    r = ref.RefProblem(prob.data, prob.nrows, prob.ncols, np.float32).pdhg(b[1], o, 20)
    for v in "xyzw":
        assert np.array_equal(st[v], r[v]), v
