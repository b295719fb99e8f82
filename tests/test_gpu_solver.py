"""GPU parity tests of the PRODUCT path: Python front-end -> prost_command (libprost.so, host C++)
-> kernel C ABI (libprost_hip.so) -> MI355X, against the CPU oracle, the golden fixtures produced
by the real reference, and scipy restatements of the reference's own MATLAB tests.

Bar: PDHG iterates bit-exact with the oracle (both sides evaluate the reference's expressions
without FMA contraction; HIP fp32 divide/sqrt are correctly rounded).  Residual scalars: rel 1e-5
(the reference reduces in T with an unspecified order; the kernels accumulate in double).
ADMM: rel 1e-4 on the iterates after 20 iterations (CG step lengths come from reductions).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import oracle
import prost_amd as prost
from prost_amd import synthetic
from reference_matrices import spdiags_const, spmat_gradient2d, spmat_gradient3d

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PRECISIONS = [("single", np.float32), ("double", np.float64)]
STEPS = ["alg1", "alg2", "goldstein", "boyd"]


@pytest.fixture(autouse=True)
def _gpu(hip):
    prost.set_gpu(0)
    yield
    prost.set_precision("double")


def run_product(prob, backend, opts, iters):
    s = prost.Solver(prob, backend, opts)
    s.iterate(iters)
    st = s.state()
    s.destroy()
    return st


def run_oracle(prob, backend, opts, iters, dtype):
    prob.finalize()
    b = [backend[0], {k: v for k, v in backend[1].items() if k not in ("allow_fused", "device_cg", "allow_arg_fusion", "cg_graph", "fused_rounds", "pixel_rounds", "allow_speculation", "allow_pair_kernel", "allow_device_rules")}]
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, opts, dtype)
    s.initialize()
    s.iterate(iters)
    st = s.state()
    st.update(s.scalars())
    return st


def assert_same_iterates(st, ost, exact=True, tol=0.0):
    for v in "xyzw":
        if exact:
            assert np.array_equal(st[v], ost[v]), (v, float(np.abs(st[v] - ost[v]).max()))
        else:
            scale = max(1.0, float(np.abs(ost[v]).max()))
            assert float(np.abs(st[v] - ost[v]).max()) <= tol * scale, (v, float(np.abs(st[v] - ost[v]).max()))


# ---------------------------------------------------------------------------------------------
# PDHG
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", STEPS)
@pytest.mark.parametrize("fused", [True, False])
def test_pdhg_iterates_match_oracle(precision, dtype, step, fused):
    prost.set_precision(precision)
    for (nx, ny, L), res_iter in (((40, 64, 2), 3), ((33, 30, 1), 1), ((16, 1028, 1), 10)):
        prob, u, q, f = synthetic.rof_problem(nx, ny, L, seed=5)
        b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
        b[1]["allow_fused"] = fused
        o = prost.options(max_iters=60, num_cback_calls=0, verbose=False)
        for k in (1, 2, 11, 50):
            st = run_product(prob, b, o, k)
            assert st["path"] == ("pdhg:fused-grad2d" if fused else "pdhg:generic")
            ost = run_oracle(prob, b, o, k, dtype)
            assert_same_iterates(st, ost)
            for name in ("tau", "sigma", "theta"):
                assert st[name] == ost[name], name
            for name in ("primal_res", "dual_res", "dual_var_norm", "eps_primal", "eps_dual"):
                assert np.isclose(st[name], ost[name], rtol=1e-5, atol=1e-6), (name, st[name], ost[name])


@pytest.mark.parametrize("precision,dtype,step", [("single", np.float32, "alg1"), ("single", np.float32, "alg2"), ("single", np.float32, "goldstein"),
                                                  ("single", np.float32, "boyd"), ("double", np.float64, "alg2"), ("double", np.float64, "boyd")])
def test_pdhg_matches_reference_golden_fixture_64x64(precision, dtype, step):
    """product (pair / single launches of the gray-value kernels; goldstein / boyd with the rule on the device) vs
    tests/golden/pdhg_rof_64x64.npz: x and y of the REAL reference backend after 2, 10 and 50 iterations, residual_iter 1 and 10"""
    prost.set_precision(precision)
    g = np.load(os.path.join(GOLD, "pdhg_rof_64x64.npz"))
    name = np.dtype(dtype).name
    for res_iter in (1, 10):
        prob, u, q, _ = synthetic.rof_problem(64, 64, 1, f=g["f"].astype(np.float64))
        b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
        o = prost.options(max_iters=50, num_cback_calls=0, verbose=False)
        for k in (2, 10, 50):
            st = run_product(prob, b, o, k)
            key = "%s_%s_r%d_k%d" % (name, step, res_iter, k)
            for v in "xy":
                assert np.array_equal(st[v].astype(dtype), g[key + "_" + v]), (key, v)
            exp = g[key + "_scal"]
            got = np.array([st[n] for n in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual")])
            assert np.allclose(got, exp, rtol=1e-5 if dtype == np.float32 else 1e-12, atol=1e-4 if dtype == np.float32 else 1e-11), (key, got, exp)


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("single", [True, False])
def test_pdhg_multichannel_iterates_match_oracle(precision, dtype, single):
    """RGB / 4-channel ROF (the shape of example_rof_primaldual.m: sum_norm2(2 * nc, ...)).  single = True: one kernel per
    non-residual iteration with the channels on the wavefronts of a workgroup (kernels_fused_iter_mc.hip), two passes on
    residual iterations; False: two passes always.  Bit for bit against the oracle."""
    prost.set_precision(precision)
    for (nx, ny, L), res_iter in (((24, 16, 3), 3), ((9, 260, 4), 1), ((13, 1028, 3), 10)):
        prob, u, q, f = synthetic.rof_problem(nx, ny, L, seed=9)
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=res_iter, alg2_gamma=0.5)
        b[1]["allow_single_kernel"] = single
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
        st = run_product(prob, b, o, 25)
        assert st["path"] == "pdhg:fused-grad2d"
        bo = prost.backend.pdhg(stepsize="alg2", residual_iter=res_iter, alg2_gamma=0.5)
        assert_same_iterates(st, run_oracle(prob, bo, o, 25, dtype))


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", STEPS)
def test_pdhg_matches_reference_golden_fixture(precision, dtype, step):
    """product vs tests/golden/pdhg_rof_16x12x2.npz (iterates of the REAL reference backend)"""
    prost.set_precision(precision)
    g = np.load(os.path.join(GOLD, "pdhg_rof_16x12x2.npz"))
    name = np.dtype(dtype).name
    for res_iter in (1, 10):
        prob, u, q, _ = synthetic.rof_problem(16, 12, 2, f=g["f"])
        b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
        o = prost.options(max_iters=50, num_cback_calls=0, verbose=False)
        for k in (1, 2, 10, 50):
            st = run_product(prob, b, o, k)
            key = "%s_%s_r%d_k%d" % (name, step, res_iter, k)
            for v in "xyzw":
                assert np.array_equal(st[v].astype(dtype), g[key + "_" + v]), (key, v)
            exp = g[key + "_scal"]
            got = np.array([st[n] for n in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual")])
            assert np.allclose(got, exp, rtol=1e-5, atol=1e-5 if dtype == np.float32 else 1e-12), key


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
def test_pdhg_warm_start_moreau_and_dual(precision, dtype):
    prost.set_precision(precision)
    # warm start: the reference's first iteration ignores K^T y0 and K x0 (kty_, kx_prev_ start as zero vectors)
    prob, u, q, f = synthetic.rof_problem(24, 20, 1, seed=3)
    x0, y0 = np.linspace(0, 1, prob.ncols), np.linspace(-0.5, 0.5, prob.nrows)
    for fused in (True, False):
        b = prost.backend.pdhg(stepsize="boyd", residual_iter=4)
        b[1]["allow_fused"] = fused
        o = prost.options(max_iters=30, num_cback_calls=0, verbose=False, x0=x0, y0=y0)
        for k in (1, 2, 3, 30):
            assert_same_iterates(run_product(prob, b, o, k), run_oracle(prob, b, o, k, dtype))
    # prox_gstar given (conjugate): backend wraps it by Moreau (backend_pdhg.cu:236-250) -> generic path
    prob, u, q, f = synthetic.rof_problem(9, 14, 3, seed=3)
    prob.data["prox_gstar"] = [prost.function.conjugate(lambda i, c, p=p: p)(0, 0) for p in prob.data["prox_g"]]
    prob.data["prox_g"] = []
    prob.finalize = lambda: prob      # MATLAB's finalize would add a zero prox_g next to prox_gstar (min_max_problem.m:217-227)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=4, alg2_gamma=0.3)
    o = prost.options(max_iters=30, num_cback_calls=0, verbose=False)
    st = run_product(prob, b, o, 30)
    assert st["path"] == "pdhg:generic"
    assert_same_iterates(st, run_oracle(prob, b, o, 30, dtype))
    # solve_dual (Problem::Dualize, DualLinearOperator); exact negate on both sides
    prob, u, q, f = synthetic.rof_problem(12, 10, 1, seed=4)
    b = prost.backend.pdhg(stepsize="alg1", residual_iter=2)
    o = prost.options(max_iters=20, num_cback_calls=0, verbose=False, solve_dual=True)
    if precision == "single":          # thrust::negate<float> is exact for T = float
        assert_same_iterates(run_product(prob, b, o, 20), run_oracle(prob, b, o, 20, dtype))


def test_solve_pairs_iterations_and_polls_the_stopping_callback_once_per_launch():
    """prost.solve launches two iterations at once where nobody looks in between (row a1: the budget of Solver::Solve).  A
    registered stopping callback (the MEX gateway's Ctrl-C poll, solver.cu:151) is asked once per launch -- after every
    iteration or every second one -- and ends the run there, with the iterates the uninterrupted run has at that iteration."""
    prost.set_precision("single")
    prob, u, q, f = synthetic.rof_problem(64, 48, 1, seed=8)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    kw = dict(verbose=False, num_cback_calls=0, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    r = prost.solve(prob, b, prost.options(max_iters=40, **kw))
    assert r["result"] == "Reached maximum iterations." and int(r["pair_launches"]) >= 15, r["pair_launches"]
    polls = []
    prost.set_stop_callback(lambda: polls.append(1) or len(polls) >= 12)
    try:
        rs = prost.solve(prob, b, prost.options(max_iters=40, **kw))
    finally:
        prost.set_stop_callback(None)
    k = int(rs["iters"])
    assert rs["result"] == "Stopped by user." and len(polls) == 12
    assert 12 <= k <= 24 and int(rs["pair_launches"]) == k - 12, (k, rs["pair_launches"])       # one poll per launch: k iterations in 12 launches
    rk = prost.solve(prob, b, prost.options(max_iters=k, **kw))
    for v in "xyzw":
        assert np.array_equal(np.asarray(rs[v]), np.asarray(rk[v])), v
    r2 = prost.solve(prob, b, prost.options(max_iters=40, **kw))          # the callback is gone again
    assert int(r2["pair_launches"]) == int(r["pair_launches"]) and np.array_equal(np.asarray(r2["x"]), np.asarray(r["x"]))


def test_verbose_output_reaches_the_front_end_print_function():
    """prost_set_output_callback (the MEX gateway's std::cout -> mexPrintf redirect, prost.cpp:15-44): a verbose solve hands
    its header, the scheduled "It k: Feas_p=..." lines (solver.cu:161-171: num_cback_calls + 1 of them, scientific with two
    digits) and the closing line to the registered function; afterwards the library prints to stdout again."""
    import re
    prost.set_precision("single")
    prob, u, q, f = synthetic.rof_problem(48, 40, 1, seed=8)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=5, alg2_gamma=0.5)
    o = prost.options(max_iters=100, num_cback_calls=4, verbose=True, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    chunks = []
    prost.set_output_callback(chunks.append)
    try:
        r = prost.solve(prob, b, o)
    finally:
        prost.set_output_callback(None)
    text = "".join(chunks)
    assert r["result"] == "Reached maximum iterations."
    assert text.startswith("prost v") and "# primal variables: %d" % (48 * 40) in text and "# dual variables: %d" % (2 * 48 * 40) in text
    lines = [l for l in text.splitlines() if l.startswith("It ")]
    assert [int(l[3:6]) for l in lines] == [1, 34, 67, 100], lines                   # common.cu:33-46: linspace(0, max_iters - 1, 4) = 0, 33, 66, 99 (, 99)
    for l in lines:
        assert re.fullmatch(r"It +\d+: Feas_p=\d\.\d\de[+-]\d\d, Eps_p=\d\.\d\de[+-]\d\d, Feas_d=\d\.\d\de[+-]\d\d, Eps_d=\d\.\d\de[+-]\d\d; ", l), l
    assert text.rstrip().endswith("Reached maximum of 100 iterations.")
    n = len(chunks)
    prost.solve(prob, b, prost.options(max_iters=10, num_cback_calls=0, verbose=False))
    assert len(chunks) == n


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["alg2", "alg1"])
@pytest.mark.parametrize("residual_iter", [4, 5, 10])
@pytest.mark.parametrize("with_comm", [False, True])
def test_speculative_next_launch_is_invisible(precision, dtype, step, residual_iter, with_comm):
    """With alg1 / alg2 the pair launch that follows a residual iteration is enqueued BEFORE the host waits for the residual sums
    (into spare buffers) and adopted by a buffer exchange if the solver goes on, forgotten otherwise.  Whatever the caller does in
    between -- checked iteration (the loop of prost.solve), reading the state (z, w need the rebuilt previous iterate, which uses
    the same spare buffers), unchecked iteration, a solve that stops on its tolerance -- the results equal those of a solver that
    never speculates, and the oracle's.  with_comm: a (one-rank, host-callback) communicator is attached -- the sums are all-reduced on
    the side stream and the speculative pair runs beside the collective instead of beside the host's look at the sums."""
    prost.set_precision(precision)
    if with_comm:
        prost.comm_init_host(lambda a: None, 1)
    try:
        _speculation_body(dtype, step, residual_iter)
    finally:
        if with_comm:
            prost.comm_destroy()
        prost.set_precision("double")


def _speculation_body(dtype, step, residual_iter):
    prob, u, q, f = synthetic.rof_problem(44, 252, 1, seed=9)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    runs = {}
    for spec in (True, False):
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.4)
        b[1]["allow_speculation"] = spec
        s = prost.Solver(prob, b, o)
        trace = []
        s.iterate(residual_iter + 1, checked=True)               # ends right after a residual iteration: a speculative pair is in flight
        trace.append(s.state())                                   # ... and is forgotten here (the read-out rebuilds the previous iterate)
        s.iterate(3 * residual_iter + 1, checked=True)            # several adoptions in a row
        s.iterate(3)                                              # unchecked: nobody asks for residuals
        trace.append(s.state(vectors=False)["primal_res"])        # accessor only: speculates, then ...
        s.iterate(1, checked=True)                                # ... a budget of one: forgotten again
        s.iterate(2 * residual_iter, checked=True)
        trace.append(s.state())
        s.destroy()
        runs[spec] = trace
        # that it happened: several speculative launches, most of them adopted (one forgotten at each read-out / budget of one)
        launched, adopted = trace[-1]["speculative_launches"], trace[-1]["speculative_adopted"]
        assert (launched >= 5 and 3 <= adopted < launched) if spec else launched == adopted == 0, (launched, adopted)
    for a, b_ in zip(runs[True], runs[False]):
        if isinstance(a, dict):
            for v in "xyzw":
                assert np.array_equal(a[v], b_[v]), v
            for v in ("tau", "sigma", "theta", "iteration", "primal_res", "dual_res", "pair_launches"):
                assert a[v] == b_[v], v
        else:
            assert a == b_
    total = int(runs[True][-1]["iteration"])
    ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.4), o, total, dtype)
    for v in "xyzw":
        assert np.array_equal(runs[True][-1][v], ost[v]), v
    # a solve that stops on its tolerance: same iteration, same result with and without
    o2 = prost.options(max_iters=5000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=1e-3, tol_abs_dual=1e-3)
    res = {}
    for spec in (True, False):
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.4)
        b[1]["allow_speculation"] = spec
        res[spec] = prost.solve(prob, b, o2)
    assert res[True]["result"] == res[False]["result"] == "Converged." and res[True]["iters"] == res[False]["iters"]
    for v in "xyzw":
        assert np.array_equal(np.asarray(res[True][v]), np.asarray(res[False][v])), v


def inpaint_problem(nx, ny, L, mask, seed=3, lmb=7.0):
    """matlab/examples/example_tv_inpaint.m:15-29: sum_1d('square', m, f, lmb) with the mask m as coefficient a, vectorial TV"""
    f = synthetic.rof_image(nx, ny, L, seed)
    u, q = prost.variable(nx * ny * L), prost.variable(2 * nx * ny * L)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", mask, f, lmb))
    prob.add_function(q, prost.function.sum_norm2(2 * L, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, L))
    return prob


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("nx,ny,L", [(37, 252, 1), (40, 128, 3), (33, 66, 2), (24, 130, 1)])
def test_inpainting_mask_runs_the_double_iteration_kernels_bit_exact(precision, dtype, nx, ny, L):
    """example_tv_inpaint.m:23: a 0 / 1 mask as coefficient a of the square data term (elem_operation_1d.hpp:42-44: a == 0 skips the
    function).  A binary a is folded into the b stream (prost_hip_mask_merge) and the two-iterations-per-launch kernels run their
    straight-line instance on it (gray: fused_iter2d_x2_kernel, 2-4 channels: fused_iter2d_mc_x2_kernel): iterates == single
    launches == oracle, bit for bit, residual iterations and the rebuilt previous iterate (z, w) included.  A mask with other
    values keeps the single launches (and still equals the oracle)."""
    prost.set_precision(precision)
    rng = np.random.default_rng(12)
    n = nx * ny * L
    binary = (rng.random(n) < 0.7).astype(np.float64)
    binary[:ny] = 0.0                                     # a whole masked column, every channel's first
    soft = binary.copy(); soft[rng.integers(0, n, 50)] = 0.5
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    for mask, expect_pairs in ((binary, True), (soft, False)):
        prob = inpaint_problem(nx, ny, L, mask)
        for step, res_iter in (("alg2", 10), ("boyd", 3)):
            st = {}
            for pair in (True, False):
                b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.3)
                b[1]["allow_pair_kernel"] = pair
                s = prost.Solver(prob, b, o)
                info = s.iterate(47, time_kernels=True, sample_every=1)
                st[pair] = s.state(); s.destroy()
                assert st[pair]["path"] == "pdhg:fused-grad2d"
                paired = any("x2_kernel" in k for k in info["kernels"])
                assert paired == (pair and expect_pairs), (pair, expect_pairs, list(info["kernels"]))
            ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.3), o, 47, dtype)
            for v in "xyzw":
                assert np.array_equal(st[True][v], st[False][v]), (v, "pair vs single")
                assert np.array_equal(st[True][v], ost[v]), (v, "vs oracle", float(np.abs(st[True][v] - ost[v]).max()))
            for v in ("tau", "sigma"):
                assert st[True][v] == st[False][v] == ost[v]
        # masked pixels of a column nobody constrains stay what the TV term alone makes of them: the data term never touched them
    prost.set_precision("double")


@pytest.mark.parametrize("solve_dual", [False, True])
def test_solve_streams_the_same_result_it_hands_to_callbacks(solve_dual):
    """Without an intermediate-solution callback prost.solve takes x, y, z, w straight from the device
    (Backend::current_solution_device, widened while they arrive); with one it goes through the host vectors the callback
    sees.  Same values either way, equal to the oracle's, also under solve_dual (roles of the four vectors exchanged,
    solver.cu:216-246) and for a backend that keeps no device-resident solution (ADMM: the vector path)."""
    prost.set_precision("single")
    prob, u, q, f = synthetic.rof_problem(40, 36, 1, seed=6)
    b = prost.backend.pdhg(stepsize="alg1", residual_iter=5)
    kw = dict(max_iters=30, verbose=False, solve_dual=solve_dual, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    streamed = prost.solve(prob, b, prost.options(num_cback_calls=0, **kw))
    seen = []
    via_vectors = prost.solve(prob, b, prost.options(num_cback_calls=3, interm_cb=lambda it, x, y: seen.append((it, len(x), len(y))) or False, **kw))
    assert seen and streamed["result"] == via_vectors["result"] == "Reached maximum iterations."
    ro = oracle.solve(synthetic.rof_problem(40, 36, 1, seed=6)[0], b, prost.options(num_cback_calls=0, **kw), np.float32)
    for v in "xyzw":
        a_, b_ = np.asarray(streamed[v]).reshape(-1), np.asarray(via_vectors[v]).reshape(-1)
        assert a_.shape == b_.shape == ro[v].shape, (v, a_.shape, b_.shape, ro[v].shape)
        assert np.array_equal(a_, b_), v
        assert np.array_equal(a_, ro[v]), (v, float(np.abs(a_ - ro[v]).max()))
    # ADMM keeps no device-resident solution: prost.solve falls back to the host vectors
    ra = prost.solve(prob, prost.backend.admm(rho0=1), prost.options(num_cback_calls=0, max_iters=5, verbose=False))
    assert np.isfinite(np.asarray(ra["x"])).all() and np.asarray(ra["x"]).size == prob.ncols


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
def test_solve_with_callbacks_matches_oracle(precision, dtype):
    """prost.solve end to end (example_rof_primaldual.m): convergence iteration, result string,
    callback schedule (linspace, solver.cu:128-135) and filled variables"""
    prost.set_precision(precision)
    seen, seen_o = [], []
    prob, u, q, f = synthetic.rof_problem(64, 64)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    r = prost.solve(prob, b, prost.options(max_iters=2000, num_cback_calls=7, verbose=False, interm_cb=lambda it, x, y: seen.append(it) or False))
    probo, uo, qo, _ = synthetic.rof_problem(64, 64)
    ro = oracle.solve(probo, b, prost.options(max_iters=2000, num_cback_calls=7, verbose=False, interm_cb=lambda it, x, y: seen_o.append(it) or False), dtype)
    assert r["result"] == ro["result"] == "Converged."
    assert int(r["iters"]) == ro["iters"] and seen == seen_o and len(seen) >= 2
    assert np.array_equal(np.asarray(r["x"]), ro["x"]) and np.array_equal(u.val, uo.val) and np.array_equal(q.val, qo.val)
    # max_iters reached + callback that stops the run
    r = prost.solve(prob, b, prost.options(max_iters=25, num_cback_calls=5, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0))
    assert r["result"] == "Reached maximum iterations." and int(r["iters"]) == 25
    r = prost.solve(prob, b, prost.options(max_iters=500, num_cback_calls=50, verbose=False, interm_cb=lambda it, x, y: it >= 100))
    assert r["result"] == "Converged." and 100 <= int(r["iters"]) <= 112


# ---------------------------------------------------------------------------------------------
# mixed operators (sparse + diags + gradient), generic PDHG and ADMM
# ---------------------------------------------------------------------------------------------
def tvl1_like_problem(nx, ny, seed=0):
    """SURVEY 8(d) C4 shape: primal u in R^(2n); v = W u with W = [diag(Ix) diag(Iy)] (block.sparse),
    g = gradient2d(nx, ny, 2) u; f(v) = sum_1d('abs', 1, b, lambda), f(g) = sum_norm2(4, false, 'abs')"""
    n = nx * ny
    Ix = synthetic.rof_image(nx, ny, 1, seed) - 0.5
    Iy = synthetic.rof_image(nx, ny, 1, seed + 1) - 0.5
    bvec = synthetic.rof_image(nx, ny, 1, seed + 2) - 0.5
    W = sp.hstack([sp.diags(Ix), sp.diags(Iy)]).tocsc()
    u = prost.variable(2 * n)
    v, g = prost.variable(n), prost.variable(4 * n)
    prob = prost.min_problem([u], [v, g])
    prob.add_function(v, prost.function.sum_1d("abs", 1, bvec, 5.0))
    prob.add_function(g, prost.function.sum_norm2(4, False, "abs"))
    prob.add_constraint(u, v, prost.block.sparse(W))
    prob.add_constraint(u, g, prost.block.gradient2d(nx, ny, 2))
    return prob


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("arg_fusion", [True, False])
def test_generic_pdhg_on_mixed_blocks(precision, dtype, arg_fusion):
    """arg_fusion: the proxes form the PDHG prox arguments on the fly (prost_hip_prox_elem_arg_*) instead of reading
    a separately written argument vector -- same bits either way."""
    prost.set_precision(precision)
    prob = tvl1_like_problem(24, 18)
    b = prost.backend.pdhg(stepsize="boyd", residual_iter=5)
    b[1]["allow_arg_fusion"] = arg_fusion
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    st = run_product(prob, b, o, 60)
    assert st["path"] == "pdhg:generic"
    assert_same_iterates(st, run_oracle(prob, b, o, 60, dtype))


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("arg_fusion", [True, False])
def test_generic_pdhg_arg_sources_cover_every_prox_kind(precision, dtype, arg_fusion):
    """Generic path (allow_fused off) on ROF-like problems whose prox lists mix the kinds that evaluate from an
    argument source: elem 1d with vector coefficients, norm2 of dims 2 / 3 / 7 (register and two-pass kernels),
    conjugated elem operations (prox_f given -> Moreau), the identity prox of an uncovered range, ragged counts
    (scalar kernels).  Against the oracle, bit for bit."""
    prost.set_precision(precision)
    rng = np.random.default_rng(11)
    cases = []
    for (nx, ny, L) in ((20, 16, 1), (9, 7, 3)):            # 9*7*3 = 189 elements: not a multiple of the vector width
        prob, u, q, f = synthetic.rof_problem(nx, ny, L, seed=3)
        cases.append(prob)
    # uncovered primal range (identity prox inserted by the problem) + prox_f (Moreau) on a 7-component norm
    n = 28
    u = prost.variable(n + 5); q = prost.variable(7 * 8)
    prob = prost.min_max_problem([u], [q])
    A = sp.random(7 * 8, n + 5, density=0.3, random_state=5, format="csc")
    prob.add_dual_pair(u, q, prost.block.sparse(A))
    prob.add_function(q, prost.function.conjugate(prost.function.sum_norm2(7, False, "huber", 1, 0, 1, 0, 0, 0.3)))
    cases.append(prob)
    for prob in cases:
        b = prost.backend.pdhg(stepsize="alg1", residual_iter=3)
        b[1]["allow_fused"] = False
        b[1]["allow_arg_fusion"] = arg_fusion
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
        st = run_product(prob, b, o, 30)
        assert st["path"] == "pdhg:generic"
        assert_same_iterates(st, run_oracle(prob, b, o, 30, dtype))


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("device_cg,fused_rounds", [(True, True), (True, False), (False, False)])
def test_admm_matches_oracle(precision, dtype, device_cg, fused_rounds):
    """device_cg=True: CG scalars resident on the device (no host round trip) -- fused_rounds: a CG round in four launches with
    the operator [W; grad] applied inside them (prost_hip_cgls_round_*), else the staged rounds around LinearOperator::Eval;
    device_cg=False: host-driven CGLS with one blocking nrm2 per scalar, as the reference does it."""
    prost.set_precision(precision)
    prob = tvl1_like_problem(16, 12)
    b = prost.backend.admm(rho0=1, residual_iter=2)
    b[1]["device_cg"] = device_cg
    b[1]["fused_rounds"] = fused_rounds
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    tol = 2e-4 if dtype == np.float32 else 1e-9
    for k in (1, 5, 20):
        st = run_product(prob, b, o, k)
        assert st["path"] == ("admm:pixel-op" if fused_rounds else "admm:generic")      # [W ; gradient2d(L = 2)]: the two-launch rounds
        ost = run_oracle(prob, b, o, k, dtype)
        assert_same_iterates(st, ost, exact=False, tol=tol)
        assert np.isclose(st["rho"], ost["rho"], rtol=1e-6)
        assert st["cg_iterations"] == ost["cg_iterations"]
        for name in ("primal_res", "dual_res"):
            assert np.isclose(st[name], ost[name], rtol=1e-3, atol=1e-5), name


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
def test_operator_norm_estimate_equals_the_oracle_bit_for_bit(precision, dtype):
    """Problem::normest (problem.cu:429-500) divides the initial step sizes when the estimate is further than 0.1 from 1
    (backend_pdhg.cu:274-286).  Its two norms per round are reductions the reference leaves to thrust; here and in the oracle they are
    order-independent sums (reduce.hpp dd_t / ExactSum), so the estimate -- 100 rounds of a power iteration -- and the rescaled tau, sigma
    are the SAME bits for any operator whose products are summed row by row (rounds 3-4: up to 138 ulp apart, iterates then only
    tolerance-compared)."""
    prost.set_precision(precision)
    rng = np.random.default_rng(17)
    cases = []
    # a random sparse operator (short rows), a diagonal band operator, a two-column image gradient, the C4 operator [W ; gradient2d(L = 2)]
    A = sp.random(60, 45, density=0.06, random_state=3, format="csc", dtype=np.float64); A.data = np.round(A.data * 8) / 4 + 0.25
    u, v = prost.variable(45), prost.variable(60)
    p1 = prost.min_max_problem([u], [v])
    p1.add_function(u, prost.function.sum_1d("square", 1, rng.random(45), 2.0)); p1.add_function(v, prost.function.sum_1d("ind_box01"))
    p1.add_dual_pair(u, v, prost.block.sparse(A))
    cases.append(("sparse", p1))
    u, v = prost.variable(50), prost.variable(50)
    p2 = prost.min_max_problem([u], [v])
    p2.add_function(u, prost.function.sum_1d("abs")); p2.add_function(v, prost.function.sum_1d("ind_box01"))
    p2.add_dual_pair(u, v, prost.block.diags(50, 50, [1.5, -0.5, 2.0], [-1, 0, 3]))
    cases.append(("diags", p2))
    cases.append(("grad 2 columns", synthetic.rof_problem(2, 40, 1, seed=4)[0]))
    cases.append(("grad 1x64x3", synthetic.rof_problem(1, 64, 3, seed=4)[0]))
    cases.append(("c4", tvl1_like_problem(16, 12)))
    fired = 0
    for name, prob in cases:
        o = prost.options(max_iters=20, num_cback_calls=0, verbose=False)
        for fused in (True, False):
            b = prost.backend.pdhg(stepsize="alg1", residual_iter=5, scale_steps_operator=True)
            b[1]["allow_fused"] = fused
            for k in (0, 7):
                st = run_product(prob, b, o, k)
                ost = run_oracle(prob, b, o, k, dtype)
                assert st["tau"] == ost["tau"] and st["sigma"] == ost["sigma"], (name, fused, k, st["tau"], ost["tau"])
                if k:
                    assert_same_iterates(st, ost, exact=True)
                fired += 1 if (k == 0 and st["tau"] != 1.0) else 0
    assert fired >= 4          # (the rescale did fire: otherwise this test would compare 1 with 1)
    prost.set_precision("double")


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step,residual_iter", [("boyd", 1), ("alg2", 4), ("goldstein", 3)])
def test_generic_pdhg_with_the_operator_inside_the_prox_kernels(precision, dtype, step, residual_iter):
    """allow_op_fusion (round 5; off by default) and residual_sums_in_prox (default: from 2^23 elements on): the prox launches of the generic path form K^T y / K x
    for their own elements from the operator's blocks -- row patterns, CSR rows, gradient stencils, in block order -- and add up the residual
    terms themselves, so K x is never written and an iteration is 4 launches instead of 9.  Iterates, step sizes and decisions are those of
    the separate products and of the oracle, bit for bit, on: example_deblurring.m's shape (two sparse constraint blocks, Moreau-wrapped
    proxes, the identity on the primal side), example_multilabel_fast.m as written (sparse gradient over 3 labels + the sum row, norm2 over 6
    components, linear terms), and [W ; gradient2d(L = 2)] with a stencil block; the reference's zero vectors of iterations 0 / 1, warm
    starts and the read-out of z / w included; with the rule on the device and on the host.  (The residual NORMS are compared at 1e-5
    relative, not bit for bit: the sums inside the prox launches are order-independent double-double sums, the separate reduction adds
    plain double partials -- the two agree to the last digits of a double, which is what the comparisons below allow.)"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import generic_rule_rate, multilabel_fast
    prost.set_precision(precision)
    rng = np.random.default_rng(5)
    problems = [("deblurring-like", generic_rule_rate.problem(40, 36)), ("multilabel_fast", multilabel_fast.describe(28, 24)[0]), ("c4", tvl1_like_problem(24, 20))]
    for name, prob in problems:
        prob.finalize()
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, x0=rng.random(prob.ncols), y0=0.1 * rng.standard_normal(prob.nrows))
        for iters in (1, 2, 23):
            st = {}
            # (True, .): the operator inside the prox launches; ("sums", .): separate products, the prox launches add up the residual terms
            # (residual_sums_in_prox = 2: always -- the default waits for 2^23 elements); (False, .): separate products and reductions
            for opf, dev in ((True, True), (True, False), (False, True), ("sums", True), ("sums", False)):
                b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.2)
                b[1]["allow_op_fusion"] = 2 if opf is True else 0
                b[1]["residual_sums_in_prox"] = 2 if opf == "sums" else 0
                b[1]["allow_device_rules"] = dev
                st[(opf, dev)] = run_product(prob, b, o, iters)
                assert st[(opf, dev)]["path"] == "pdhg:generic" and st[(opf, dev)]["operator_in_prox_kernels"] == (1.0 if opf is True else 0.0), (name, opf)
                assert st[(opf, dev)]["residual_sums_in_prox_launches"] == (1.0 if opf == "sums" else 0.0), (name, opf)
            ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.2), o, iters, dtype)
            for key, s_ in st.items():
                assert_same_iterates(s_, ost, exact=True)
                assert s_["tau"] == ost["tau"] and s_["sigma"] == ost["sigma"], (name, iters, key)
                for r_ in ("primal_res", "dual_res"):
                    assert np.isclose(s_[r_], ost[r_], rtol=1e-5, atol=1e-7), (name, iters, key, r_)
    prost.set_precision("double")


def pixel_coupled_problem(nx, ny, L, has_d=True, d_first=True, seed=0):
    """K = [D ; gradient2d(nx, ny, L)] (or the other order, or the gradient alone): D = [diag(w_0) ... diag(w_{L-1})] couples the L
    channels of one pixel -- the C4 shape for L = 2"""
    n = nx * ny
    w = [synthetic.rof_image(nx, ny, 1, seed + c) - 0.5 for c in range(L)]
    bvec = synthetic.rof_image(nx, ny, 1, seed + 7) - 0.5
    u = prost.variable(L * n)
    v, g = prost.variable(n), prost.variable(2 * L * n)
    cons = ([v, g] if d_first else [g, v]) if has_d else [g]
    prob = prost.min_problem([u], cons)
    if has_d == "csr":
        # round 6: D as a general sparse block with one row per pixel -- 0 .. 4 entries per row anywhere in the L n columns (empty rows and
        # empty columns included), the shape of a warp matrix that gathers at displaced pixels
        rng = np.random.default_rng(seed + 11)
        counts = rng.integers(0, 5, size=n)
        rows = np.repeat(np.arange(n), counts)
        near = np.clip(rows % n + rng.integers(-3 * ny, 3 * ny + 1, size=rows.size), 0, n - 1)
        cols = near + n * rng.integers(0, L, size=rows.size)
        D = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, L * n))     # (duplicates are summed: fewer entries)
        D.eliminate_zeros()
        prob.add_function(v, prost.function.sum_1d("abs", 1, bvec, 5.0))
        prob.add_constraint(u, v, prost.block.sparse(D.tocsc()))
    elif has_d:
        prob.add_function(v, prost.function.sum_1d("abs", 1, bvec, 5.0))
        prob.add_constraint(u, v, prost.block.sparse(sp.hstack([sp.diags(wc) for wc in w]).tocsc()))
    prob.add_function(g, prost.function.sum_norm2(2 * L, False, "abs"))
    prob.add_constraint(u, g, prost.block.gradient2d(nx, ny, L))
    return prob


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("nx,ny,L,has_d,d_first", [(16, 12, 2, True, True), (9, 20, 2, True, False), (7, 8, 1, True, True), (12, 16, 3, True, True),
                                                    (11, 24, 2, False, True), (5, 1028, 1, False, True), (40, 264, 2, True, True),
                                                    (16, 12, 2, "csr", True), (9, 20, 2, "csr", False), (7, 8, 1, "csr", True), (12, 16, 3, "csr", True),
                                                    (40, 264, 2, "csr", True)])
def test_admm_cg_variants_agree_bit_for_bit(precision, dtype, nx, ny, L, has_d, d_first):
    """One ADMM run, four implementations of the CGLS graph projection -- CG rounds of TWO launches (pixel-ordered, operators
    [D ; gradient2d]), of FOUR launches (operator inside the stage kernels), the staged rounds around LinearOperator::Eval, and the
    host-driven solve with one blocking nrm2 per scalar (the reference's sequence) -- form every vector element by the same
    expressions and every CG scalar from order-independent sums (reduce.hpp): x, y, z, w, rho, the residual norms and the CG
    iteration counts are IDENTICAL, and equal to the oracle's, which accumulates the same way (oracle/prost_oracle.cpp, ExactSum).
    Round 4 compared these with 2e-4 / 2e-5: the sums were grouped differently per path (profiles/r04_fuzz.log: one solve in
    ~30 000 ended in a different round)."""
    prost.set_precision(precision)
    prob = pixel_coupled_problem(nx, ny, L, has_d, d_first, seed=3)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    variants = {"pixel": dict(), "fused4": dict(pixel_rounds=False), "staged": dict(fused_rounds=False), "host": dict(device_cg=False)}
    for iters in (1, 4, 13):
        st = {}
        for name, kw in variants.items():
            b = prost.backend.admm(rho0=2, residual_iter=2)
            b[1].update(kw)
            st[name] = run_product(prob, b, o, iters)
        assert st["pixel"]["path"] == "admm:pixel-op" and st["fused4"]["path"] == "admm:fused-op" and st["staged"]["path"] == "admm:generic"
        ost = run_oracle(prob, prost.backend.admm(rho0=2, residual_iter=2), o, iters, dtype)
        for name in variants:
            for v in "xyzw":
                assert np.array_equal(st[name][v], st["pixel"][v]), (iters, name, v, float(np.abs(st[name][v] - st["pixel"][v]).max()))
            assert st[name]["cg_iterations"] == ost["cg_iterations"], (iters, name)
            assert st[name]["rho"] == st["pixel"]["rho"]
            for r_ in ("primal_res", "dual_res"):
                assert st[name][r_] == st["pixel"][r_], (iters, name, r_)
        assert_same_iterates(st["pixel"], ost, exact=True)
        assert st["pixel"]["rho"] == ost["rho"]
        for r_ in ("primal_res", "dual_res"):
            # (fp64 on the random rows of "csr": the product's sums are double-double, the oracle's exact -- equal after rounding except for a
            #  sum that lies within 2^-106 of a rounding boundary; seen once: 0.7075602990803983 against ...84)
            slack = np.spacing(ost[r_]) if has_d == "csr" and dtype == np.float64 else 0.0
            assert abs(st["pixel"][r_] - ost[r_]) <= slack, (iters, r_, st["pixel"][r_], ost[r_])
    prost.set_precision("double")


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("m,n1,n2", [(2, 84, 6), (40, 24, 56), (300, 256, 192)])
def test_admm_fused_rounds_with_two_blocks_on_the_same_rows_bit_for_bit(precision, dtype, m, n1, n2):
    """K = [A B]: two sparse blocks on the SAME rows.  LinearOperator::Eval with accumulate adds block by block onto what is there --
    r = (r0 + A t_1) + B t_2 -- and the operator-in-stage kernels must do the same (round 6: they formed r0 + (A t_1 + B t_2) in
    INIT_RK / PRE_ZK, one unit in the last place apart; found by tools/fuzz_parity.py --mode generic on a 2-row operator whose CG
    solves run on rounding noise and amplify it).  Four-launch rounds == staged rounds == oracle, bit for bit."""
    prost.set_precision(precision)
    rng = np.random.default_rng(17)
    A = sp.random(m, n1, density=min(1.0, 3.0 / n1), random_state=3, format="csc") + sp.csc_matrix(([1.0], ([0], [0])), shape=(m, n1))
    B = sp.random(m, n2, density=min(1.0, 3.0 / n2), random_state=4, format="csc") + sp.csc_matrix(([0.5], ([m - 1], [n2 - 1])), shape=(m, n2))
    u1, u2, v = prost.variable(n1), prost.variable(n2), prost.variable(m)
    prob = prost.min_problem([u1, u2], [v])
    prob.add_function(u1, prost.function.sum_1d("square", 1, rng.standard_normal(n1), 2.0))
    prob.add_function(u2, prost.function.sum_1d("abs", 1, rng.standard_normal(n2), 0.7))
    prob.add_function(v, prost.function.sum_1d("huber", 1, rng.standard_normal(m), 1, 0, 0, 0.4))
    prob.add_constraint(u1, v, prost.block.sparse(A.tocsc()))
    prob.add_constraint(u2, v, prost.block.sparse(B.tocsc()))
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    for iters in (2, 9):
        st = {}
        for fused_rounds in (True, False):
            b = prost.backend.admm(rho0=1, residual_iter=4)
            b[1]["fused_rounds"] = fused_rounds
            b[1]["pixel_rounds"] = False
            st[fused_rounds] = run_product(prob, b, o, iters)
        assert st[True]["path"] == "admm:fused-op" and st[False]["path"] == "admm:generic"
        ost = run_oracle(prob, prost.backend.admm(rho0=1, residual_iter=4), o, iters, dtype)
        for k in (True, False):
            assert_same_iterates(st[k], ost, exact=True)
            assert st[k]["cg_iterations"] == ost["cg_iterations"] and st[k]["rho"] == ost["rho"], (iters, k)
    prost.set_precision("double")


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("nx,ny,L", [(10, 16, 6), (9, 13, 5)])
def test_admm_fused_rounds_on_a_gradient3d_operator_match_the_oracle(precision, dtype, nx, ny, L):
    """the four-launch CG rounds with a gradient3d stencil inside the stage kernels (16 bytes of rows per lane where every
    boundary allows it: 16 rows; one row per lane otherwise: 13 rows), volumetric TV through ADMM: min_u lmb/2 |u - f|^2 + |g|_{2,1}
    s.t. g = grad3d u -- against the oracle, and the staged rounds as the A/B"""
    prost.set_precision(precision)
    n = nx * ny * L
    f = synthetic.rof_image(nx, ny, L, 5)
    u, g = prost.variable(n), prost.variable(3 * n)
    prob = prost.min_problem([u], [g])
    prob.add_function(u, prost.function.sum_1d("square", 1, f, 8.0))
    prob.add_function(g, prost.function.sum_norm2(3, False, "abs"))
    prob.add_constraint(u, g, prost.block.gradient3d(nx, ny, L))
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    tol = 2e-4 if dtype == np.float32 else 1e-9
    st = {}
    for fused_rounds in (True, False):
        b = prost.backend.admm(rho0=3, residual_iter=2)
        b[1]["fused_rounds"] = fused_rounds
        st[fused_rounds] = run_product(prob, b, o, 12)
    ost = run_oracle(prob, prost.backend.admm(rho0=3, residual_iter=2), o, 12, dtype)
    for k, s_ in st.items():
        assert s_["path"] == ("admm:fused-op" if k else "admm:generic")
        assert_same_iterates(s_, ost, exact=False, tol=tol)
        assert s_["cg_iterations"] == ost["cg_iterations"], k
        assert np.isclose(s_["rho"], ost["rho"], rtol=1e-6)
    assert_same_iterates(st[True], st[False], exact=False, tol=tol / 4)
    prost.set_precision("double")


def test_admm_and_pdhg_agree_at_convergence_on_the_c4_shape():
    """the PRODUCT's two backends on the C4 shape (block.sparse + gradient2d(L = 2), sum_1d abs + sum_norm2(4) abs), fp64: ADMM
    (device-resident CGLS graph projection; no reference-held vector pins it) and generic PDHG (bit-exact
    against the oracle, which the real reference build pins) converge to the same energy and the same sum(x) -- the check the
    survey ran on the reference itself (sum x = 465.796 vs 465.885 on a 32 x 32 TV-L1 problem) -- and both match the oracle's
    ADMM run"""
    from test_oracle_reference_tests import c4_shape_problem
    prost.set_precision("double")
    prob, energy = c4_shape_problem(32, 32)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    out = {}
    for name, b, its in (("admm", prost.backend.admm(rho0=1), 1500), ("pdhg", prost.backend.pdhg(stepsize="boyd", residual_iter=10), 4000)):
        st = run_product(prob, b, o, its)
        assert st["path"] == ("admm:pixel-op" if name == "admm" else "pdhg:generic"), st["path"]
        out[name] = (energy(st["x"]), st["x"].sum(), st["x"])
    (ea, sa, xa), (ep, sp_, xp) = out["admm"], out["pdhg"]
    assert abs(ea - ep) / ep < 1e-3, (ea, ep)
    assert abs(sa - sp_) / abs(sp_) < 1e-4, (sa, sp_)
    ost = run_oracle(prob, prost.backend.admm(rho0=1), o, 1500, np.float64)
    assert abs(energy(ost["x"]) - ea) / ea < 1e-6
    assert float(np.abs(ost["x"] - xa).max()) < 1e-6 * max(1.0, float(np.abs(xa).max()))


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("cg_graph", [False, True])
@pytest.mark.parametrize("cg_tol_min,cg_max_iter", [(0.3, 10), (1e-2, 25), (1e-5, 3)])
def test_admm_device_cg_stopping_rules(precision, dtype, cg_tol_min, cg_max_iter, cg_graph):
    """Early convergence (rounds queued after the stopping test must not touch x), the iteration cap,
    and odd sizes (vector tails) -- iterates and CG iteration counts against the oracle."""
    prost.set_precision(precision)
    prob = tvl1_like_problem(23, 19, seed=3)
    b = prost.backend.admm(rho0=2, residual_iter=1, cg_tol_min=cg_tol_min, cg_tol_max=cg_tol_min * 1e-3, cg_max_iter=cg_max_iter)
    b[1]["cg_graph"] = cg_graph          # the CG rounds replayed from a captured HIP graph (opt-in) or launched directly
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    tol = 5e-4 if dtype == np.float32 else 1e-9
    seen = set()
    for k in (1, 2, 7, 15):
        st = run_product(prob, b, o, k)
        ost = run_oracle(prob, b, o, k, dtype)
        assert_same_iterates(st, ost, exact=False, tol=tol)
        assert st["cg_iterations"] == ost["cg_iterations"], (k, st["cg_iterations"], ost["cg_iterations"])
        seen.add(int(st["cg_iterations"]))
    if cg_max_iter == 3:
        assert 3 in seen and seen <= {0, 3}     # 0: the very first solve starts from x = 0 with |A'b| < eps
    if cg_tol_min == 0.3:
        assert min(seen) < cg_max_iter          # the early-exit path was taken
    # ROF through ADMM agrees with PDHG at convergence (the two backends solve the same problem)
    prob, u, q, f = synthetic.rof_problem(32, 32)
    r1 = prost.solve(prob, prost.backend.admm(rho0=15), prost.options(max_iters=400, num_cback_calls=0, verbose=False, tol_rel_primal=1e-6, tol_rel_dual=1e-6, tol_abs_primal=1e-6, tol_abs_dual=1e-6))
    x_admm = np.asarray(r1["x"]).copy()
    r2 = prost.solve(prob, prost.backend.pdhg(stepsize="alg2", alg2_gamma=0.5, residual_iter=10), prost.options(max_iters=3000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-6, tol_rel_dual=1e-6, tol_abs_primal=1e-6, tol_abs_dual=1e-6))
    assert np.abs(x_admm - np.asarray(r2["x"])).max() < 5e-3


# ---------------------------------------------------------------------------------------------
# eval_linop / eval_prox: the reference's own MATLAB tests through the product
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision,dtype", PRECISIONS)
def test_eval_linop_reference_tests(precision, dtype):
    prost.set_precision(precision)
    rng = np.random.default_rng(0)
    # test_linop_gradient2d.m / test_linop_gradient3d.m
    for d3, (nx, ny, L) in ((False, (307, 229, 8)), (True, (151, 291, 7))):
        k = 3 if d3 else 2
        blk = (prost.block.gradient3d if d3 else prost.block.gradient2d)(nx, ny, L, False)
        linop = [blk(0, 0, nx * ny * k * L, nx * ny * L)[0]]
        K = (spmat_gradient3d if d3 else spmat_gradient2d)(nx, ny, L)
        inp, inp2 = rng.random(nx * ny * L), rng.random(nx * ny * L * k)
        x, _, _, _ = prost.eval_linop(linop, inp, False)
        y, rowsum, colsum, ms = prost.eval_linop(linop, inp2, True)
        assert np.linalg.norm(x - K @ inp) <= 1e-3 and np.linalg.norm(y - K.T @ inp2) <= 1e-3
        assert np.all(rowsum == 2) and np.all(colsum == (6 if d3 else 4)) and ms >= 0
        xo, _, _ = oracle.eval_linop(linop, inp, False, dtype)
        yo, _, _ = oracle.eval_linop(linop, inp2, True, dtype)
        assert np.array_equal(x, xo) and np.array_equal(y, yo)
    # test_linop_diags.m (smaller grid) and test_linop_sparse_zero.m mixed in one operator
    nrows, ncols = 1200, 331
    linop, Krows, row = [], [], 0
    for i in range(2):
        col, krow = 0, []
        for j in range(3):
            if (i + j) % 3 == 0:
                Kb = sp.random(nrows, ncols, density=0.01, random_state=i * 3 + j)
                linop.append(prost.block.sparse(Kb)(row, col, nrows, ncols)[0])
            elif (i + j) % 3 == 1:
                fac = rng.random(11); ofs = rng.permutation(nrows + ncols - 2)[:11] - nrows + 1
                Kb = spdiags_const(nrows, ncols, fac, ofs)
                linop.append(prost.block.diags(nrows, ncols, fac, ofs)(row, col, nrows, ncols)[0])
            else:
                Kb = sp.csr_matrix((nrows, ncols))
                linop.append(prost.block.zero()(row, col, nrows, ncols)[0])
            krow.append(Kb); col += ncols
        Krows.append(sp.hstack(krow)); row += nrows
    K = sp.vstack(Krows).tocsr()
    inp, inp2 = rng.standard_normal(K.shape[1]), rng.standard_normal(K.shape[0])
    x, _, _, _ = prost.eval_linop(linop, inp, False)
    y, rowsum, colsum, _ = prost.eval_linop(linop, inp2, True)
    tol = 1e-3 if dtype == np.float64 else 1e-2
    assert np.linalg.norm(x - K @ inp) <= tol and np.linalg.norm(y - K.T @ inp2) <= tol
    assert np.allclose(rowsum, np.asarray(abs(K).sum(axis=1)).ravel(), rtol=1e-5, atol=1e-6)
    assert np.allclose(colsum, np.asarray(abs(K).sum(axis=0)).ravel(), rtol=1e-5, atol=1e-6)
    xo, _, _ = oracle.eval_linop(linop, inp, False, dtype)
    assert np.array_equal(x, xo)


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
def test_eval_prox_reference_tests(precision, dtype):
    prost.set_precision(precision)
    rng = np.random.default_rng(1)
    # test_prox_sum_norm2.m
    N, d = 6000, 7
    P = -2 + 4 * rng.random((N, d))
    Q, _ = prost.eval_prox(prost.function.sum_norm2(d, False, "ind_leq0", np.ones(N), 1, np.ones(N), 0, 0, 0, 0), P.reshape(-1, order="F"), 1, np.ones(N * d))
    nrm = np.sqrt((P ** 2).sum(axis=1, keepdims=True))
    assert np.abs(Q.reshape((N, d), order="F") - np.where(nrm <= 1, P, P / nrm)).max() < 1e-5
    # test_prox_conjugate.m
    N = 5000
    a, b, c, dd, e, y = (rng.random(N) for _ in range(6))
    tau, Tau = rng.random(), rng.random(N) + 1e-3
    f = prost.function.sum_1d("abs", a, b, c, dd, e)
    x, _ = prost.eval_prox(f, y, tau, Tau)
    x2, _ = prost.eval_prox(prost.function.conjugate(prost.function.conjugate(f)), y, tau, Tau)
    assert np.abs(x - x2).max() <= (1e-5 if dtype == np.float64 else 2e-3)
    assert np.array_equal(x, oracle.eval_prox(f, y, tau, Tau, dtype))
    assert np.array_equal(x2, oracle.eval_prox(prost.function.conjugate(prost.function.conjugate(f)), y, tau, Tau, dtype))
    # every registered function through the factory, both elem operations
    for fn in prost.function.FUNCTIONS_1D:
        for builder in (prost.function.sum_1d(fn, 1.5, b, 2.0, 0.1, 0.2, 0.5 if fn == "lq" else 0.7, 1.1),
                        prost.function.sum_norm2(5, False, fn, 1.5, 0.3, 2.0, 0.1, 0.2, 0.5 if fn == "lq" else 0.7, 1.1)):
            got, _ = prost.eval_prox(builder, y * 4 - 2, tau, Tau)
            exp = oracle.eval_prox(builder, y * 4 - 2, tau, Tau, dtype)
            if fn == "lq":
                assert np.allclose(got, exp, rtol=5e-5, atol=5e-5)
            else:
                assert np.array_equal(got, exp), fn
    # ind_epi_quad (no reference test covers it; oracle pinned by ProjectEpiQuadNd goldens)
    count, dim = 1000, 3
    arg = rng.uniform(-2, 2, count * dim)
    eq = prost.function.sum_ind_epi_quad(dim, False, rng.uniform(0.5, 2, count), rng.uniform(-1, 1, count * (dim - 1)), rng.uniform(-1, 1, count))
    got, _ = prost.eval_prox(eq, arg, 1.0, np.ones(count * dim))
    assert np.allclose(got, oracle.eval_prox(eq, arg, 1.0, np.ones(count * dim), dtype), rtol=2e-5, atol=2e-5)
    from prost_amd import _capi
    with pytest.raises(_capi.ProstError, match="doesn't match size of prox"):
        _capi.command("eval_prox", [prost.function.sum_1d("abs")(0, 5), np.ones((4, 1)), 1.0, np.ones((4, 1))], nlhs=1)


# ---------------------------------------------------------------------------------------------
# full-size properties (BASELINE sizes; the oracle is too slow here, so size-independent checks)
# ---------------------------------------------------------------------------------------------
def test_fullsize_fused_equals_generic_and_gap_decreases():
    prost.set_precision("single")
    n = 2048
    prob, u, q, f = synthetic.rof_problem(n, n)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    states = {}
    for fused in (True, False):
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        b[1]["allow_fused"] = fused
        states[fused] = run_product(prob, b, o, 25)
    for v in "xyzw":
        assert np.array_equal(states[True][v], states[False][v]), v       # two independent kernel paths, identical bits
    assert states[True]["primal_res"] == states[False]["primal_res"] or np.isclose(states[True]["primal_res"], states[False]["primal_res"], rtol=1e-6)


def test_large_rgb_single_kernel_equals_generic():
    """RGB 1536 x 1024 (the multi-channel one-kernel iteration at a size where thousands of workgroups with their
    per-column LDS barriers are in flight) against the generic nine-vector path: identical bits after 23 iterations,
    residual iterations included."""
    prost.set_precision("single")
    prob, u, q, f = synthetic.rof_problem(1536, 1024, 3, seed=6)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    states = {}
    for fused in (True, False):
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=5, alg2_gamma=0.5)
        b[1]["allow_fused"] = fused
        states[fused] = run_product(prob, b, o, 23)
    assert states[True]["path"] == "pdhg:fused-grad2d" and states[False]["path"] == "pdhg:generic"
    for v in "xyzw":
        assert np.array_equal(states[True][v], states[False][v]), v
    assert np.isclose(states[True]["primal_res"], states[False]["primal_res"], rtol=1e-5)
    prost.set_precision("double")


def test_fullsize_4096_adjointness_and_energy():
    """<Kx, y> = <x, K^T y> at 4096^2 (double), and the ROF energy decreases along PDHG iterates"""
    prost.set_precision("double")
    n = 4096
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(n * n), rng.standard_normal(2 * n * n)
    linop = [prost.block.gradient2d(n, n, 1)(0, 0, 2 * n * n, n * n)[0]]
    kx, _, _, _ = prost.eval_linop(linop, x, False)
    kty, _, _, _ = prost.eval_linop(linop, y, True)
    assert np.isclose(np.dot(kx, y), np.dot(x, kty), rtol=1e-9)
    prost.set_precision("single")
    prob, u, q, f = synthetic.rof_problem(n, n)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    s = prost.Solver(prob, b, o)
    energies = []
    for it in range(4):
        s.iterate(40)
        st = s.state()
        xx = st["x"].reshape(n, n)            # [x][y], y contiguous
        gx = np.diff(xx, axis=0, append=xx[-1:, :]); gy = np.diff(xx, axis=1, append=xx[:, -1:])
        energies.append(0.5 * 10.0 * ((st["x"] - f) ** 2).sum() + np.sqrt(gx ** 2 + gy ** 2).sum())
        assert np.isfinite(st["x"]).all() and np.abs(st["y"]).max() <= np.sqrt(2) + 1e-4      # dual stays in the unit balls
    s.destroy()
    assert energies[-1] < energies[0] and energies[3] <= energies[2] * (1 + 1e-6)


def test_fullsize_c4_admm_device_cg_agrees_with_host_cg():
    """C4 at its BASELINE size (TV-L1 flow-like, 1024^2, block.sparse + gradient2d(L=2)): the fused passes with
    device-resident CG scalars against the reference's launch sequence with host-side scalars.  The two differ only
    in the association order of the norm reductions, so the iterates agree to fp32 round-off amplified by 10 CG
    iterations (rel 1e-3 after 15 outer iterations), take the same number of CG iterations, and the primal residual
    falls."""
    prost.set_precision("single")
    prob = tvl1_like_problem(1024, 1024)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    st = {}
    for device_cg in (True, False):
        b = prost.backend.admm(rho0=1)
        b[1]["device_cg"] = device_cg
        s = prost.Solver(prob, b, o)
        s.iterate(3); early = s.state()["primal_res"]
        s.iterate(12); st[device_cg] = s.state(); s.destroy()
        assert st[device_cg]["primal_res"] < early
    for v in "xz":
        scale = float(np.abs(st[False][v]).max())
        assert float(np.abs(st[True][v] - st[False][v]).max()) <= 1e-3 * scale, v
    assert st[True]["cg_iterations"] == st[False]["cg_iterations"]
    assert np.isclose(st[True]["primal_res"], st[False]["primal_res"], rtol=1e-2)
    prost.set_precision("double")


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("step", STEPS)
@pytest.mark.parametrize("residual_iter", [1, 2, 3, 4, 7, 10])
def test_pair_kernel_schedule_is_invisible(prec, dtype, step, residual_iter):
    """Two iterations per launch (prost_hip_fused_iteration2) where nobody observes the iterate in
    between: the state after ANY number of iterations -- x, y, the constraint variables z, w (which
    need x^(k-1), y^(k-1)), the residuals and the step sizes -- is bit-identical to the path that
    launches every iteration separately, and to the oracle."""
    prost.set_precision(prec)
    for (nx, ny) in ((24, 16), (9, 252), (40, 500), (24, 18), (9, 251)):
        for iters in (1, 2, 3, 5, 8, 9, 10, 11, 23):
            states = []
            for pair in (True, False):
                prob, u, q, f = synthetic.rof_problem(nx, ny, seed=3)
                b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.5)
                b[1]["allow_pair_kernel"] = pair
                o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
                s = prost.Solver(prob, b, o)
                s.iterate(iters)
                st = s.state()
                # a second batch on the same handle: the pairing restarts from an odd/even offset
                s.iterate(7)
                st2 = s.state()
                s.destroy()
                states.append((st, st2))
            for a_, b_ in zip(states[0], states[1]):
                for v in "xyzw":
                    assert np.array_equal(a_[v], b_[v]), (nx, ny, iters, v)
                for v in ("tau", "sigma", "theta", "iteration"):
                    assert v in a_ and a_[v] == b_[v], (nx, ny, iters, v, a_[v], b_[v])
                # same residual terms, summed in double in a different order (62- vs 63-lane strips)
                for v in ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm"):
                    assert np.isclose(a_[v], b_[v], rtol=1e-6, atol=0), (nx, ny, iters, v, a_[v], b_[v])
        prob, u, q, f = synthetic.rof_problem(nx, ny, seed=3)
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.5)
        o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
        assert_same_iterates(run_product(prob, b, o, 23), run_oracle(prob, b, o, 23, dtype))


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("step", STEPS)
@pytest.mark.parametrize("L,residual_iter,data_term", [(3, 10, "square"), (3, 4, "square"), (3, 3, "square"), (4, 10, "square"), (2, 10, "square"), (2, 5, "square"), (3, 10, "abs"), (2, 10, "abs")])
def test_multichannel_pair_schedule_is_invisible(prec, dtype, step, L, residual_iter, data_term):
    """Vectorial TV with 2 / 3 / 4 channels, fp32 and fp64: two iterations per launch (prost_hip_fused_iteration_mc_x2) wherever neither k
    nor k+2 is a residual iteration (k+1 may be one: the kernel forms its sums).  The state after ANY number of iterations -- x, y, the constraint variables z, w (which need
    the previous iterate, rebuilt by one single launch after a pair), residuals, step sizes -- is bit-identical to the path that
    launches every iteration separately, and the iterates equal the oracle's."""
    prost.set_precision(prec)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    for (nx, ny) in ((12, 16), (9, 252), (40, 500), (10, 67)):
        prob, u, q, f = synthetic.rof_problem(nx, ny, L, seed=3, data_term=data_term, lmb=10.0 if data_term == "square" else 0.7)
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
                x2 = [k for k in names if k.startswith("fused_iter2d_mc_x2_kernel")]
                if residual_iter >= 3 and iters >= 8:        # a pair starts at k >= 2 unless k or k + 2 is a residual iteration
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
        assert_same_iterates(run_product(prob, bo, o, 23), run_oracle(prob, bo, o, 23, dtype))


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
def test_pair_kernel_inside_solve_with_callbacks(prec, dtype):
    """prost.solve with an intermediate-solution callback schedule and a convergence stop: fused
    pairs never straddle a callback iteration, so the callback sees the same iterates (and the same
    iteration numbers) with and without the pair kernel."""
    prost.set_precision(prec)
    seen = {}
    for pair in (True, False):
        prob, u, q, f = synthetic.rof_problem(32, 24, seed=5)
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=5, alg2_gamma=0.5)
        b[1]["allow_pair_kernel"] = pair
        log = []
        o = prost.options(max_iters=97, num_cback_calls=7, verbose=False, tol_rel_primal=1e-30, tol_rel_dual=1e-30, tol_abs_primal=1e-30, tol_abs_dual=1e-30,
                          interm_cb=lambda it, x, y: log.append((it, np.array(x, copy=True), np.array(y, copy=True))) or False)
        res = prost.solve(prob, b, o)
        seen[pair] = (log, res)
    la, lb = seen[True][0], seen[False][0]
    assert len(la) == len(lb) and len(la) >= 7
    for (ia, xa, ya), (ib, xb, yb) in zip(la, lb):
        assert ia == ib and np.array_equal(xa, xb) and np.array_equal(ya, yb)
    for v in ("x", "y", "z", "w"):
        assert np.array_equal(np.asarray(seen[True][1][v]), np.asarray(seen[False][1][v])), v


def test_fullsize_4096_pair_launches_equal_single_launches():
    """the headline configuration itself (4096^2 fp32, alg2, residual_iter 10): 57 iterations run as two-
    iterations-per-launch kernels (plain, residual and rebuilt-previous-iterate paths all occur) give
    the same x, y, z, w bit for bit as 57 single launches"""
    prost.set_precision("single")
    n = 4096
    states = {}
    for pair in (True, False):
        prob, u, q, f = synthetic.rof_problem(n, n)
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        b[1]["allow_pair_kernel"] = pair
        o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
        s = prost.Solver(prob, b, o)
        s.iterate(57)
        states[pair] = s.state()
        s.destroy()
    for v in "xyzw":
        assert np.array_equal(states[True][v], states[False][v]), v
    assert states[True]["iteration"] == states[False]["iteration"] == 57
    for v in ("tau", "sigma", "theta"):
        assert states[True][v] == states[False][v]
    for v in ("primal_res", "dual_res"):
        assert np.isclose(states[True][v], states[False][v], rtol=1e-6)


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["alg2", "boyd"])
def test_pair_kernel_convergence_stop_is_identical(prec, dtype, step):
    """prost.solve stopping on the residual criterion: the run with pair launches stops at the same iteration
    with the same x, y and constraint variables z, w (rebuilt previous iterate) as the run without"""
    prost.set_precision(prec)
    out = {}
    for pair in (True, False):
        prob, u, q, f = synthetic.rof_problem(40, 36, seed=8)
        b = prost.backend.pdhg(stepsize=step, residual_iter=6, alg2_gamma=0.5)
        b[1]["allow_pair_kernel"] = pair
        o = prost.options(max_iters=5000, num_cback_calls=3, verbose=False, tol_rel_primal=2e-3, tol_rel_dual=2e-3, tol_abs_primal=2e-3, tol_abs_dual=2e-3)
        out[pair] = prost.solve(prob, b, o)
    assert out[True]["iters"] == out[False]["iters"] and out[True]["iters"] < 5000 and out[True]["result"] == out[False]["result"]
    for v in "xyzw":
        assert np.array_equal(np.asarray(out[True][v]), np.asarray(out[False][v])), v


@pytest.mark.parametrize("mode,cases", [("fused", 250), ("generic", 1200), ("large", 60), ("sharded", 150)])
def test_randomised_differential_run_against_the_oracle(mode, cases):
    """tools/fuzz_parity.py.  fused: random shapes / geometries / step rules / precisions / iteration counts of the fused PDHG paths
    (gray, 2-4 channels, volumes; ROF, TV-L1, inpainting), default launch schedule.  generic: random compositions of sparse /
    gradient / diags / identity / zero blocks over one or two primal and up to three dual variables with functions of the sum_1d /
    sum_norm2 family (per-element coefficients, conjugates), min-max and constrained form, PDHG and ADMM.  PDHG bit for bit against
    the oracle, ADMM within its tolerance -- apart from the classes the tool's header explains and counts (residual-threshold ties,
    initial steps rescaled by a norm estimate that differs in the last places, sparse rows long enough for cooperative sums).
    large: shapes of up to 40 M elements (several row strips, hundreds of column chunks), fused path == generic path on the device.
    sharded: one image over 2-5 column slabs with random halo widths, owned columns == the oracle's whole-image iterates."""
    # (the tool runs in a child that the fork server starts -- tests/conftest.py: a process that has initialised the GPU starts no others)
    import multiprocessing as mp
    import sys

    import multirank_workers as workers
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx = mp.get_context("forkserver")
    out = ctx.Queue()
    cmd = [sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--mode", mode, "--cases", str(cases), "--seed", "3", "--budget-s", "150"]
    p = ctx.Process(target=workers.run_command, args=(cmd, {}, root, out))
    p.start()
    rc, stdout, stderr = out.get(timeout=700)
    p.join(timeout=60)
    summary = [l for l in stdout.splitlines() if l.startswith("fuzz_parity:")]
    assert rc == 0 and summary, stdout[-3000:] + stderr[-2000:]
    assert " 0 failures" in summary[0], summary[0]


def test_solve_with_a_one_element_variable():
    """a dual variable of ONE element (found by tools/fuzz_parity.py): the result vector of length one used to come back as a scalar,
    which prob.fill_variables could not slice"""
    prost.set_precision("double")
    n = 12
    u, q = prost.variable(n), prost.variable(1)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", 1, np.linspace(0, 1, n), 2.0))
    prob.add_function(q, prost.function.sum_1d("ind_box01"))
    prob.add_dual_pair(u, q, prost.block.diags(1, n, [1.0, -1.0], [0, 1]))
    r = prost.solve(prob, prost.backend.pdhg(stepsize="alg1"), prost.options(max_iters=30, num_cback_calls=0, verbose=False))
    assert r["y"].shape == (1,) and q.val.shape == (1,) and u.val.shape == (n,)
    ro = oracle.solve(prob, prost.backend.pdhg(stepsize="alg1"), prost.options(max_iters=30, num_cback_calls=0, verbose=False), np.float64)
    assert r["iters"] == ro["iters"] and np.array_equal(r["x"], ro["x"]) and np.array_equal(r["y"], ro["y"])


def _stencil_matrices():
    """sparse matrices that ARE stencils: the gradient of the reference's examples (spmat_gradient2d.m), its 3-D form, a 3 x 3 blur
    with zero boundary (as example_deblurring.m builds one), and a matrix without structure as the negative control"""
    rng = np.random.default_rng(5)
    nx, ny = 40, 130
    blur = sp.kron(sp.diags([0.25, 0.5, 0.25], [-1, 0, 1], shape=(nx, nx)), sp.diags([0.2, 0.6, 0.2], [-1, 0, 1], shape=(ny, ny))).tocsc()
    # convmtx2's FULL convolution (example_deblurring.m:15-16): the output image is larger than the input, so column - row drifts from one image
    # column to the next and only ANCHORED patterns (offsets from the row's first column, round 5) repeat; one and three channels
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import deblurring
    kernel = np.array([[0.0, 0.1, 0.2], [0.05, 0.3, 0.0], [0.15, 0.0, 0.1], [0.0, 0.05, 0.05]])
    conv = deblurring.convmtx2(kernel, 66, 30)
    conv_motion = sp.kron(sp.eye(3), deblurring.convmtx2(deblurring.motion_kernel(9, 45.0), 48, 20)).tocsc()
    return [("gradient2d", spmat_gradient2d(nx, ny, 1), 2), ("gradient2d rgb", spmat_gradient2d(24, 70, 3), 2), ("gradient3d", spmat_gradient3d(12, 30, 14), 2),
            ("blur", blur, 2), ("full convolution", conv, 2), ("full convolution, motion kernel, rgb", conv_motion, 2),
            ("random", sp.random(5200, 4100, density=3.0 / 4100, random_state=3, format="csc"), 0)]


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
def test_stencils_written_out_as_sparse_matrices_run_from_row_patterns(precision, dtype):
    """BlockSparse::Initialize recognises rows that repeat a few (column - row, value) sequences -- or, for matrices between different
    geometries, (column - first column of the row, value) sequences: anchored tables -- and applies such a matrix from one
    16-bit pattern number per row + a table (prost_hip_pattern_spmv_*): the products and a PDHG solve equal the CSR path's
    (set_quirks(sparse_patterns=0)) and the oracle's bit for bit; a matrix without structure stays CSR."""
    prost.set_precision(precision)
    rng = np.random.default_rng(1)
    try:
        for name, K, expect in _stencil_matrices():
            K = sp.csc_matrix(K)
            m, n = K.shape
            u, q = prost.variable(n), prost.variable(m)
            res = {}
            for on in (1, 0):
                prost.set_quirks(sparse_patterns=on)
                prob = prost.min_max_problem([u], [q])
                prob.add_function(u, prost.function.sum_1d("square", 1, np.linspace(0, 1, n), 4.0))
                prob.add_function(q, prost.function.sum_1d("ind_box01", 0.5, -0.5))
                prob.add_dual_pair(u, q, prost.block.sparse(K))
                prob.finalize()
                b = prost.backend.pdhg(stepsize="alg2", residual_iter=5, alg2_gamma=0.3, scale_steps_operator=False)
                o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
                s = prost.Solver(prob, b, o)
                s.iterate(23)
                res[on] = s.state()
                s.destroy()
                if on:
                    assert res[on]["sparse_pattern_products"] == expect, (name, res[on]["sparse_pattern_products"])
                    ost = run_oracle(prob, b, o, 23, dtype)
                else:
                    assert res[on]["sparse_pattern_products"] == 0
            # (rows of more than 6 entries on average: the CSR kernels sum with cooperating lanes, the patterns -- like the oracle --
            # sequentially: equal to rounding there, bit for bit otherwise)
            long_rows = K.nnz > 6 * min(m, n)
            for v in "xyzw":
                if long_rows:
                    assert np.allclose(res[1][v], res[0][v], rtol=0, atol=1e-4 if dtype == np.float32 else 1e-12), (name, v)
                else:
                    assert np.array_equal(res[1][v], res[0][v]), (name, v)
                assert np.array_equal(res[1][v], ost[v]), (name, v, "oracle")
            # the products themselves, accumulating and not, through eval_linop
            for tr in (False, True):
                rhs = rng.uniform(-1, 1, m if tr else n)
                prost.set_quirks(sparse_patterns=1)
                a = np.asarray(prost.eval_linop(prob.data["linop"], rhs, tr)[0]).ravel()
                prost.set_quirks(sparse_patterns=0)
                b_ = np.asarray(prost.eval_linop(prob.data["linop"], rhs, tr)[0]).ravel()
                assert np.array_equal(a, oracle.eval_linop(prob.data["linop"], rhs, tr, dtype)[0].ravel()), (name, tr)
                assert np.allclose(a, b_, rtol=0, atol=1e-5 if dtype == np.float32 else 1e-13) if long_rows else np.array_equal(a, b_), (name, tr)
    finally:
        prost.set_quirks(sparse_patterns=1)


def test_sparse_blocks_without_entries_and_with_empty_rows():
    """edge cases of the row-pattern recognition: a block.sparse of a matrix without a single entry (no table to build: stays CSR), and a
    large stencil matrix most of whose rows are empty (one pattern of length zero beside the others)"""
    prost.set_precision("double")
    n = 600
    rhs = np.linspace(-1, 1, n)
    empty = sp.csc_matrix((n, n))
    got = np.asarray(prost.eval_linop(prost.block.sparse(empty)(0, 0, n, n)[0:1], rhs, False)[0]).ravel()
    assert np.array_equal(got, np.zeros(n))
    K = sp.lil_matrix((n, n))
    for r in range(0, n, 7):
        K[r, r] = 2.0
        if r + 3 < n:
            K[r, r + 3] = -0.5
    K = sp.csc_matrix(K)
    lin = prost.block.sparse(K)(0, 0, n, n)[0:1]
    for tr in (False, True):
        got = np.asarray(prost.eval_linop(lin, rhs, tr)[0]).ravel()
        assert np.array_equal(got, oracle.eval_linop(lin, rhs, tr, np.float64)[0].ravel()), tr
        assert np.allclose(got, (K.T if tr else K) @ rhs, rtol=0, atol=1e-15)


def test_large_sparse_blocks_transposed_on_all_host_cores():
    """BlockSparse::CreateFromCSC transposes the CSC arrays it is handed (csr2csc, reference src/common.cu:55-82).  From 4 M entries on
    the counting sort runs on all host cores where the row sub-ranges touch narrow column ranges (banded matrices: stencils, warps) and
    sequentially otherwise (an unstructured matrix): both against the oracle's sequential transposition, through the forward product
    (which reads the transposed arrays) and the preconditioner sums."""
    prost.set_precision("double")
    rng = np.random.default_rng(3)
    n = 1100
    band = sp.csc_matrix(spmat_gradient2d(n, n, 1))
    band.data = band.data * rng.uniform(0.5, 1.5, band.nnz)          # per-entry values: no row patterns, the CSR arrays themselves are applied
    # (NOT scipy.sparse.random: with a legacy seed it draws the positions through a permutation of all rows x columns -- 13 TiB here)
    nnz_r = 5_000_000
    rand = sp.coo_matrix((rng.uniform(-1, 1, nnz_r), (rng.integers(0, 1_500_000, nnz_r), rng.integers(0, 1_200_000, nnz_r))), shape=(1_500_000, 1_200_000)).tocsc()
    for name, K in (("banded", band), ("unstructured", rand)):
        assert K.nnz > (1 << 22), (name, K.nnz)
        lin = prost.block.sparse(K)(0, 0, K.shape[0], K.shape[1])[0:1]
        rhs = rng.uniform(-1, 1, K.shape[1])
        got, rowsum, colsum, _ = prost.eval_linop(lin, rhs, False)
        want, orow, ocol = oracle.eval_linop(lin, rhs, False, np.float64)
        assert np.array_equal(np.asarray(got).ravel(), want.ravel()), name
        assert np.array_equal(np.asarray(rowsum).ravel(), orow.ravel()) and np.array_equal(np.asarray(colsum).ravel(), ocol.ravel()), name


# ---------------------------------------------------------------------------------------------
# residual-driven step rules and the stopping test on the device (kernels_pdhg_rule.hip)
# ---------------------------------------------------------------------------------------------
RULE_SCALARS = ("tau", "sigma", "theta", "iteration", "primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual")


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["goldstein", "boyd"])
@pytest.mark.parametrize("residual_iter", [1, 3, 10])
@pytest.mark.parametrize("with_comm", [False, True])
def test_device_resident_step_rules_are_invisible(precision, dtype, step, residual_iter, with_comm):
    """goldstein / boyd (boyd with residual_iter = 1 is the reference's DEFAULT, pdhg.m:4-14) on the one-kernel 2-D path: batches of
    iterations run with the rule and the stopping test evaluated ON THE DEVICE, one host wait per batch.  Whatever the caller does --
    unchecked iteration, the loop of prost.solve, reading the state in the middle, batches cut by the budget, more than one batch
    (> 240 iterations) -- iterates, every step-size decision (tau, sigma after each stretch), residual norms and iteration counts
    equal those of the host-side rule EXACTLY, and the iterates equal the oracle's bit for bit.  Tolerances are chosen so that the
    rules' comparisons flip during the run.  with_comm: a (one-rank, host-callback) communicator -- the sums pass the all-reduce
    before the rule kernel reads them."""
    prost.set_precision(precision)
    if with_comm:
        prost.comm_init_host(lambda a: None, 1)
    try:
        prob, u, q, f = synthetic.rof_problem(44, 252, 1, seed=9)
        o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=2e-2, tol_rel_dual=2e-2, tol_abs_primal=0, tol_abs_dual=0)
        runs = {}
        for dev in (True, False):
            b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
            b[1]["allow_device_rules"] = dev
            s = prost.Solver(prob, b, o)
            trace = []
            s.iterate(7)                                  # iterations 0, 1 on the host loop, then one short batch
            trace.append(s.state())
            s.iterate(500)                                # three batches (240 + 240 + 20)
            trace.append(s.state(vectors=False))
            s.iterate(2)                                  # a budget below the batch threshold: host loop
            s.iterate(31)
            trace.append(s.state())
            s.destroy()
            runs[dev] = trace
            assert (trace[-1]["device_rule_batches"] >= 4) if dev else trace[-1]["device_rule_batches"] == 0, trace[-1]["device_rule_batches"]
        changed = set()
        for a, b_ in zip(runs[True], runs[False]):
            for v in RULE_SCALARS + ("pair_launches",):
                assert a[v] == b_[v], (v, a[v], b_[v])
            changed.add((a["tau"], a["sigma"]))
            if "x" in a:
                for v in "xyzw":
                    assert np.array_equal(a[v], b_[v]), v
        assert len(changed) >= 2, changed                 # the rule did fire between the read-outs
        if not with_comm:
            ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o, 540, dtype)
            assert_same_iterates(runs[True][-1], ost)
            assert runs[True][-1]["tau"] == ost["tau"] and runs[True][-1]["sigma"] == ost["sigma"]
    finally:
        if with_comm:
            prost.comm_destroy()


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["goldstein", "boyd"])
@pytest.mark.parametrize("L,nx,ny,residual_iter", [(3, 24, 60, 1), (4, 20, 33, 3), (2, 30, 64, 10), (3, 40, 124, 1), (3, 36, 252, 10)])
def test_device_resident_step_rules_with_colour_channels(precision, dtype, step, L, nx, ny, residual_iter):
    """2-4 channels (the reference's examples read RGB images): the single-iteration kernels -- channels in one lane for 2, on the
    wavefronts of a workgroup for 3 / 4 -- and their double-iteration kernel read the step sizes from the device record as well.  Same checks as for gray
    values: identical to the host-side rule in every scalar and iterate, identical to the oracle, a complete solve stops at the same
    iteration."""
    prost.set_precision(precision)
    prob, u, q, f = synthetic.rof_problem(nx, ny, L, seed=12)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=2e-2, tol_rel_dual=2e-2, tol_abs_primal=0, tol_abs_dual=0)
    runs = {}
    for dev in (True, False):
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
        b[1]["allow_device_rules"] = dev
        s = prost.Solver(prob, b, o)
        s.iterate(9)
        a = s.state()
        s.iterate(300)
        runs[dev] = (a, s.state())
        s.destroy()
        assert (runs[dev][1]["device_rule_batches"] >= 3) if dev else runs[dev][1]["device_rule_batches"] == 0
    for a, b_ in zip(runs[True], runs[False]):
        for v in RULE_SCALARS:
            assert a[v] == b_[v], (v, a[v], b_[v])
        for v in "xyzw":
            assert np.array_equal(a[v], b_[v]), v
    ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o, 309, dtype)
    assert_same_iterates(runs[True][1], ost)
    assert runs[True][1]["tau"] == ost["tau"] and runs[True][1]["sigma"] == ost["sigma"]
    o2 = prost.options(max_iters=3000, num_cback_calls=0, verbose=False, tol_rel_primal=5e-3, tol_rel_dual=5e-3, tol_abs_primal=5e-3, tol_abs_dual=5e-3)
    b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
    got, exp = prost.solve(prob, b, o2), oracle.solve(prob, b, o2, dtype)
    assert got["result"] == exp["result"] == "Converged." and int(got["iters"]) == int(exp["iters"]), (got["iters"], exp["iters"])
    for v in "xyzw":
        assert np.array_equal(np.asarray(got[v]), np.asarray(exp[v])), v


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["goldstein", "boyd"])
@pytest.mark.parametrize("nx,ny,L,residual_iter", [(12, 64, 6, 1), (10, 124, 5, 3), (9, 66, 16, 10), (14, 33, 4, 1), (8, 252, 14, 10)])
def test_device_resident_step_rules_on_volumes(precision, dtype, step, nx, ny, L, residual_iter):
    """gradient3d (round 5): the one-kernel iteration, its planes-across-wavefronts form and the double-iteration kernel read tau, sigma,
    theta and the prox terms from the device record (prost_hip_fused_iteration3d_rec / _3d_pw_rec / _3d_x2_rec), the fold of a residual
    launch applies the rule and the stopping test -- the reference's DEFAULT options (boyd, residual_iter 1) on a volume ran one host
    round trip per iteration before.  Identical to the host-side rule in every scalar and iterate, identical to the oracle, a complete
    solve stops at the same iteration.  (ny = 33: odd height, the two-pass kernels -- no record kernels there, the host loop stays.)"""
    prost.set_precision(precision)
    prob, u, q, f = synthetic.tv3d_problem(nx, ny, L, lmb=6.0, seed=12)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=2e-2, tol_rel_dual=2e-2, tol_abs_primal=0, tol_abs_dual=0)
    runs = {}
    record_kernels = ny % (4 if dtype == np.float32 else 2) == 0
    for dev in (True, False):
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
        b[1]["allow_device_rules"] = dev
        s = prost.Solver(prob, b, o)
        s.iterate(9)
        a = s.state()
        s.iterate(300)
        runs[dev] = (a, s.state())
        s.destroy()
        assert runs[dev][1]["path"] == "pdhg:fused-grad3d"
        assert (runs[dev][1]["device_rule_batches"] >= 3) if (dev and record_kernels) else runs[dev][1]["device_rule_batches"] == 0
    for a, b_ in zip(runs[True], runs[False]):
        for v in RULE_SCALARS:
            assert a[v] == b_[v], (v, a[v], b_[v])
        for v in "xyzw":
            assert np.array_equal(a[v], b_[v]), v
    ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o, 309, dtype)
    assert_same_iterates(runs[True][1], ost)
    assert runs[True][1]["tau"] == ost["tau"] and runs[True][1]["sigma"] == ost["sigma"]
    o2 = prost.options(max_iters=3000, num_cback_calls=0, verbose=False, tol_rel_primal=5e-3, tol_rel_dual=5e-3, tol_abs_primal=5e-3, tol_abs_dual=5e-3)
    b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
    got, exp = prost.solve(prob, b, o2), oracle.solve(prob, b, o2, dtype)
    assert got["result"] == exp["result"] == "Converged." and int(got["iters"]) == int(exp["iters"]), (got["iters"], exp["iters"])
    for v in "xyzw":
        assert np.array_equal(np.asarray(got[v]), np.asarray(exp[v])), v


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["goldstein", "boyd"])
@pytest.mark.parametrize("residual_iter,cbacks", [(1, 0), (4, 7), (10, 0)])
def test_device_resident_stopping_test_stops_where_the_host_loop_stops(precision, dtype, step, residual_iter, cbacks):
    """complete prost.solve runs that stop on their tolerance in the MIDDLE of a device batch: result, iteration count and x, y, z, w
    equal the host-side rule's and the oracle's (solver.cu:141-150 evaluated by the rule kernel; the launches enqueued behind the
    stopping iteration return at once and the buffer roles are put back).  cbacks: scheduled observations cut the batches short."""
    prost.set_precision(precision)
    prob, u, q, f = synthetic.rof_problem(60, 124, 1, seed=4)
    for tol in (1e-2, 2e-3):
        o = prost.options(max_iters=3000, num_cback_calls=cbacks, verbose=False, tol_rel_primal=tol, tol_rel_dual=tol, tol_abs_primal=tol, tol_abs_dual=tol)
        res = {}
        for dev in (True, False):
            b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
            b[1]["allow_device_rules"] = dev
            res[dev] = prost.solve(prob, b, o)
        exp = oracle.solve(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o, dtype)
        assert res[True]["result"] == res[False]["result"] == exp["result"] == "Converged.", (res[True]["result"], exp["result"])
        assert int(res[True]["iters"]) == int(res[False]["iters"]) == int(exp["iters"]), (res[True]["iters"], res[False]["iters"], exp["iters"])
        assert int(res[True]["iters"]) >= 5            # (the batches start at iteration 2)
        for v in "xyzw":
            assert np.array_equal(np.asarray(res[True][v]), np.asarray(res[False][v])), v
            assert np.array_equal(np.asarray(res[True][v]), np.asarray(exp[v])), v
    # the checked loop, entered again after it has stopped: one more launch and the same answer, like the host loop
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=3e-2, tol_rel_dual=3e-2, tol_abs_primal=3e-2, tol_abs_dual=3e-2)
    its = {}
    for dev in (True, False):
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
        b[1]["allow_device_rules"] = dev
        s = prost.Solver(prob, b, o)
        seq = [s.iterate(1000, checked=True)["converged"], s.state(vectors=False)["iteration"]]
        seq += [s.iterate(1000, checked=True)["converged"], s.state(vectors=False)["iteration"]]
        s.iterate(40)                                          # unchecked: runs all 40 whatever the residuals say
        seq.append(s.state(vectors=False)["iteration"])
        st = s.state()
        s.destroy()
        its[dev] = (seq, st)
    # (one more LAUNCH: one iteration, or two where the host loop pairs them)
    assert its[True][0] == its[False][0] and its[True][0][0] and its[True][0][2] and its[True][0][3] - its[True][0][1] in (1, 2), its[True][0]
    assert its[True][0][4] == its[True][0][3] + 40
    for v in "xyzw":
        assert np.array_equal(its[True][1][v], its[False][1][v]), v


def _deblurring_like_problem(nx, ny, seed, primal_form=True):
    """example_deblurring.m:29-37 in small: min_problem with TWO constraint blocks on u -- a sparse blur operator B (two-tap average along y) with
    a square data term on v = B u and the sparse gradient with the TV norm on g = grad u -- no function on u itself (the identity
    prox); primal_form = False: the same operator as a min_max problem with the dual functions written directly"""
    import scipy.sparse as sp
    n = nx * ny
    rng = np.random.default_rng(seed)
    # (two taps per row and per column: rows this short are summed sequentially by the CSR kernels, like the oracle -- bit for bit;
    # longer rows are summed by several lanes and compare with a tolerance, tests/test_gpu_kernels.py::test_csr_spmv)
    B = sp.kron(sp.identity(nx), sp.diags([np.full(ny, 0.5), np.full(ny - 1, 0.5)], [0, 1])).tocsr()
    f = synthetic.rof_image(nx, ny, 1, seed=seed)
    fb = B @ f + 0.02 * rng.standard_normal(n)
    u, v, g = prost.variable(n), prost.variable(n), prost.variable(2 * n)
    if primal_form:
        prob = prost.min_problem([u], [v, g])
        prob.add_function(v, prost.function.sum_1d("square", 1, fb, 20.0, 0, 0))
        prob.add_function(g, prost.function.sum_norm2(2, False, "abs", 1, 0, 1, 0, 0))
        prob.add_constraint(u, v, prost.block.sparse(B))
        prob.add_constraint(u, g, prost.block.sparse(spmat_gradient2d(nx, ny, 1)))
    else:
        prob = prost.min_max_problem([u], [v, g])
        prob.add_function(v, prost.function.conjugate(prost.function.sum_1d("square", 1, fb, 20.0, 0, 0)))
        prob.add_function(g, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
        prob.add_dual_pair(u, v, prost.block.sparse(B))
        prob.add_dual_pair(u, g, prost.block.gradient2d(nx, ny, 1))
    return prob


@pytest.mark.parametrize("step", ["goldstein", "boyd"])
def test_device_resident_step_rules_on_the_generic_path_with_a_communicator(step):
    """the same batches when the four sums pass an all-reduce first (a one-rank host-callback communicator): the reduction launch
    does not apply the rule, the all-reduce and the one-thread rule kernel follow -- identical to the host-side rule in every scalar
    and iterate"""
    prost.set_precision("single")
    prob = _deblurring_like_problem(26, 40, 3, True)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=2e-2, tol_rel_dual=2e-2, tol_abs_primal=0, tol_abs_dual=0)
    prost.comm_init_host(lambda a: None, 1)
    try:
        runs = {}
        for dev in (True, False):
            b = prost.backend.pdhg(stepsize=step, residual_iter=1)
            b[1]["allow_device_rules"] = dev
            s = prost.Solver(prob, b, o)
            s.iterate(9)
            a = s.state()
            s.iterate(300)
            runs[dev] = (a, s.state())
            s.destroy()
            assert (runs[dev][1]["device_rule_batches"] >= 3) if dev else runs[dev][1]["device_rule_batches"] == 0
        for a, b_ in zip(runs[True], runs[False]):
            for v in RULE_SCALARS:
                assert a[v] == b_[v], (v, a[v], b_[v])
            for v in "xyzw":
                assert np.array_equal(a[v], b_[v]), v
    finally:
        prost.comm_destroy()


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", ["goldstein", "boyd"])
@pytest.mark.parametrize("residual_iter,primal_form", [(1, True), (3, False), (10, True)])
def test_device_resident_step_rules_on_the_generic_path(precision, dtype, step, residual_iter, primal_form):
    """the residual-driven rules on the device for ANY operator (round 4): example_deblurring.m's shape -- two constraint blocks, the
    identity prox on u, Moreau-wrapped elem operations on the constrained variables, boyd with residual_iter = 1 (:40-41) -- runs the
    generic kernels in batches with ONE host wait: the proxes and the residual reductions read tau, sigma, theta from the device record
    (prost_hip_use_step_record), a one-thread kernel applies the rule and the stopping test behind the sums.  Iterates, step sizes,
    residual norms and iteration counts equal the host-side rule's EXACTLY at every read-out and the oracle's iterates bit for bit;
    a complete solve stops in the middle of a batch at the oracle's iteration."""
    prost.set_precision(precision)
    prob = _deblurring_like_problem(26, 40, 3, primal_form)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=2e-2, tol_rel_dual=2e-2, tol_abs_primal=0, tol_abs_dual=0)
    runs = {}
    for dev in (True, False):
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
        b[1]["allow_device_rules"] = dev
        s = prost.Solver(prob, b, o)
        trace = []
        s.iterate(7)
        trace.append(s.state())
        s.iterate(500)
        trace.append(s.state(vectors=False))
        s.iterate(2)
        s.iterate(31)
        trace.append(s.state())
        s.destroy()
        runs[dev] = trace
        assert trace[-1]["path"] == "pdhg:generic"
        assert (trace[-1]["device_rule_batches"] >= 4) if dev else trace[-1]["device_rule_batches"] == 0, trace[-1]["device_rule_batches"]
    changed = set()
    for a, b_ in zip(runs[True], runs[False]):
        for v in RULE_SCALARS:
            assert a[v] == b_[v], (v, a[v], b_[v])
        changed.add((a["tau"], a["sigma"]))
        if "x" in a:
            for v in "xyzw":
                assert np.array_equal(a[v], b_[v]), v
    assert len(changed) >= 2, changed
    # against the oracle: the residual SUMS are reduced in another order than the oracle's (compared with a tolerance everywhere), so a
    # comparison of the rule that sits on a tie may fall the other way -- then both product paths follow the same other branch (checked
    # above).  Exact equality is asserted where the step sizes say that no tie was hit, and must hold for the example's own shape.
    ost = run_oracle(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o, 540, dtype)
    same_branches = runs[True][-1]["tau"] == ost["tau"] and runs[True][-1]["sigma"] == ost["sigma"]
    assert same_branches or not primal_form
    if same_branches:
        assert_same_iterates(runs[True][-1], ost)
    for tol in (2e-2, 5e-3):
        o2 = prost.options(max_iters=4000, num_cback_calls=3, verbose=False, tol_rel_primal=tol, tol_rel_dual=tol, tol_abs_primal=tol, tol_abs_dual=tol)
        res = {}
        for dev in (True, False):
            b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
            b[1]["allow_device_rules"] = dev
            res[dev] = prost.solve(prob, b, o2)
        exp = oracle.solve(prob, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o2, dtype)
        assert res[True]["result"] == res[False]["result"] and int(res[True]["iters"]) == int(res[False]["iters"]), (res[True]["iters"], res[False]["iters"])
        for v in "xyzw":
            assert np.array_equal(np.asarray(res[True][v]), np.asarray(res[False][v])), v
        if same_branches:
            assert res[True]["result"] == exp["result"] and int(res[True]["iters"]) == int(exp["iters"]), (res[True]["iters"], exp["iters"])
            for v in "xyzw":
                assert np.array_equal(np.asarray(res[True][v]), np.asarray(exp[v])), v


# ---------------------------------------------------------------------------------------------
# gradients handed over as sparse matrices on the fused kernels (position-dependent Tau)
# ---------------------------------------------------------------------------------------------
def _rof_sparse_gradient(nx, ny, f, lmb, as_block=False, data="square", L=1):
    """example_rof_primal.m / example_rof_primaldual.m with the gradient written as prost.block.sparse(spmat_gradient2d(nx, ny, nc))"""
    n = nx * ny * L
    u, q = prost.variable(n), prost.variable(2 * n)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d(data, 1, f, lmb))
    prob.add_function(q, prost.function.sum_norm2(2 * L, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, L) if as_block else prost.block.sparse(spmat_gradient2d(nx, ny, L)))
    return prob


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("step", STEPS)
def test_gradient_handed_over_as_a_sparse_matrix_runs_the_fused_kernels(precision, dtype, step):
    """prost.block.sparse(spmat_gradient2d(nx, ny, 1)) -- the way example_rof_primal.m:10,28, example_nonconvex_rof.m:13,45 and
    example_deblurring.m:10,37 write the operator -- is recognised entry for entry and runs the one-kernel / two-iterations-per-launch
    gradient kernels with the preconditioners THAT MATRIX has (Tau_j = 1 / column sum: 1/4 inside, 1/3 on the edges, 1/2 in the corners;
    problem.cu:262-287), not the constants of block.gradient2d (block_gradient2d.cu:154-163).  x, y, z, w, the step sizes and the
    preconditioner vectors equal the ORACLE's -- which runs the matrix as block.sparse -- bit for bit, and the generic path's
    (set_quirks(sparse_stencils=0)); the same description with block.gradient2d gives different iterates (so the test would notice
    the wrong preconditioners).  Heights on and off the vector width, pair and single launches, residual iterations."""
    prost.set_precision(precision)
    try:
        for (nx, ny), res_iter in (((40, 64), 3), ((33, 30), 1), ((16, 1028), 10), ((21, 263), 4)):
            f = synthetic.rof_image(nx, ny, 1, seed=5)
            prob = _rof_sparse_gradient(nx, ny, f, 8.0)
            b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
            o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=0, tol_abs_dual=0)
            info = prost.problem_info(prob)
            tr = np.asarray(info["scaling_right"]).reshape(nx, ny)
            assert set(np.round(1 / tr.ravel()).astype(int)) == {2, 3, 4} and np.all(np.asarray(info["scaling_left"]) == 0.5)
            for k in (1, 2, 11, 50):
                st = run_product(prob, b, o, k)
                assert st["path"] == "pdhg:fused-grad2d(sparse)", st["path"]
                ost = run_oracle(prob, b, o, k, dtype)
                assert_same_iterates(st, ost)
                for name in ("tau", "sigma", "theta"):
                    assert st[name] == ost[name], name
                for name in ("primal_res", "dual_res", "dual_var_norm", "eps_primal", "eps_dual"):
                    assert np.isclose(st[name], ost[name], rtol=1e-5, atol=1e-6), (name, st[name], ost[name])
            if step in ("alg2", "boyd") and res_iter >= 3:
                assert st["pair_launches"] > 0, (nx, ny, st["pair_launches"])
            prost.set_quirks(sparse_stencils=0)
            gen = run_product(prob, b, o, 50)
            prost.set_quirks(sparse_stencils=1)
            assert gen["path"] == "pdhg:generic"
            assert_same_iterates(st, gen)
            blk = run_product(_rof_sparse_gradient(nx, ny, f, 8.0, as_block=True), b, o, 50)
            assert blk["path"] == "pdhg:fused-grad2d" and not np.array_equal(blk["x"], st["x"])
        # a complete solve stops where the oracle stops
        prob = _rof_sparse_gradient(48, 60, synthetic.rof_image(48, 60, 1, seed=2), 8.0)
        b = prost.backend.pdhg(stepsize=step, residual_iter=2, alg2_gamma=0.5)
        o = prost.options(max_iters=4000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=1e-3, tol_abs_dual=1e-3)
        got, exp = prost.solve(prob, b, o), oracle.solve(prob, b, o, dtype)
        assert got["result"] == exp["result"] == "Converged." and int(got["iters"]) == int(exp["iters"]) and got["path"] == "pdhg:fused-grad2d(sparse)"
        for v in "xyzw":
            assert np.array_equal(np.asarray(got[v]), np.asarray(exp[v])), v
    finally:
        prost.set_quirks(sparse_stencils=1)


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("L", [2, 3, 4])
def test_colour_gradient_handed_over_as_a_sparse_matrix_runs_the_fused_kernels(precision, dtype, L):
    """spmat_gradient2d(nx, ny, nc) with nc = 2, 3, 4 channels (the examples read RGB images: example_rof_primal.m:3-10): the one-kernel
    iterations (channels in one lane for 2, on the wavefronts of a workgroup for 3 / 4) and the multi-channel double-iteration kernel
    with the position-dependent Tau of the matrix; iterates == the oracle running block.sparse, bit for bit, residual iterations included"""
    prost.set_precision(precision)
    for (nx, ny), res_iter, step in (((24, 64), 3, "alg2"), ((17, 30), 1, "boyd"), ((12, 260), 10, "goldstein")):
        f = synthetic.rof_image(nx, ny, L, seed=6)
        prob = _rof_sparse_gradient(nx, ny, f, 6.0, L=L)
        b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5)
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=0, tol_abs_dual=0)
        for k in (1, 2, 11, 40):
            st = run_product(prob, b, o, k)
            assert st["path"] == "pdhg:fused-grad2d(sparse)", st["path"]
            ost = run_oracle(prob, b, o, k, dtype)
            assert_same_iterates(st, ost)
            assert st["tau"] == ost["tau"] and st["sigma"] == ost["sigma"]
            for name in ("primal_res", "dual_res", "eps_primal", "eps_dual"):
                assert np.isclose(st[name], ost[name], rtol=1e-5, atol=1e-6), (name, st[name], ost[name])
        if res_iter >= 3:
            assert st["pair_launches"] > 0, (L, step, st["pair_launches"])     # two iterations per launch: the multi-channel pair kernel's instance for these matrices


def _rof_primal_as_the_example_writes_it(nx, ny, L, f, lmb, as_block=False, split=(100, 500)):
    """example_rof_primal.m:15-28, line for line: the problem in its PRIMAL form (min_problem: prox_f on the constrained variable, from
    which the PDHG backend derives prox_f* by Moreau's identity, backend_pdhg.cu:255-266), the data term spread over three
    sub-variables with slices of f as coefficient b, the gradient as a sparse matrix"""
    n = nx * ny * L
    u, g = prost.variable(n), prost.variable(2 * n)
    subs = [prost.sub_variable(u, size) for size in list(split) + [n - sum(split)]] if split else []      # :19-21, before min_problem
    prob = prost.min_problem([u], [g])
    if split:
        at = 0
        for sv in subs:
            prob.add_function(sv, prost.function.sum_1d("square", 1, f[at:at + sv.dim], lmb, 0, 0))
            at += sv.dim
    else:
        prob.add_function(u, prost.function.sum_1d("square", 1, f, lmb, 0, 0))
    prob.add_function(g, prost.function.sum_norm2(2 * L, False, "abs", 1, 0, 1, 0, 0))
    prob.add_constraint(u, g, prost.block.gradient2d(nx, ny, L) if as_block else prost.block.sparse(spmat_gradient2d(nx, ny, L)))
    return prob


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("L", [1, 2, 3, 4])
def test_rof_in_primal_form_with_sub_variables_runs_the_fused_kernels(precision, dtype, L):
    """example_rof_primal.m as it is written -- min_problem, three sub-variables carrying the data term, sum_norm2('abs') on the
    constrained variable (prox_f* = its Moreau wrap), block.sparse(spmat_gradient2d), boyd with residual_iter = 1 (:32-36) -- runs the
    one-kernel iterations: the pieces of prox_g merged into one coefficient stream, the Moreau wrap evaluated inside the kernel with
    the reference's expressions (prox_moreau.cu:98-134), the step-size rule on the device.  x, y, z, w and the step sizes == the ORACLE
    running the generic description, bit for bit; other step rules, block.gradient2d and an unsplit data term as well."""
    prost.set_precision(precision)
    o = prost.options(max_iters=200, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=0, tol_abs_dual=0)
    for (nx, ny), res_iter, step, as_block, split in (((30, 32), 1, "boyd", False, (100, 500)), ((25, 31), 1, "boyd", False, (100, 500)),
                                                     ((20, 36), 3, "alg2", False, (7, 1)), ((18, 40), 2, "goldstein", True, (100, 500)),
                                                     ((21, 33), 5, "alg1", False, None), ((16, 504), 10, "alg2", False, (100, 500)),
                                                     ((12, 250), 4, "boyd", True, (100, 500))):
        f = synthetic.rof_image(nx, ny, L, seed=8)
        prob = _rof_primal_as_the_example_writes_it(nx, ny, L, f, 10.0, as_block, split)
        b = prost.backend.pdhg(stepsize=step, residual_iter=res_iter, alg2_gamma=0.5, tau0=1, sigma0=1)
        for k in (1, 2, 9, 40):
            st = run_product(prob, b, o, k)
            assert st["path"] == ("pdhg:fused-grad2d" if as_block else "pdhg:fused-grad2d(sparse)"), st["path"]
            ost = run_oracle(prob, b, o, k, dtype)
            assert_same_iterates(st, ost)
            for name in ("tau", "sigma", "theta"):
                assert st[name] == ost[name], (name, k)
            for name in ("primal_res", "dual_res", "eps_primal", "eps_dual"):
                assert np.isclose(st[name], ost[name], rtol=1e-5, atol=1e-6), (name, st[name], ost[name])
        if step in ("boyd", "goldstein"):
            assert st["device_rule_batches"] >= 1, st["device_rule_batches"]
        if L == 1 and step == "alg2":
            assert st["pair_launches"] > 0, st["pair_launches"]           # two iterations per launch: the pair kernel's Moreau instance
        b[1]["allow_fused"] = False
        gen = run_product(prob, b, o, 40)
        assert gen["path"] == "pdhg:generic"
        assert_same_iterates(st, gen)
    # a complete solve stops where the oracle stops
    f = synthetic.rof_image(36, 40, L, seed=9)
    prob = _rof_primal_as_the_example_writes_it(36, 40, L, f, 10.0)
    b = prost.backend.pdhg(stepsize="boyd", residual_iter=1, alg2_gamma=0.5, tau0=1, sigma0=1)
    o = prost.options(max_iters=6000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=1e-3, tol_abs_dual=1e-3)
    got, exp = prost.solve(prob, b, o), oracle.solve(prob, b, o, dtype)
    assert got["result"] == exp["result"] == "Converged." and int(got["iters"]) == int(exp["iters"]) and got["path"] == "pdhg:fused-grad2d(sparse)"
    for v in "xyzw":
        assert np.array_equal(np.asarray(got[v]), np.asarray(exp[v])), v


def test_data_terms_on_sub_variables_that_do_not_merge_into_one_stream():
    """pieces of prox_g on sub-variables: different scalar coefficients c become ONE per-pixel vector (gray values: the run-time
    dispatched one-kernel instance reads it; three channels: no kernel takes a per-pixel c, the generic path runs); a scalar b next to
    vector b's is filled in; different functions, or a piece that is not an elem_operation:1d, stay generic -- all equal the oracle"""
    prost.set_precision("single")
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    b = prost.backend.pdhg(stepsize="boyd", residual_iter=2, tau0=1, sigma0=1)
    for L, fns, lmbs, scalar_b, want in ((1, ("square",) * 3, (10.0, 4.0, 7.0), False, "pdhg:fused-grad2d"), (1, ("square",) * 3, (10.0,) * 3, True, "pdhg:fused-grad2d"),
                                         (3, ("square",) * 3, (10.0, 4.0, 7.0), False, "pdhg:generic"), (1, ("square", "abs", "square"), (10.0,) * 3, False, "pdhg:generic"),
                                         (1, ("square", "zero", "square"), (10.0,) * 3, False, "pdhg:generic")):
        nx, ny = 22, 36
        n = nx * ny * L
        f = synthetic.rof_image(nx, ny, L, seed=11)
        u, g = prost.variable(n), prost.variable(2 * n)
        sizes = (130, 401, n - 531)
        subs = [prost.sub_variable(u, k) for k in sizes]
        prob = prost.min_problem([u], [g])
        at = 0
        for i, sv in enumerate(subs):
            bcoef = 0.25 if (scalar_b and i == 1) else f[at:at + sv.dim]
            prob.add_function(sv, prost.function.zero() if fns[i] == "zero" else prost.function.sum_1d(fns[i], 1, bcoef, lmbs[i], 0, 0))
            at += sv.dim
        prob.add_function(g, prost.function.sum_norm2(2 * L, False, "abs", 1, 0, 1, 0, 0))
        prob.add_constraint(u, g, prost.block.gradient2d(nx, ny, L))
        for k in (1, 12, 40):
            st = run_product(prob, b, o, k)
            assert st["path"] == want, (st["path"], want, fns, lmbs)
            assert_same_iterates(st, run_oracle(prob, b, o, k, np.float32))


@pytest.mark.parametrize("precision,dtype", PRECISIONS)
@pytest.mark.parametrize("L", [1, 3])
def test_nonconvex_rof_example_runs_the_fused_kernels(precision, dtype, L):
    """example_nonconvex_rof.m:16-51 as written: the regulariser given as conjugate(sum_norm2(..., 'truncquad', 1, 0, 1, 0, 0, alpha,
    lambda)) on the dual variable (a Moreau wrap around the run-time dispatched norm2 operation), the gradient as a sparse matrix,
    alg2 with residual_iter = -1.  One-kernel iterations; x, y, z, w == the oracle's generic evaluation bit for bit"""
    prost.set_precision(precision)
    nx, ny = 26, 36
    n = nx * ny * L
    f = synthetic.rof_image(nx, ny, L, seed=10)
    u, q = prost.variable(n), prost.variable(2 * n)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", 1, f, 1))
    prob.add_function(q, prost.function.conjugate(prost.function.sum_norm2(2 * L, False, "truncquad", 1, 0, 1, 0, 0, 30, 0.05)))
    prob.add_dual_pair(u, q, prost.block.sparse(spmat_gradient2d(nx, ny, L)))
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=-1, alg2_gamma=0.25)
    o = prost.options(max_iters=2000, num_cback_calls=0, verbose=False)
    for k in (1, 2, 15, 60):
        st = run_product(prob, b, o, k)
        assert st["path"] == "pdhg:fused-grad2d(sparse)", st["path"]
        assert_same_iterates(st, run_oracle(prob, b, o, k, dtype))


def test_matrices_that_are_not_quite_the_gradient_stay_on_the_generic_path():
    """the recognition compares every entry: five labels (more channels than the one-kernel iterations take), a perturbed value, a missing entry, the TV-L1 data
    term (no position-dependent instance of the pair kernel: single launches), the inpainting mask (per-pixel a) -- all still equal
    the oracle; the first three on the generic path"""
    prost.set_precision("single")
    nx, ny = 24, 40
    n = nx * ny
    f = synthetic.rof_image(nx, ny, 1, seed=7)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=5, alg2_gamma=0.5)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
    K = sp.csc_matrix(spmat_gradient2d(nx, ny, 1))
    K2 = K.copy(); K2.data[7] = 1.5
    K3 = K.tolil(); K3[3, 3] = 0; K3 = sp.csc_matrix(K3)
    for name, M, rows, cols in (("five labels", spmat_gradient2d(nx, ny // 5, 5), 2 * n, n), ("perturbed", K2, 2 * n, n), ("missing", K3, 2 * n, n)):
        u, q = prost.variable(cols), prost.variable(rows)
        prob = prost.min_max_problem([u], [q])
        prob.add_function(u, prost.function.sum_1d("square", 1, f, 8.0))
        prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
        prob.add_dual_pair(u, q, prost.block.sparse(M))
        st = run_product(prob, b, o, 20)
        assert st["path"] == "pdhg:generic", (name, st["path"])
        assert_same_iterates(st, run_oracle(prob, b, o, 20, np.float32))
    for data in ("abs",):
        prob = _rof_sparse_gradient(nx, ny, f, 0.7, data=data)
        st = run_product(prob, b, o, 33)
        assert st["path"] == "pdhg:fused-grad2d(sparse)" and st["pair_launches"] == 0
        assert_same_iterates(st, run_oracle(prob, b, o, 33, np.float32))
    mask = (np.arange(n) % 3 != 0).astype(np.float64)
    u, q = prost.variable(n), prost.variable(2 * n)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", mask, f, 8.0))
    prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, prost.block.sparse(K))
    st = run_product(prob, b, o, 33)
    assert st["path"] == "pdhg:fused-grad2d(sparse)"
    assert_same_iterates(st, run_oracle(prob, b, o, 33, np.float32))
