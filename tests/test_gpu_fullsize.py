"""Parity at the BASELINE sizes themselves (configs C2 and C3), against the CPU oracle.

The size-dependent code of the headline path -- the launch geometry `run_iter2` derives from the image
size (3876-wave grid, 18..36-column chunks, XCD tile order: kernels_fused_iter2.hip), 32-bit byte offsets
per plane, the > 2^31-byte dual vector of the 2048 x 2048 x 64 volume -- is not exercised by the small
oracle comparisons of test_gpu_solver.py / test_gpu_fused3d.py.  Here:

  C2  4096 x 4096 fp32, alg2, residual_iter 10: product (default path: pair launches) == oracle.Solver,
      bit for bit, on x, y, z, w after 12 iterations (iterations 0 and 10 are residual iterations, 2..9 and
      11 run as pairs / singles, z and w need the rebuilt previous iterate).
  C3  2048 x 2048 x 64 fp32: (a) fused path (one-kernel iterations, `_pw` layout + residual variant) ==
      generic nine-vector path, compared ON THE DEVICE (prost_hip_compare_*: no 8 GB read-back);
      (b) sub-volumes of the full-size run == oracle runs on the cropped volume.  One PDHG iteration moves
      information by one voxel (K^T reads p - e, K reads p + e), the alg2 step sizes do not depend on the
      data and the preconditioners are the constants 1/2, 1/6, so after k iterations every voxel further
      than k from the crop's artificial faces carries the bits of the full-size run.  The crops sit in the
      far corner (faces that coincide with the true volume faces are exact) and mid-volume; the third
      gradient component of ANY voxel lies beyond byte 2^31 of y.
"""
import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import synthetic

pytestmark = pytest.mark.gpu

ZERO_TOL = dict(tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)


@pytest.fixture(autouse=True)
def _gpu(hip):
    prost.set_gpu(0)
    prost.set_precision("single")
    yield
    prost.set_precision("double")


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_c2_4096_default_path_matches_oracle_bit_for_bit(prec, dtype):
    """double = the precision the reference front end ships with (config.hpp:7): the fp64 instance of the pair kernel
    (2 rows per lane, its own chunk geometry) at the headline size"""
    prost.set_precision(prec)
    n, k = 4096, 12
    prob, u, q, f = synthetic.rof_problem(n, n)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o)
    info = s.iterate(k, time_kernels=True)
    st = s.state()
    s.destroy()
    assert st["path"] == "pdhg:fused-grad2d"
    # the launch kinds the headline run consists of all occurred: plain pairs, a pair carrying the residual sums
    # or storing the middle iterate, single launches (iterations 0, 1 and the residual iteration 10)
    assert any(name.startswith("fused_iter2d_x2_kernel") for name in info["kernels"]), info["kernels"]
    oracle.set_num_threads(16)
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, dtype)
    os_.initialize()
    os_.iterate(k)
    ost = os_.state()
    sc = os_.scalars()
    del os_
    for v in "xyzw":
        assert st[v].shape == ost[v].shape
        assert np.array_equal(st[v], ost[v]), (v, int((st[v] != ost[v]).sum()), float(np.abs(st[v] - ost[v]).max()))
    for v in ("tau", "sigma", "theta"):
        assert st[v] == sc[v], v
    # residual scalars: double accumulation here, T-precision sums in the oracle (tolerance as in test_gpu_solver.py)
    for v in ("primal_res", "dual_res"):
        assert np.isclose(st[v], sc[v], rtol=1e-5), (v, st[v], sc[v])
    if prec == "double":
        return
    # the same 12 iterations through prost.solve: the result is streamed from the device into the caller's double arrays in
    # 32 MiB pieces (x: 2, y and z: 4 each, w: 2) through the pinned staging buffers -- every element must arrive, in place
    r = prost.solve(prob, b, prost.options(max_iters=k, num_cback_calls=0, verbose=False, **ZERO_TOL))
    assert r["result"] == "Reached maximum iterations." and r["iters"] == k
    for v in "xyzw":
        got = np.asarray(r[v]).reshape(-1)
        assert got.dtype == np.float64 and got.shape == ost[v].shape
        assert np.array_equal(got, ost[v].astype(np.float64)), (v, int((got != ost[v]).sum()))


def _crop(f, nx, ny, L, x0, x1, y0, y1, l0, l1):
    return np.ascontiguousarray(f.reshape(L, nx, ny)[l0:l1, x0:x1, y0:y1]).reshape(-1)


def _read_block(solver, which, comps, nx, ny, L, x0, x1, y0, y1, l0, l1):
    """sub-volume [l0:l1, x0:x1, y0:y1] of every component plane of device vector `which` -> (comps, l, x, y)"""
    n = nx * ny * L
    offs = [c * n + l * nx * ny + x * ny + y0 for c in range(comps) for l in range(l0, l1) for x in range(x0, x1)]
    seg = solver.read(which, offs, y1 - y0)
    return seg.reshape(comps, l1 - l0, x1 - x0, y1 - y0)


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_c3_2048x2048x64_fused_equals_generic_on_device_and_subvolumes_match_oracle(prec, dtype):
    """double: the one-row-per-lane, two-halo-lane geometry of the double-iteration kernel, planes of 32 MiB, 6.4 GB dual
    vector; one oracle crop (single: three)"""
    prost.set_precision(prec)
    nx, ny, L, k = 2048, 2048, 64, 12
    f = synthetic.rof_image(nx, ny, L, 42)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    solvers = {}
    for fused in (True, False):
        prob, u, q, _ = synthetic.tv3d_problem(nx, ny, L, f=f)
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        b[1]["allow_fused"] = fused
        s = prost.Solver(prob, b, o)
        s.iterate(k)
        solvers[fused] = s
        del prob
    # (a) two independent kernel paths, every bit of x, y and the previous iterate equal -- compared on the device
    diff = solvers[True].compare(solvers[False])
    for v, (count, total) in diff.items():
        assert count == 0, (v, count, total)
    # negative control of the comparison entry: one more iteration on one side must show up
    solvers[False].iterate(1)
    moved = solvers[True].compare(solvers[False])
    assert moved["x"][0] > nx * ny * L // 2 and moved["y"][0] > nx * ny * L // 2, moved
    solvers[False].destroy()
    s = solvers[True]
    scal = s.state(vectors=False)
    assert scal["path"] == "pdhg:fused-grad3d" and scal["iteration"] == k
    # (b) sub-volumes against the oracle on the cropped volume
    oracle.set_num_threads(16)
    margin = k + 1
    crops = [  # (x0, x1, y0, y1, l0, l1): far corner (three true faces), mid-volume, first planes / last columns
        (nx - 44, nx, ny - 72, ny, L - 36, L),
        (1000, 1044, 1990, 2048, 20, 56),
        (0, 40, 0, 64, 0, 34),
    ]
    if prec == "double":
        crops = crops[:1]
    for (x0, x1, y0, y1, l0, l1) in crops:
        cx, cy, cl = x1 - x0, y1 - y0, l1 - l0
        cprob, _, _, _ = synthetic.tv3d_problem(cx, cy, cl, f=_crop(f, nx, ny, L, x0, x1, y0, y1, l0, l1))
        cprob.finalize()
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        osv = oracle.Solver(cprob.data, cprob.nrows, cprob.ncols, b, o, dtype)
        osv.initialize()
        osv.iterate(k)
        ost = osv.state()
        osc = osv.scalars()
        assert (osc["tau"], osc["sigma"], osc["theta"]) == (scal["tau"], scal["sigma"], scal["theta"])     # data-independent steps
        # voxels further than k from an ARTIFICIAL face (a face that is also a face of the full volume is exact)
        lo = lambda a0: 0 if a0 == 0 else margin
        hi = lambda a1, full, c: c if a1 == full else c - margin
        sx, sy, sl = slice(lo(x0), hi(x1, nx, cx)), slice(lo(y0), hi(y1, ny, cy)), slice(lo(l0), hi(l1, L, cl))
        assert (sx.stop - sx.start) * (sy.stop - sy.start) * (sl.stop - sl.start) > 4000
        got_x = _read_block(s, "x", 1, nx, ny, L, x0, x1, y0, y1, l0, l1)
        got_y = _read_block(s, "y", 3, nx, ny, L, x0, x1, y0, y1, l0, l1)
        exp_x = ost["x"].reshape(1, cl, cx, cy)
        exp_y = ost["y"].reshape(3, cl, cx, cy)
        assert np.array_equal(got_x[:, sl, sx, sy], exp_x[:, sl, sx, sy]), ("x", (x0, y0, l0))
        assert np.array_equal(got_y[:, sl, sx, sy], exp_y[:, sl, sx, sy]), ("y", (x0, y0, l0))
        # the margin is needed: right at an artificial face the crop's boundary condition differs
        if x0 > 0:
            assert not np.array_equal(got_x[:, :, :2, :], exp_x[:, :, :2, :])
    s.destroy()


@pytest.mark.parametrize("prec", ["single", "double"])
def test_rgb_4096_pair_path_equals_generic_on_device(prec):
    """the vectorial TV of example_rof_primaldual.m (RGB, sum_norm2(6, ...)) at 4096^2: the multi-channel double-iteration
    kernel (fused_iter2d_mc_x2_kernel, its RES variant, the single-iteration kernel on the remaining iterations) against
    the generic nine-vector path, every element of x, y and the previous iterate compared on the device"""
    prost.set_precision(prec)
    n, L, k = 4096, 3, 12
    f = synthetic.rof_image(n, n, L, 42)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **ZERO_TOL)
    solvers = {}
    for fused in (True, False):
        prob, u, q, _ = synthetic.rof_problem(n, n, L, f=f)
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        b[1]["allow_fused"] = fused
        s = prost.Solver(prob, b, o)
        info = s.iterate(k, time_kernels=fused)
        if fused:
            assert any(name.startswith("fused_iter2d_mc_x2_kernel") for name in info["kernels"]), info["kernels"]
        solvers[fused] = s
        del prob
    assert solvers[True].state(vectors=False)["path"] == "pdhg:fused-grad2d" and solvers[False].state(vectors=False)["path"] == "pdhg:generic"
    for v, (count, total) in solvers[True].compare(solvers[False]).items():
        assert count == 0, (v, count, total)
    sa, sb = solvers[True].state(vectors=False), solvers[False].state(vectors=False)
    for v in ("tau", "sigma", "theta"):
        assert sa[v] == sb[v], v
    for v in ("primal_res", "dual_res"):
        assert np.isclose(sa[v], sb[v], rtol=1e-5), v
    solvers[False].iterate(1)          # negative control
    assert solvers[True].compare(solvers[False])["x"][0] > n * n
    for s in solvers.values():
        s.destroy()


def test_c4_1024_admm_matches_oracle():
    """C4 at its BASELINE size (TV-L1 flow-like, 1024^2, block.sparse + gradient2d(L = 2), ADMM with the device-resident CGLS
    graph projection) against oracle.Solver after 5 outer iterations at the tolerance of row a5 (2e-4 of the vector's scale in
    fp32, equal CG iteration counts).  ADMM / CGLS are parity-unpinned by the reference: this compares the
    product with the restatement at full size."""
    from test_gpu_solver import tvl1_like_problem
    n, k = 1024, 5
    prob = tvl1_like_problem(n, n)
    b = prost.backend.admm(rho0=1)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o)
    s.iterate(k)
    st = s.state()
    s.destroy()
    assert st["path"] == "admm:pixel-op"
    prob.finalize()
    oracle.set_num_threads(16)
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float32)
    os_.initialize()
    os_.iterate(k)
    ost = os_.state()
    ost.update(os_.scalars())
    for v in "xyzw":
        scale = max(1.0, float(np.abs(ost[v]).max()))
        assert float(np.abs(st[v] - ost[v]).max()) <= 2e-4 * scale, (v, float(np.abs(st[v] - ost[v]).max()), scale)
    assert st["cg_iterations"] == ost["cg_iterations"]
    assert np.isclose(st["rho"], ost["rho"], rtol=1e-6)


def _admm_vs_oracle(prob, k, path, threads=16):
    b = prost.backend.admm(rho0=1)
    o = prost.options(max_iters=100, num_cback_calls=0, verbose=False, **ZERO_TOL)
    s = prost.Solver(prob, b, o)
    s.iterate(k)
    st = s.state()
    s.destroy()
    assert st["path"] == path, st["path"]
    prob.finalize()
    oracle.set_num_threads(threads)
    os_ = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float32)
    os_.initialize()
    os_.iterate(k)
    ost = os_.state()
    ost.update(os_.scalars())
    del os_
    for v in "xyzw":
        scale = max(1.0, float(np.abs(ost[v]).max()))
        assert float(np.abs(st[v] - ost[v]).max()) <= 2e-4 * scale, (v, float(np.abs(st[v] - ost[v]).max()), scale)
    assert st["cg_iterations"] == ost["cg_iterations"]
    assert np.isclose(st["rho"], ost["rho"], rtol=1e-6)


def test_c4w_1024_admm_on_a_warp_matrix_matches_oracle():
    """BASELINE.json configs[3] as worded -- "block_sparse WARP MATRIX + gradient2d" -- at 1024^2: W gathers the two flow channels at
    columns displaced by a smooth flow of up to 5 pixels (4 non-zeros per row, general CSR: synthetic.warp_matrix; the reference applies
    it with cusparse csrmv, block_sparse.cu:146-211).  One row of W per pixel, so the CG rounds are the two-launch ones with the operand
    of every entry recomputed at the gathered pixel (admm:pixel-op, round 6); product against oracle.Solver after 5 outer iterations at
    the tolerance of row a5 (2e-4 of the vector's scale, equal CG iteration counts).  ADMM / CGLS parity is unpinned by the reference."""
    W = synthetic.warp_matrix(64).tocsr()
    assert (np.diff(W.indptr) == 4).all() and W.shape == (64 * 64, 2 * 64 * 64)
    far = np.abs((W.indices % (64 * 64)) - np.repeat(np.arange(64 * 64), 4))
    assert far.max() >= 4 * 64                     # entries several image columns away from the row's own pixel: a gather, not a diagonal
    _admm_vs_oracle(synthetic.tvl1_flow_problem(1024, warp=True), 5, "admm:pixel-op")


def test_c4_2048_streams_from_hbm_and_4096_falls_back_to_the_four_launch_rounds():
    """C4 beyond the Infinity Cache: at 2048^2 the two-launch pixel rounds (a solve's working set is 640 MB); at 4096^2 one partial per
    pixel tile no longer fits the reduction workspace, prost_hip_pixel_op_supported answers no and the solve runs the four-launch rounds
    (round-5 advice: the pixel path was chosen there and every solve failed)"""
    _admm_vs_oracle(synthetic.tvl1_flow_problem(2048), 3, "admm:pixel-op")
    _admm_vs_oracle(synthetic.tvl1_flow_problem(4096), 2, "admm:fused-op")


def test_c2_4096_boyd_pair_and_single_launches_take_the_same_decisions():
    """residual-driven steps (boyd) at 4096^2, 30 iterations, residual_iter 5, tolerances 1e-2 (so that the rule's
    comparisons `residual < eps` flip during the run: no change at iterations 0 and 5, tau /= 1.05 from iteration 10 on, where
    the dual residual is within 10 % of eps_dual): the run with two iterations per launch -- whose straight-line instance forms
    the residual SUMS in fp32 with FMAs -- takes the same six decisions as the run with single launches (tau, sigma
    identical), so the iterates stay identical bit for bit"""
    n, k = 4096, 30
    tol = dict(tol_rel_primal=1e-2, tol_rel_dual=1e-2, tol_abs_primal=1e-2, tol_abs_dual=1e-2)
    solvers = {}
    for pair in (True, False):
        prob, u, q, f = synthetic.rof_problem(n, n)
        b = prost.backend.pdhg(stepsize="boyd", residual_iter=5)
        b[1]["allow_pair_kernel"] = pair
        o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, **tol)
        s = prost.Solver(prob, b, o)
        info = s.iterate(k, time_kernels=True)
        assert any(name.startswith("fused_iter2d_x2_kernel") for name in info["kernels"]) == pair, info["kernels"]
        solvers[pair] = s
        del prob
    for v, (count, total) in solvers[True].compare(solvers[False]).items():
        assert count == 0, (v, count, total)
    sa, sb = solvers[True].state(vectors=False), solvers[False].state(vectors=False)
    assert sa["iteration"] == sb["iteration"] == k
    for v in ("tau", "sigma"):
        assert sa[v] == sb[v], (v, sa[v], sb[v])
    assert 0.75 < sa["tau"] < 0.9 and sa["sigma"] > 1.1    # boyd rebalanced three or four times (1.05^-4 = 0.823)
    for v in ("primal_res", "dual_res"):
        assert np.isclose(sa[v], sb[v], rtol=1e-5), v
    for s in solvers.values():
        s.destroy()


@pytest.mark.parametrize("step,residual_iter", [("alg2", 10), ("boyd", 5)])
def test_rof_1024_solved_to_tolerance_stops_where_the_oracle_stops(step, residual_iter):
    """ROF 1024^2 fp32 solved to the tolerance of example_rof_primaldual.m (1e-4 on all four) against the ORACLE: (a) the trace of a
    run observed at every residual iteration -- tau, sigma identical (every decision of boyd's rule: the product evaluates it on the
    device, from residual sums its pair kernel forms in fp32 with FMAs), residual norms and eps to the stated tolerance -- and (b) a
    complete prost.solve: result string, stopping iteration, x, y, z, w bit for bit.  This is the check that the tolerance-compared
    residual sums never flip a decision on the way to convergence at a size where they add up 10^6 terms."""
    n = 1024
    tol = dict(tol_rel_primal=1e-4, tol_rel_dual=1e-4, tol_abs_primal=1e-4, tol_abs_dual=1e-4)
    prob, u, q, f = synthetic.rof_problem(n, n, lmb=10.0)
    b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter, alg2_gamma=0.5)
    o = prost.options(max_iters=10000, num_cback_calls=0, verbose=False, **tol)
    oracle.set_num_threads(16)
    prob.finalize()
    orc = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float32)
    orc.initialize()
    s = prost.Solver(prob, b, o)
    changes, k, converged = set(), 0, False
    worst = 0.0
    while k < 4000 and not converged:
        # up to and including the next residual iteration (0, R, 2R, ...): observed right after the rule / the stopping test ran
        step_k = 1 if k == 0 else residual_iter
        converged = bool(s.iterate(step_k, checked=True)["converged"])
        orc.iterate(step_k)
        k += step_k
        st, osc = s.state(vectors=False), orc.scalars()
        assert st["iteration"] == osc["iteration"] == k
        assert st["tau"] == osc["tau"] and st["sigma"] == osc["sigma"] and st["theta"] == osc["theta"], (k, st["tau"], osc["tau"], st["sigma"], osc["sigma"])
        changes.add((st["tau"], st["sigma"]))
        for name in ("primal_res", "dual_res", "eps_primal", "eps_dual"):
            rel = abs(st[name] - osc[name]) / max(abs(osc[name]), 1e-30)
            worst = max(worst, rel)
            assert rel < 2e-3, (k, name, st[name], osc[name])          # near convergence the sums are differences of nearly equal numbers
        o_conv = osc["primal_res"] < osc["eps_primal"] and osc["dual_res"] < osc["eps_dual"]
        assert converged == bool(o_conv), (k, st["primal_res"], st["eps_primal"], st["dual_res"], st["eps_dual"], osc)
    assert converged and 50 < k < 4000, k
    assert len(changes) >= (3 if step == "boyd" else 10), len(changes)
    s.destroy()
    del orc
    got = prost.solve(prob, b, o)
    exp = oracle.solve(prob, b, o, np.float32)
    assert got["result"] == exp["result"] == "Converged." and int(got["iters"]) == int(exp["iters"]) == k, (got["iters"], exp["iters"], k)
    for v in "xyzw":
        assert np.array_equal(np.asarray(got[v]), np.asarray(exp[v])), v
