"""The reference's example scripts, mirrored in examples/, run end to end on the MI355X (reduced sizes)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import prost_amd as prost

pytestmark = pytest.mark.gpu


def test_example_rof_primaldual_converges_by_its_own_gap_callback():
    import rof_rgb_gap_callback as ex
    prost.set_gpu(0)
    prost.set_precision("double")          # the MEX default (config.hpp:7)
    result, gaps, img = ex.main(nx=70, ny=48, nc=3, max_iters=10000, verbose=False)
    assert result["result"] == "Stopped by user." or result["result"] == "Converged."
    assert gaps and gaps[-1] < 1e-5 and gaps[-1] <= gaps[0]
    assert img.shape == (3, 70, 48) and np.isfinite(img).all()


def test_example_rof_primal_with_sub_variables_runs_fused_and_equals_the_generic_path():
    """example_rof_primal.m as written (primal form, three sub-variables, sparse gradient, boyd / residual_iter = 1): the one-kernel
    iterations with the rule on the device; the same run on the generic path stops at the same iteration with the same image"""
    import rof_primal_sub_variables as ex
    prost.set_gpu(0)
    prost.set_precision("double")
    result, gaps, img, _ = ex.main(nx=70, ny=48, nc=3, max_iters=10000, verbose=False)
    assert result["result"] == "Converged." and result["path"] == "pdhg:fused-grad2d(sparse)"
    assert len(gaps) >= 2 and abs(gaps[-1]) < abs(gaps[0]) and abs(gaps[-1]) < 1e-2
    ref, _, img_ref, _ = ex.main(nx=70, ny=48, nc=3, max_iters=10000, verbose=False, backend_opts={"allow_fused": False})
    assert ref["path"] == "pdhg:generic" and int(ref["iters"]) == int(result["iters"]) and np.array_equal(img, img_ref)


def test_example_tvl1_removes_salt_and_pepper_noise():
    import tvl1_salt_and_pepper as ex
    prost.set_gpu(0)
    prost.set_precision("double")
    result, err_noisy, err_denoised = ex.main(nx=96, ny=64, nc=1, max_iters=20000, verbose=False)
    assert result["result"] in ("Converged.", "Reached maximum iterations.")
    assert err_denoised < 0.5 * err_noisy


def test_example_tv_inpaint_fills_the_holes_with_pair_launches():
    """example_tv_inpaint.m: the mask as coefficient a of the data term.  The solve runs two iterations per launch (binary mask folded
    into the data stream), converges, keeps the known pixels close to the data and gives the oracle's solution and energy."""
    import oracle
    import tv_inpaint as ex
    prost.set_gpu(0)
    prost.set_precision("double")
    result, energy, m, f, u = ex.main(nx=60, ny=48, nc=3, max_iters=30000, verbose=False, tol=1e-6)
    assert result["result"] == "Converged." and int(result["pair_launches"]) > int(result["iters"]) // 3
    assert np.isfinite(u).all() and np.abs((u - f)[m > 0]).mean() < 0.1
    # the same description through the CPU oracle (restatement of the reference path)
    rng = np.random.default_rng(42)
    from prost_amd import synthetic
    nx, ny, nc = 60, 48, 3
    uu, qq = prost.variable(nx * ny * nc), prost.variable(2 * nx * ny * nc)
    prob = prost.min_max_problem([uu], [qq])
    prob.add_function(uu, prost.function.sum_1d("square", m, f, 7))
    prob.add_function(qq, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(uu, qq, prost.block.gradient2d(nx, ny, nc))
    o = prost.options(max_iters=30000, num_cback_calls=250, verbose=False, tol_rel_primal=1e-6, tol_rel_dual=1e-6, tol_abs_dual=1e-6, tol_abs_primal=1e-6)
    ro = oracle.solve(prob, prost.backend.pdhg(stepsize="boyd", residual_iter=10), o, np.float64)
    assert ro["iters"] == result["iters"]
    assert np.array_equal(np.asarray(ro["x"]).reshape(-1), u)


def _compare_with_oracle(describe, args, prec, dtype, iters, solve_iters, tol=0.0):
    """iterates after `iters` iterations and a complete prost.solve (callback schedule, stopping test) against oracle.Solver /
    oracle.solve on the SAME description; tol > 0: relative to the vector's scale instead of bit for bit (operators with long sparse
    rows, whose products cooperating lanes sum in another order than the oracle's row loop -- cuSPARSE defines none)"""
    def same(a, b):
        a, b = np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)
        return np.array_equal(a, b) if tol == 0 else float(np.abs(a - b).max()) <= tol * max(1.0, float(np.abs(b).max()))
    import oracle
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        d = describe(*args)
        prob, backend = d[0], d[1]
        o = prost.options(max_iters=1000, num_cback_calls=0, verbose=False)
        paths = set()
        for k in iters:
            s = prost.Solver(prob, backend, o)
            s.iterate(k)
            st = s.state()
            s.destroy()
            paths.add(st["path"])
            prob.finalize()
            orc = oracle.Solver(prob.data, prob.nrows, prob.ncols, backend, o, dtype)
            orc.initialize()
            orc.iterate(k)
            ost = orc.state(); ost.update(orc.scalars())
            for v in "xyzw":
                assert same(st[v], ost[v]), (k, v, float(np.abs(st[v] - ost[v]).max()))
            assert (st["tau"] == ost["tau"] and st["sigma"] == ost["sigma"]) if tol == 0 else np.isclose(st["tau"], ost["tau"], rtol=1e-6), (k, st["tau"], ost["tau"])
        d = describe(*args)
        prob, backend, opts = d[0], d[1], dict(d[2])
        opts["max_iters"] = solve_iters
        res = prost.solve(prob, backend, opts)
        d2 = describe(*args)
        ro = oracle.solve(d2[0], d2[1], opts, dtype)
        assert res["result"] == ro["result"] and (int(res["iters"]) == int(ro["iters"]) if tol == 0 else abs(int(res["iters"]) - int(ro["iters"])) <= 3), \
            (res["result"], res["iters"], ro["result"], ro["iters"])
        for v in "xyzw":
            got, exp = np.asarray(res[v]).reshape(-1), np.asarray(ro[v]).reshape(-1)
            assert same(got, exp) if tol == 0 else float(np.abs(got - exp).max()) <= 50 * tol * max(1.0, float(np.abs(exp).max())), v
        return paths, res
    finally:
        prost.set_precision("double")


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_example_multilabel_fast_as_written_matches_the_oracle(prec, dtype):
    """example_multilabel_fast.m:21-54 as written: sparse gradient over 3 labels + the kron(ones(1, L), speye) sum row on a second dual
    variable, ind_geq0 with a linear term, the zero function with a linear term, vectorial TV over 2 L components; boyd / residual_iter 10"""
    import multilabel_fast as ex
    paths, res = _compare_with_oracle(ex.describe, (28, 24), prec, dtype, (1, 2, 25), 400)
    assert paths == {"pdhg:generic"}
    lab = np.asarray(res["x"]).reshape(3, 28, 24)
    assert np.isfinite(lab).all() and lab.min() >= 0


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_example_multilabel_tight_as_written_matches_the_oracle(prec, dtype):
    """example_multilabel_tight.m:42-94 as written: two primal and three dual variables, gradient2d + two sparse_kron_id blocks + identity"""
    import multilabel_tight as ex
    paths, res = _compare_with_oracle(ex.describe, (20, 12), prec, dtype, (1, 2, 25), 400)
    assert paths == {"pdhg:generic"}


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_example_deblurring_as_written_matches_the_oracle(prec, dtype):
    """example_deblurring.m:10-41 as written: kron(speye(nc), convmtx2(kernel)) and spmat_gradient2d as sparse matrices on two constrained
    variables (the blur matrix's rows hold up to klen entries: summed in CSR order), square data term with a per-element b, the default
    backend options boyd / residual_iter 1"""
    import deblurring as ex
    # (round 4 compared this example with a tolerance: the blur matrix is a FULL convolution between two geometries, stayed CSR, and rows of
    # more than 6 entries are summed by cooperating lanes.  With the anchored row patterns of round 5 its products walk every row in CSR
    # order -- the oracle's -- and the example is bit for bit like the other two.)
    paths, res = _compare_with_oracle(lambda nx, ny, nc: ex.describe(nx, ny, nc, klen=5), (24, 16, 2), prec, dtype, (1, 2, 31), 300, tol=0.0)
    assert paths == {"pdhg:generic"}


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_example_tvl1_as_written_matches_the_oracle(prec, dtype):
    """example_tvl1.m:21-53 as written (abs data term, vectorial TV through block.gradient2d, boyd / residual_iter 10, tolerances 1e-7): iterates
    after 1, 2 and 25 iterations and the complete solve -- callback schedule, stopping iteration, x, y, z, w -- bit for bit with the oracle
    (round 5 checked only that the noise goes down)"""
    import tvl1_salt_and_pepper as ex
    paths, res = _compare_with_oracle(lambda nx, ny, nc: ex.describe(nx, ny, nc)[:3], (48, 32, 1), prec, dtype, (1, 2, 25), 600)
    assert paths == {"pdhg:fused-grad2d"}
    paths, res = _compare_with_oracle(lambda nx, ny, nc: ex.describe(nx, ny, nc)[:3], (36, 28, 3), prec, dtype, (1, 25), 300)
    assert paths == {"pdhg:fused-grad2d"}


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_example_rof_primaldual_as_written_matches_the_oracle(prec, dtype):
    """example_rof_primaldual.m:15-46 as written (RGB: sum_norm2(6, ...), alg2 with gamma = 0.05 lmb, residual_iter 10, 250 callback
    calls) without the gap callback that may end the run early: bit for bit with the oracle"""
    import rof_rgb_gap_callback as ex
    paths, res = _compare_with_oracle(ex.describe, (40, 28, 3), prec, dtype, (1, 2, 25), 500)
    assert paths == {"pdhg:fused-grad2d"}


@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_example_rof_dual_as_written_matches_the_oracle(prec, dtype):
    """example_rof_dual.m:10-41 as written: the DUAL problem as a prost.min_problem over q with the constraint w = -grad' q handed over as
    a sparse block, goldstein / residual_iter 100; the image is the dual variable of that problem"""
    import rof_dual as ex
    paths, res = _compare_with_oracle(lambda nx, ny, nc: ex.describe(nx, ny, nc)[:3], (28, 20, 2), prec, dtype, (1, 2, 25), 450)
    assert paths == {"pdhg:generic"}


def test_example_rof_dual_runs_with_its_gap_callback_and_reads_the_image_from_the_dual_variables():
    import rof_dual as ex
    prost.set_gpu(0)
    prost.set_precision("double")
    result, gaps, img, f = ex.main(nx=40, ny=32, nc=2, max_iters=6000, verbose=False)
    assert result["result"] in ("Stopped by user.", "Converged.", "Reached maximum iterations.")
    assert gaps and abs(gaps[-1]) < abs(gaps[0])
    assert img.shape == (2, 40, 32) and np.isfinite(img).all()
    # the denoised image stays close to the data (lmb = 0.3 is a strong regulariser: within the data's range, smoother than it)
    assert float(np.abs(img.reshape(-1) - f).mean()) < 0.5 and img.std() < f.std()


def test_example_multilabel_callback_reads_the_labelling_and_never_stops_the_run():
    """example_multilabel_callback.m as the interm_cb of example_multilabel_fast.m:60-63"""
    import multilabel_callback as cb
    import multilabel_fast as ex
    prost.set_gpu(0)
    prost.set_precision("double")
    nx, ny, L = 24, 20, 3
    prob, backend, opts, u, f, im = ex.describe(nx, ny, max_iters=200, num_cback_calls=4)
    seen = []
    opts["interm_cb"] = lambda it, x, y: cb.multilabel_callback(it, x, y, ny, nx, L, im, show=lambda it_, im_, lab: seen.append((it_, lab.copy())))
    res = prost.solve(prob, backend, opts)
    assert res["result"] == "Reached maximum iterations." and int(res["iters"]) == 200          # the callback returns false: never "Stopped by user."
    assert len(seen) >= 4 and all(lab.shape == (L, nx, ny) and np.isfinite(lab).all() for _, lab in seen)
    assert np.array_equal(seen[-1][1].reshape(-1), np.asarray(res["x"]).reshape(-1)[:nx * ny * L])
