"""The prost::Block / prost::Prox plugin API from OUTSIDE the library (reference mechanism:
matlab/+prost/private/custom.cpp:11-28 + cmake/CustomSources.cmake.example:1-26).

tests/plugins/ holds two user-side sources that are NOT part of libprost.so: a linear-operator block (plain C++, calls a
prost_hip_* entry point from EvalLocalAdd) and a proximal operator with its own gfx950 kernel (hipcc).  They are compiled
here, by this test, against include/ + prost_amd/lib/*.so only, register themselves in Factory<T>::block_reg() /
prox_reg() from static initialisers and are loaded with the `load_plugin` command.  The GPU tests then use them through the
unchanged front end: eval_linop, eval_prox and a PDHG solve that mixes the plugin block with block.gradient2d -- compared
with the ORACLE running the library's own equivalents (block.diags / sum_1d('abs')) in their place.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
PLUGIN_DIR = os.path.join(HERE, "plugins")


@pytest.fixture(scope="module")
def plugin():
    subprocess.check_call(["make", "-C", PLUGIN_DIR], stdout=subprocess.DEVNULL)
    path = os.path.join(PLUGIN_DIR, "build", "libprost_test_plugins.so")
    before = prost.registered()
    prost.load_plugin(path)
    return path, before


def scaled_identity(n, scale):
    """front-end builder of the plugin block, in the style of matlab/+prost/+block/*.m"""
    return lambda row, col, nrows, ncols: [["test:scaled_identity", row, col, [n, float(scale)]], [n, n]]


def soft_threshold(lmb):
    """front-end builder of the plugin prox, in the style of matlab/+prost/+function/*.m (diagsteps = true like sum_1d)"""
    return lambda idx, count: ["test:soft_threshold", idx, count, True, [float(lmb)]]


def test_plugin_builds_against_the_public_headers_and_registers_itself(plugin):
    """CPU: the plugin links against include/ + the two shared libraries only, and its static initialisers reach the
    registries of the running libprost.so (both precisions)"""
    path, before = plugin
    assert "test:scaled_identity" not in before["block"] and "test:soft_threshold" not in before["prox"]
    for prec in ("single", "double"):
        prost.set_precision(prec)
        now = prost.registered()
        assert "test:scaled_identity" in now["block"] and "test:soft_threshold" in now["prox"]
        assert set(before["block"]) <= set(now["block"]) and set(before["prox"]) <= set(now["prox"])
    prost.set_precision("double")
    with pytest.raises(prost.ProstError, match="cannot load"):
        prost.load_plugin(os.path.join(PLUGIN_DIR, "build", "no_such_plugin.so"))
    # the plugin's objects reference nothing but the public interface: every undefined prost symbol it needs is exported
    # by libprost.so / libprost_hip.so (checked by the dynamic loader above: RTLD_NOW resolves all of them)
    out = subprocess.run(["nm", "-D", "--undefined-only", "-C", path], capture_output=True, text=True).stdout
    assert "prost::Factory<float>::block_reg" in out and "prost::CurrentStream()" in out and "prost_hip_axpy_f32" in out


def _mixed_problem(nx, ny, scale, lmb, plugin_ops, f):
    """min_u max_{q, r}  <grad u, q> + <scale u, r> + 5 |u - f|^2 - ind(|q| <= 1) - lmb |r|_1 : the ROF description with one
    more dual variable r coupled through the scaled identity and penalised by lmb |r|_1 (its prox = soft thresholding)"""
    n = nx * ny
    u, q, r = prost.variable(n), prost.variable(2 * n), prost.variable(n)
    prob = prost.min_max_problem([u], [q, r])
    prob.add_function(u, prost.function.sum_1d("square", 1, f, 10.0))
    prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
    prob.add_function(r, soft_threshold(lmb) if plugin_ops else prost.function.sum_1d("abs", 1, 0, lmb))
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, 1))
    prob.add_dual_pair(u, r, scaled_identity(n, scale) if plugin_ops else prost.block.diags(n, n, [scale], [0]))
    return prob


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_plugin_block_and_prox_evaluate_like_the_library_equivalents(hip, plugin, prec, dtype):
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        n, scale, lmb = 1000, 0.5, 0.75
        rng = np.random.default_rng(3)
        x = rng.standard_normal(n).astype(dtype).astype(np.float64)
        for transpose in (False, True):
            got, rowsum, colsum, _ = prost.eval_linop([scaled_identity(n, scale)(0, 0, n, n)[0]], x, transpose)
            exp, erow, ecol, _ = prost.eval_linop([prost.block.diags(n, n, [scale], [0])(0, 0, n, n)[0]], x, transpose)
            assert np.array_equal(got, exp) and np.array_equal(got, (dtype(scale) * x.astype(dtype)).astype(np.float64))
            assert np.array_equal(rowsum, erow) and np.array_equal(colsum, ecol)          # Block::row_sum / col_sum of the plugin
        Tau = (0.5 + rng.random(n)).astype(dtype).astype(np.float64)
        arg = (2 * rng.standard_normal(n)).astype(dtype).astype(np.float64)
        arg[:3] = [0.0, lmb * 0.3, -lmb * 0.3]
        got, _ = prost.eval_prox(soft_threshold(lmb), arg, 0.4, Tau)
        exp, _ = prost.eval_prox(prost.function.sum_1d("abs", 1, 0, lmb), arg, 0.4, Tau)
        assert np.array_equal(got, exp)
        orc = oracle.eval_prox(prost.function.sum_1d("abs", 1, 0, lmb), arg, 0.4, Tau, dtype)
        assert np.array_equal(got, np.asarray(orc, dtype=np.float64))
    finally:
        prost.set_precision("double")


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
@pytest.mark.parametrize("step", ["alg2", "boyd"])
def test_pdhg_with_plugin_block_and_prox_matches_the_oracle(hip, plugin, prec, dtype, step):
    """generic PDHG path: the operator is [gradient2d; plugin block], the preconditioners come from the plugin's row_sum /
    col_sum, prox_fstar on the second dual variable is the plugin's kernel.  Iterates == oracle (which runs block.diags and
    sum_1d('abs') in their place), bit for bit; and a full prost.solve stops at the same iteration."""
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        nx, ny, scale, lmb = 24, 36, 0.5, 0.3
        f = synthetic.rof_image(nx, ny, 1, 11)
        b = prost.backend.pdhg(stepsize=step, residual_iter=4, alg2_gamma=0.5)
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
        prob = _mixed_problem(nx, ny, scale, lmb, True, f)
        ref = _mixed_problem(nx, ny, scale, lmb, False, f)
        ref.finalize()
        for k in (1, 9, 40):
            s = prost.Solver(prob, b, o)
            s.iterate(k)
            st = s.state()
            s.destroy()
            assert st["path"] == "pdhg:generic"
            orc = oracle.Solver(ref.data, ref.nrows, ref.ncols, b, o, dtype)
            orc.initialize()
            orc.iterate(k)
            ost, osc = orc.state(), orc.scalars()
            for v in "xyzw":
                assert np.array_equal(st[v], ost[v]), (k, v, float(np.abs(st[v] - ost[v]).max()))
            for v in ("tau", "sigma"):
                assert st[v] == osc[v], (k, v)
        info = prost.problem_info(prob)
        rinfo = prost.problem_info(ref)
        assert np.array_equal(info["scaling_left"], rinfo["scaling_left"]) and np.array_equal(info["scaling_right"], rinfo["scaling_right"])
        o2 = prost.options(max_iters=3000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=1e-3, tol_abs_dual=1e-3)
        got, exp = prost.solve(prob, b, o2), prost.solve(ref, b, o2)
        assert got["result"] == exp["result"] == "Converged." and got["iters"] == exp["iters"]
        assert np.array_equal(np.asarray(got["x"]), np.asarray(exp["x"]))
    finally:
        prost.set_precision("double")
