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


# ---- the ELEM_OPERATION surface: user-written functors through prost::ProxElemOperation<T, OP> (tests/plugins/elem_operations.hip) ----

def _coeffs(*vals):
    """a, b, c, d, e, alpha, beta with the defaults of sum_1d.m:35-77 / sum_norm2.m for the ones left out"""
    vals = tuple(vals) + (1, 0, 1, 0, 0, 0, 0)[len(vals):]
    return [np.atleast_1d(np.asarray(v, dtype=np.float64)).ravel() for v in vals]


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_plugins_written_with_the_reference_iterator_signatures(hip, plugin, prec, dtype):
    """include/prost/compat/thrust_ranges.hpp: a block and a prox whose EvalLocalAdd / EvalAdjointLocalAdd / EvalLocal take
    thrust::device_vector<T>::iterator ranges and call thrust::transform on them -- the reference's plugin contract (block.hpp:66-77,
    prox.hpp:117-126), bodies as a user of the reference writes them (tests/plugins/thrust_style_plugins.hip) -- evaluate like the
    library's block.identity / sum_1d('abs') and like the oracle, in eval_linop / eval_prox and inside a PDHG solve"""
    prost.set_gpu(0)
    prost.set_precision(prec)
    t_block = lambda n, scale: (lambda row, col, nrows, ncols: [["test:thrust:scaled_identity", row, col, [n, float(scale)]], [n, n]])
    t_prox = lambda lmb: (lambda idx, count: ["test:thrust:soft_threshold", idx, count, True, [float(lmb)]])
    try:
        now = prost.registered()
        assert "test:thrust:scaled_identity" in now["block"] and "test:thrust:soft_threshold" in now["prox"]
        n, scale, lmb = 1003, 0.5, 0.75
        rng = np.random.default_rng(3)
        x = rng.standard_normal(n).astype(dtype).astype(np.float64)
        for transpose in (False, True):
            got, rowsum, colsum, _ = prost.eval_linop([t_block(n, scale)(0, 0, n, n)[0]], x, transpose)
            exp, erow, ecol, _ = prost.eval_linop([prost.block.diags(n, n, [scale], [0])(0, 0, n, n)[0]], x, transpose)
            assert np.array_equal(got, exp) and np.array_equal(rowsum, erow) and np.array_equal(colsum, ecol)
        Tau = (0.5 + rng.random(n)).astype(dtype).astype(np.float64)
        arg = (2 * rng.standard_normal(n)).astype(dtype).astype(np.float64)
        for conj in (False, True):
            f_plug = prost.function.conjugate(t_prox(lmb)) if conj else t_prox(lmb)
            f_lib = prost.function.sum_1d("abs", 1, 0, lmb)
            f_lib = prost.function.conjugate(f_lib) if conj else f_lib
            got, _ = prost.eval_prox(f_plug, arg, 0.4, Tau)
            orc = oracle.eval_prox(f_lib, arg, 0.4, Tau, dtype)
            assert np.array_equal(got, np.asarray(orc, dtype=np.float64)), conj
        # inside a solve: [gradient2d ; thrust-style block], prox_fstar on the second dual variable = the thrust-style prox
        nx, ny = 24, 36
        f = synthetic.rof_image(nx, ny, 1, 11)

        def problem(plug):
            npx = nx * ny
            u, q, r = prost.variable(npx), prost.variable(2 * npx), prost.variable(npx)
            prob = prost.min_max_problem([u], [q, r])
            prob.add_function(u, prost.function.sum_1d("square", 1, f, 10.0))
            prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
            prob.add_function(r, t_prox(0.3) if plug else prost.function.sum_1d("abs", 1, 0, 0.3))
            prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, 1))
            prob.add_dual_pair(u, r, t_block(npx, 0.5) if plug else prost.block.diags(npx, npx, [0.5], [0]))
            return prob
        b = prost.backend.pdhg(stepsize="boyd", residual_iter=4)
        o = prost.options(max_iters=100, num_cback_calls=0, verbose=False)
        ref = problem(False)
        ref.finalize()
        for k in (1, 23):
            s = prost.Solver(problem(True), b, o)
            s.iterate(k)
            st = s.state()
            s.destroy()
            orc = oracle.Solver(ref.data, ref.nrows, ref.ncols, b, o, dtype)
            orc.initialize()
            orc.iterate(k)
            ost = orc.state()
            for v in "xyzw":
                assert np.array_equal(st[v], ost[v]), (k, v, float(np.abs(st[v] - ost[v]).max()))
    finally:
        prost.set_precision("double")


def plugin_norm2(name, dim, interleaved, *coeffs):
    """front-end builder in the style of sum_norm2.m:83-86, naming a plugin operation"""
    return lambda idx, count: [name, idx, count, False, [count // dim, dim, bool(interleaved), _coeffs(*coeffs)]]


def plugin_1d(name, *coeffs):
    """sum_1d.m:79-80"""
    return lambda idx, count: [name, idx, count, True, [count, 1, False, _coeffs(*coeffs)]]


def plugin_groups(name, dim, interleaved):
    """sum_ind_simplex.m:7-9: no coefficients"""
    return lambda idx, count: [name, idx, count, False, [count // dim, dim, bool(interleaved)]]


def _coefficient_sets(rng, count, dtype):
    """(a, b, c, d, e, alpha, beta): scalars; per-group vectors; the a == 0 branch of the 1-D operation"""
    r = lambda lo, hi: (lo + (hi - lo) * rng.random(count)).astype(dtype).astype(np.float64)
    return [(1.0, 0.0, 1.0, 0.0, 0.0, 0.25, 0.0),
            (1.5, 0.3, 2.0, -0.2, 0.4, 0.7, 0.1),
            (r(0.5, 2), r(-1, 1), r(0.5, 3), 0.1, r(0, 1), r(0.1, 2), 0.0),
            (0.0, 0.3, 2.0, -0.2, 0.4, 0.7, 0.1)]


def test_elem_operation_headers_are_what_the_reference_exports():
    """CPU: the plugin-facing headers exist under the reference's paths and declare the reference's names (SURVEY 8b bullet 2;
    prox_elem_operation.hpp:32-117, elem_operation.hpp:30-40, vector.hpp:32-63, shared_mem.hpp:28-62)"""
    inc = os.path.join(os.path.dirname(HERE), "include", "prost", "prox")
    for rel, names in {"prox_elem_operation.hpp": ["class ProxElemOperation", "struct ElemOpCoefficients", "kCoeffsCount == 0", "kCoeffsCount != 0"],
                       "prox_elem_operation.inl": ["ProxElemOperationKernel", "CurrentStream()"],
                       "vector.hpp": ["class Vector", "operator[]"],
                       "shared_mem.hpp": ["class SharedMem", "operator[]"],
                       "elemop/elem_operation.hpp": ["struct ElemOperation", "kCoeffsCount", "kDim", "SharedMemType", "GetSharedMemCount"],
                       "elemop/elem_operation_1d.hpp": ["struct ElemOperation1D"],
                       "elemop/elem_operation_norm2.hpp": ["struct ElemOperationNorm2"],
                       "elemop/elem_operation_ind_sum.hpp": ["struct ElemOperationIndSum"],
                       "elemop/elem_operation_ind_simplex.hpp": ["struct ElemOperationIndSimplex", "GetSharedMemCount"],
                       "elemop/function_1d.hpp": ["Function1DHuber", "Function1DLq", "Function1DTruncLinear"]}.items():
        text = open(os.path.join(inc, rel)).read()
        for n in names:
            assert n in text, (rel, n)


def test_elem_operation_plugin_registers_its_operations(plugin):
    for prec in ("single", "double"):
        prost.set_precision(prec)
        reg = prost.registered()["prox"]
        for n in ["test:op:norm2_huber", "test:op:abs_1d", "test:op:simplex_lds", "test:op:partial"] + \
                 ["test:tpl:%s:%s" % (k, f) for k in ("1d", "norm2") for f in prost.function.FUNCTIONS_1D]:
            assert n in reg, n
    prost.set_precision("double")


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_user_written_elem_operations_equal_the_library_and_the_oracle(hip, plugin, prec, dtype):
    """eval_prox of functor operations compiled OUT OF TREE through ProxElemOperation<T, OP> == the library's
    elem_operation:norm2:huber / elem_operation:1d:abs == the oracle, bit for bit: dims 1 / 2 / 3 / 7 (register-tile
    instances for dim <= 4, the one-group-per-lane kernel beyond), both layouts, counts with and without a 16-byte tail,
    scalar and per-group coefficients, and through conjugate() -- which evaluates the operation with invert_tau = true."""
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        rng = np.random.default_rng(17)
        for count in (1024, 1003, 3):
            for dim, interleaved in [(1, False), (2, False), (2, True), (3, False), (3, True), (4, True), (7, False), (7, True)]:
                n = count * dim
                arg = (2 * rng.standard_normal(n)).astype(dtype).astype(np.float64)
                if interleaved:
                    arg[:dim] = 0                       # a zero-norm group
                else:
                    arg[0::count][:dim] = 0
                Tau = (0.5 + rng.random(n)).astype(dtype).astype(np.float64)
                for cs in _coefficient_sets(rng, count, dtype)[:3]:
                    for wrap in (lambda f: f, prost.function.conjugate):
                        got, _ = prost.eval_prox(wrap(plugin_norm2("test:op:norm2_huber", dim, interleaved, *cs)), arg, 0.4, Tau)
                        lib, _ = prost.eval_prox(wrap(prost.function.sum_norm2(dim, interleaved, "huber", *cs)), arg, 0.4, Tau)
                        orc = oracle.eval_prox(wrap(prost.function.sum_norm2(dim, interleaved, "huber", *cs)), arg, 0.4, Tau, dtype)
                        assert np.array_equal(got, lib), (count, dim, interleaved)
                        assert np.array_equal(got, orc), (count, dim, interleaved, float(np.abs(got - orc).max()))
            arg = (2 * rng.standard_normal(count)).astype(dtype).astype(np.float64)
            Tau = (0.5 + rng.random(count)).astype(dtype).astype(np.float64)
            for cs in _coefficient_sets(rng, count, dtype):
                for wrap in (lambda f: f, prost.function.conjugate):
                    got, _ = prost.eval_prox(wrap(plugin_1d("test:op:abs_1d", *cs)), arg, 0.4, Tau)
                    orc = oracle.eval_prox(wrap(prost.function.sum_1d("abs", *cs)), arg, 0.4, Tau, dtype)
                    assert np.array_equal(got, orc), (count, float(np.abs(got - orc).max()))
    finally:
        prost.set_precision("double")


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
def test_elem_operation_with_a_shared_memory_slice_and_partial_results(hip, plugin, prec, dtype):
    """SharedMem: a simplex projection that sorts in its per-thread LDS slice (GetSharedMemCount(dim) = dim entries of T)
    == the oracle's elem_operation:ind_simplex, bit for bit, up to dim 40 (40 KiB of LDS per workgroup in fp64);
    kPartialResult: an operation that writes res[0] only leaves the other components of the result as they were."""
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        rng = np.random.default_rng(5)
        for count in (515, 64):
            for dim, interleaved in [(2, False), (3, True), (4, False), (7, True), (33, False), (40, True)]:
                arg = rng.standard_normal(count * dim).astype(dtype).astype(np.float64)
                Tau = np.ones(count * dim)
                got, _ = prost.eval_prox(plugin_groups("test:op:simplex_lds", dim, interleaved), arg, 1.0, Tau)
                orc = oracle.eval_prox(prost.function.sum_ind_simplex(dim, interleaved), arg, 1.0, Tau, dtype)
                assert np.array_equal(got, orc), (count, dim, interleaved, float(np.abs(got - orc).max()))
                grp = got.reshape(count, dim) if interleaved else got.reshape(dim, count).T
                assert np.allclose(grp.sum(axis=1), 1, atol=1e-5 if dtype == np.float32 else 1e-12) and (grp >= 0).all()
        # the PUBLIC ElemOperationIndSum / ElemOperationIndSimplex templates, instantiated out of tree, == the built-in operations
        for count in (1001, 64):
            for dim, interleaved in [(2, False), (3, True), (5, False), (8, True), (19, False)]:
                arg = rng.standard_normal(count * dim).astype(dtype).astype(np.float64)
                Tau = np.ones(count * dim)
                for name, builtin in (("test:tpl:ind_sum", prost.function.sum_ind_sum), ("test:tpl:ind_simplex", prost.function.sum_ind_simplex)):
                    got, _ = prost.eval_prox(plugin_groups(name, dim, interleaved), arg, 1.0, Tau)
                    lib, _ = prost.eval_prox(builtin(dim, interleaved), arg, 1.0, Tau)
                    orc = oracle.eval_prox(builtin(dim, interleaved), arg, 1.0, Tau, dtype)
                    assert np.array_equal(got, lib) and np.array_equal(got, orc), (name, count, dim, interleaved)
        count, dim = 1000, 3
        arg = rng.standard_normal(count * dim).astype(dtype).astype(np.float64)
        got, _ = prost.eval_prox(plugin_groups("test:op:partial", dim, False), arg, 1.0, np.ones(count * dim))
        assert np.array_equal(got[:count], 2 * arg[:count]) and np.array_equal(got[count:], np.zeros(2 * count))   # eval_prox zero-fills the result vector
        # the same operation WITHOUT the kPartialResult declaration (a straight port from the reference): it runs the register-tile
        # path, which preloads res unless the operation opts out (kWritesAllComponents) -- unwritten components keep their content
        for dim, interleaved, count in [(3, False, 1000), (3, True, 1000), (2, False, 4099), (4, True, 64)]:
            arg = rng.standard_normal(count * dim).astype(dtype).astype(np.float64)
            got, _ = prost.eval_prox(plugin_groups("test:op:partial_undeclared", dim, interleaved), arg, 1.0, np.ones(count * dim))
            grp = got.reshape(count, dim) if interleaved else got.reshape(dim, count).T
            agr = arg.reshape(count, dim) if interleaved else arg.reshape(dim, count).T
            assert np.array_equal(grp[:, 0], 2 * agr[:, 0]) and np.array_equal(grp[:, 1:], np.zeros((count, dim - 1))), (dim, interleaved, count)
    finally:
        prost.set_precision("double")


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype,tol", [("single", np.float32, 5e-5), ("double", np.float64, 1e-10)])
def test_public_operation_templates_equal_the_built_in_operations(hip, plugin, prec, dtype, tol):
    """ElemOperation1D / ElemOperationNorm2 over the 14 public Function1D* functors, instantiated out of tree, == the
    library's run-time dispatched operations of the same names == the oracle (bit for bit; lq: Newton on pow(), stated)"""
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        rng = np.random.default_rng(23)
        count = 777
        for fn in prost.function.FUNCTIONS_1D:
            for cs in _coefficient_sets(rng, count, dtype)[1:3]:
                if fn == "lq":                  # q = 1/2: the analytic branch (function_1d.hpp:195-202), as tests/test_gpu_kernels.py::test_prox_elem
                    cs = cs[:5] + (0.5 if np.isscalar(cs[5]) else np.full(count, 0.5), cs[6])
                arg = (2 * rng.standard_normal(count)).astype(dtype).astype(np.float64)
                Tau = (0.5 + rng.random(count)).astype(dtype).astype(np.float64)
                got, _ = prost.eval_prox(plugin_1d("test:tpl:1d:" + fn, *cs), arg, 0.4, Tau)
                lib, _ = prost.eval_prox(prost.function.sum_1d(fn, *cs), arg, 0.4, Tau)
                orc = oracle.eval_prox(prost.function.sum_1d(fn, *cs), arg, 0.4, Tau, dtype)
                assert np.array_equal(got, lib), fn            # device against device: the same header, the same bits
                if fn == "lq":                                  # device against host libm (pow / sin / acos)
                    assert np.allclose(got, orc, rtol=tol, atol=tol), float(np.abs(got - orc).max())
                else:
                    assert np.array_equal(got, orc), fn
                for dim, interleaved in [(2, False), (3, True), (7, False)]:
                    arg = (2 * rng.standard_normal(count * dim)).astype(dtype).astype(np.float64)
                    Tau = (0.5 + rng.random(count * dim)).astype(dtype).astype(np.float64)
                    got, _ = prost.eval_prox(plugin_norm2("test:tpl:norm2:" + fn, dim, interleaved, *cs), arg, 0.4, Tau)
                    lib, _ = prost.eval_prox(prost.function.sum_norm2(dim, interleaved, fn, *cs), arg, 0.4, Tau)
                    orc = oracle.eval_prox(prost.function.sum_norm2(dim, interleaved, fn, *cs), arg, 0.4, Tau, dtype)
                    assert np.array_equal(got, lib), (fn, dim, interleaved)
                    if fn == "lq":
                        assert np.allclose(got, orc, rtol=tol, atol=tol), float(np.abs(got - orc).max())
                    else:
                        assert np.array_equal(got, orc), (fn, dim, interleaved)
    finally:
        prost.set_precision("double")


def _huber_tv_problem(nx, ny, f, lmb, alpha, plugin_ops, conj):
    """TV-L1-like saddle-point problem with a Huber function on the dual variable q of the gradient; a 3-entry auxiliary
    pair (w, r) comes first in both vectors, so q starts at offset 3 of the dual vector: the operation's operands are not
    16-byte aligned."""
    n = nx * ny
    u, w = prost.variable(n), prost.variable(3)
    r, q = prost.variable(3), prost.variable(2 * n)
    prob = prost.min_max_problem([u, w], [r, q])
    one = plugin_1d("test:op:abs_1d", 1, f, lmb) if plugin_ops else prost.function.sum_1d("abs", 1, f, lmb)
    prob.add_function(u, one)
    hub = plugin_norm2("test:op:norm2_huber", 2, False, 1, 0, 1, 0, 0, alpha, 0) if plugin_ops else \
        prost.function.sum_norm2(2, False, "huber", 1, 0, 1, 0, 0, alpha, 0)
    prob.add_function(q, prost.function.conjugate(hub) if conj else hub)
    prob.add_function(r, prost.function.sum_1d("square", 1, 0, 1))
    prob.add_dual_pair(w, r, prost.block.identity(3))
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, 1))
    return prob


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
@pytest.mark.parametrize("conj", [False, True])
def test_pdhg_solve_with_user_written_elem_operations_matches_the_oracle(hip, plugin, prec, dtype, conj):
    """a PDHG run whose prox_g and prox_fstar are functor operations compiled out of tree (the dual one at an offset of 3
    entries: unaligned operands -> the VEC = 1 instances) == the oracle running elem_operation:1d:abs /
    elem_operation:norm2:huber in their place: x, y, z, w after 1, 7, 30 iterations bit for bit, and a complete solve
    stops at the same iteration with the same result"""
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        nx, ny = 20, 34
        f = synthetic.rof_image(nx, ny, 1, 3)
        b = prost.backend.pdhg(stepsize="boyd", residual_iter=3)
        o = prost.options(max_iters=60, num_cback_calls=0, verbose=False)
        prob = _huber_tv_problem(nx, ny, f, 2.0, 0.05, True, conj)
        ref = _huber_tv_problem(nx, ny, f, 2.0, 0.05, False, conj)
        ref.finalize()
        for k in (1, 7, 30):
            s = prost.Solver(prob, b, o)
            s.iterate(k)
            st = s.state()
            s.destroy()
            orc = oracle.Solver(ref.data, ref.nrows, ref.ncols, b, o, dtype)
            orc.initialize()
            orc.iterate(k)
            ost = orc.state()
            for v in "xyzw":
                assert np.array_equal(st[v], ost[v]), (k, v, float(np.abs(st[v] - ost[v]).max()))
        o2 = prost.options(max_iters=2000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=1e-3, tol_abs_dual=1e-3)
        got, exp = prost.solve(prob, b, o2), oracle.solve(ref, b, o2, dtype)
        assert got["result"] == exp["result"] and got["iters"] == exp["iters"]
        assert np.array_equal(np.asarray(got["x"]), np.asarray(exp["x"]))
    finally:
        prost.set_precision("double")


@pytest.mark.gpu
@pytest.mark.parametrize("prec,dtype", [("single", np.float32), ("double", np.float64)])
@pytest.mark.parametrize("step,residual_iter", [("boyd", 1), ("boyd", 3), ("goldstein", 1)])
def test_user_written_elem_operations_take_their_step_from_the_device(hip, plugin, prec, dtype, step, residual_iter):
    """round 5 (Prox::StepView): the reference's DEFAULT options (boyd, residual_iter 1) on a problem whose proxes are functor operations
    compiled out of tree -- next to in-tree ones in the same lists -- run the step-size rule and the stopping test on the device: the
    argument passes and the plugin's kernels read tau / sigma from the record, no host wait per iteration.  Iterates and step sizes ==
    the host rule's == the oracle's, bit for bit; a complete solve stops at the same iteration."""
    prost.set_gpu(0)
    prost.set_precision(prec)
    try:
        nx, ny = 20, 34
        f = synthetic.rof_image(nx, ny, 1, 3)
        prob = _huber_tv_problem(nx, ny, f, 2.0, 0.05, True, False)
        ref = _huber_tv_problem(nx, ny, f, 2.0, 0.05, False, False)
        ref.finalize()
        o = prost.options(max_iters=80, num_cback_calls=0, verbose=False)
        for k in (5, 9, 41):          # (budgets below 3 iterations keep the host loop)
            st = {}
            for dev in (True, False):
                b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
                b[1]["allow_device_rules"] = dev
                s = prost.Solver(prob, b, o)
                s.iterate(k)
                st[dev] = s.state()
                s.destroy()
            assert st[True]["device_rule_batches"] > 0 and st[False]["device_rule_batches"] == 0, (st[True]["device_rule_batches"], st[False]["device_rule_batches"])
            orc = oracle.Solver(ref.data, ref.nrows, ref.ncols, prost.backend.pdhg(stepsize=step, residual_iter=residual_iter), o, dtype)
            orc.initialize()
            orc.iterate(k)
            ost, osc = orc.state(), orc.scalars()
            for dev in (True, False):
                for v in "xyzw":
                    assert np.array_equal(st[dev][v], ost[v]), (k, dev, v, float(np.abs(st[dev][v] - ost[v]).max()))
                assert st[dev]["tau"] == osc["tau"] and st[dev]["sigma"] == osc["sigma"], (k, dev)
        o2 = prost.options(max_iters=2000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-3, tol_rel_dual=1e-3, tol_abs_primal=1e-3, tol_abs_dual=1e-3)
        b = prost.backend.pdhg(stepsize=step, residual_iter=residual_iter)
        got, exp = prost.solve(prob, b, o2), oracle.solve(ref, b, o2, dtype)
        assert got["result"] == exp["result"] and got["iters"] == exp["iters"]
        assert np.array_equal(np.asarray(got["x"]), np.asarray(exp["x"]))
    finally:
        prost.set_precision("double")
