"""Front-end description builders (prost_amd = Python mirror of matlab/+prost): index bookkeeping,
cell layouts and error behaviour must match the MATLAB package (min_max_problem.m, min_problem.m,
private/add_prox.m, +block/*.m, +function/*.m, options.m, +backend/*.m)."""
import numpy as np
import pytest

import prost_amd as prost


def test_rof_description_matches_example_rof_primaldual():
    nx, ny, nc = 5, 4, 3
    f = np.linspace(0, 1, nx * ny * nc)
    u = prost.variable(nx * ny * nc)
    q = prost.variable(2 * nx * ny * nc)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", 1, f, 10))
    prob.add_function(q, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, nc))
    assert (prob.nrows, prob.ncols) == (2 * nx * ny * nc, nx * ny * nc)
    name, idx, size, diagsteps, data = prob.data["prox_g"][0]
    assert (name, idx, size, diagsteps) == ("elem_operation:1d:square", 0, nx * ny * nc, True)      # sum_1d.m:79-80
    assert data[:3] == [nx * ny * nc, 1, False] and np.array_equal(data[3][1], f) and data[3][2][0] == 10
    name, idx, size, diagsteps, data = prob.data["prox_fstar"][0]
    assert (name, idx, size, diagsteps) == ("elem_operation:norm2:ind_leq0", 0, 2 * nx * ny * nc, False)   # sum_norm2.m:85-86
    assert data[:3] == [nx * ny, 2 * nc, False]
    assert prob.data["linop"] == [["gradient2d", 0, 0, [nx, ny, nc, False]]]                         # gradient2d.m:12-14
    assert prob.data["scaling"] == "alpha" and prob.data["scaling_alpha"] == 1                         # problem.m:10


def test_variable_offsets_and_sub_variables():
    a, b = prost.variable(10), prost.variable(6)
    b1, b2 = prost.sub_variable(b, 2), prost.sub_variable(b, 4)
    y = prost.variable(7)
    prob = prost.min_max_problem([a, b], [y])
    assert (a.idx, b.idx, b1.idx, b2.idx, y.idx) == (0, 10, 10, 12, 0)
    prob.add_function(b2, prost.function.sum_1d("abs"))
    assert prob.data["prox_g"][0][1:3] == [12, 4]
    prob.add_dual_pair(b1, y, prost.block.zero())
    assert prob.data["linop"][0][:3] == ["zero", 0, 10]
    bad = prost.variable(5)
    prost.sub_variable(bad, 2)
    with pytest.raises(ValueError, match="Size of subvariables"):
        prost.min_max_problem([bad], [y])


def test_add_prox_replaces_same_index_and_block_replaced():
    u, q = prost.variable(8), prost.variable(8)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("abs"))
    prob.add_function(u, prost.function.sum_1d("square"))
    assert len(prob.data["prox_g"]) == 1 and prob.data["prox_g"][0][0].endswith("square")     # add_prox.m
    prob.add_dual_pair(u, q, prost.block.identity(2))
    prob.add_dual_pair(u, q, prost.block.identity(3))
    assert len(prob.data["linop"]) == 1 and prob.data["linop"][0][3][2][0] == 3               # min_max_problem.m:166-183
    with pytest.raises(ValueError, match="Size of block"):
        prob.add_dual_pair(u, q, prost.block.gradient2d(2, 2, 1))
    with pytest.raises(ValueError, match="not registered"):
        prob.add_function(prost.variable(3), prost.function.zero())


def test_finalize_and_fill_variables():
    u, q = prost.variable(4), prost.variable(6)
    s1, s2 = prost.sub_variable(q, 2), prost.sub_variable(q, 4)
    prob = prost.min_max_problem([u], [q])
    prob.finalize()
    assert prob.data["prox_g"] == [["zero", 0, 4, True, []]] and prob.data["prox_fstar"] == [["zero", 0, 6, True, []]]
    prob.fill_variables({"x": np.arange(4.0), "y": np.arange(6.0) + 10})
    assert np.array_equal(u.val, np.arange(4.0)) and np.array_equal(s2.val, [12, 13, 14, 15]) and np.array_equal(s1.val, [10, 11])
    x, z = prost.variable(3), prost.variable(5)
    mp = prost.min_problem([x], [z])
    mp.add_function(z, prost.function.sum_1d("abs"))
    mp.add_constraint(x, z, prost.block.zero())
    mp.finalize()
    assert mp.data["prox_f"][0][0].endswith("abs") and mp.data["prox_g"][0][0] == "zero" and not mp.data["prox_fstar"]
    mp.fill_variables({"x": np.ones(3), "z": np.arange(5.0)})
    assert np.array_equal(z.val, np.arange(5.0))


def test_conjugate_epi_quad_defaults_and_options():
    f = prost.function.conjugate(prost.function.sum_1d("abs", 1, 0.5))(3, 9)
    assert f[0] == "moreau" and f[1:4] == [3, 9, True] and f[4][0][0] == "elem_operation:1d:abs"    # conjugate.m:7-15
    e = prost.function.sum_ind_epi_quad(3, False, 1.0, np.zeros(8), 0.0)(0, 12)
    assert e[0] == "ind_epi_quad" and e[3] is False and e[4][:3] == [4, 3, False]                   # sum_ind_epi_quad.m:17-20
    b = prost.backend.pdhg()
    assert b[0] == "pdhg" and b[1]["stepsize"] == "boyd" and b[1]["residual_iter"] == 1 and b[1]["arb_delta"] == 1.05   # pdhg.m:4-14
    a = prost.backend.admm(rho0=15)
    assert a[1]["rho0"] == 15 and a[1]["alpha"] == 1.7 and a[1]["cg_max_iter"] == 10                # admm.m:4-13
    o = prost.options(max_iters=7)
    assert o["max_iters"] == 7 and o["tol_rel_primal"] == 1e-4 and o["num_cback_calls"] == 10       # options.m:4-14
    with pytest.raises(ValueError):
        prost.options(bogus=1)
    with pytest.raises(ValueError):
        prost.backend.pdhg(step="alg1")


def test_large_results_are_views_that_keep_their_value_alive():
    """Result matrices above the view threshold are numpy views of the prost_value's storage (no second copy of the
    10^7..10^8-element result vectors); the value tree lives until the last view is gone."""
    import gc
    from prost_amd import _capi
    L = _capi.lib()
    n = _capi._VIEW_THRESHOLD + 17
    src = np.arange(n, dtype=np.float64)
    keep = []
    v = _capi.to_value([src, "tag", np.ones(3)], keep)
    owner = _capi._ValueOwner(v)
    out = _capi.from_value(v, owner)
    assert owner.used and out[1] == "tag" and np.array_equal(out[2], np.ones(3))
    big = out[0]
    assert not big.flags.owndata and np.array_equal(big, src)
    tail = big[-5:]
    del out, big, owner
    gc.collect()
    assert np.array_equal(tail, src[-5:])          # the view keeps the tree alive
    del tail
    gc.collect()
    small = _capi.from_value(_capi.to_value(np.ones(4), keep))
    assert small.flags.owndata


def test_mex_gateway_source_compiles_against_the_mex_api_declarations():
    """mex/prost_mex.cpp (the MATLAB gateway over libprost.so; reference matlab/+prost/private/prost.cpp:305-347) cannot be
    linked here -- no MATLAB, no mex.h -- but it must stay a compilable translation unit: g++ -fsyntax-only against
    tests/mex_decl.h (declarations of the MEX API functions it calls) and the real include/prost_c.h."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "mex", "prost_mex.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-include", os.path.join(root, "tests", "mex_decl.h"),
                        "-I", os.path.join(root, "include"), src], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    text = open(src).read()
    # both directions of the marshalling and the three callbacks of the reference gateway are defined, not just declared
    for needle in ("prost_value* convert(const mxArray* a) {", "mxArray* back(const prost_value* v) {", "int stop_cb(void*) {",
                   "int interm_cb(void* user", "void mexFunction(int nlhs, mxArray** plhs, int nrhs, const mxArray** prhs) {",
                   # library output -> mexPrintf with the pause(.001) flush (prost.cpp:15-44), installed per call
                   "void print_cb(void*, const char* text, size_t n) {", 'mexEvalString("pause(.001);")', "ScopedOutputRedirect redirect;",
                   # result structs are copied field by field, whatever the command returns (pair_launches, comm_info, ...)
                   "prost_value_field_count(v)", "prost_value_field_name(v, i)"):
        assert needle in text, needle
    assert "kResultFields" not in text          # no hard-coded field list that a new result field could fall through
    # the declarations header stays declarations: no function bodies
    decl = open(os.path.join(root, "tests", "mex_decl.h")).read()
    assert "{" not in decl.split('extern "C" {', 1)[1].rsplit("}", 1)[0]


def test_sparse_block_builders_do_not_touch_the_callers_matrix():
    """block.sparse / sparse_kron_id canonicalise a COPY (eliminate_zeros / sort_indices work in place, and csc_matrix(K) of a
    float64 CSC matrix is K itself)"""
    import scipy.sparse as sp
    K = sp.csc_matrix((np.array([1.0, 0.0, 2.0, 3.0]), np.array([1, 0, 0, 1]), np.array([0, 2, 4])), shape=(2, 2))
    data, indices, nnz = K.data.copy(), K.indices.copy(), K.nnz
    cell, sz = prost.block.sparse(K)(0, 0, 2, 2)
    assert K.nnz == nnz and np.array_equal(K.data, data) and np.array_equal(K.indices, indices)
    assert cell[3][0].nnz == 3 and cell[3][0].has_sorted_indices
    prost.block.sparse_kron_id(K, 2)(0, 0, 4, 4)
    assert K.nnz == nnz and np.array_equal(K.indices, indices)


def test_host_transport_callback_failures_surface_in_the_next_command():
    """an exception inside the host all-reduce / point-to-point callback (a ctypes callback on a HIP runtime thread: it could only
    be printed) is recorded, poisons what the callback was to produce and is raised by the next command"""
    from prost_amd import _capi
    del _capi._callback_error[:]
    _capi._callback_error.append(RuntimeError("peer 1 went away"))
    with pytest.raises(prost.ProstError, match="host-transport callback failed: RuntimeError: peer 1 went away"):
        prost.get_precision() if hasattr(prost, "get_precision") else _capi.command("get_precision", (), 1)
    assert not _capi._callback_error
    _capi.command("get_precision", (), 1)       # raised once


def test_teardown_commands_run_before_a_pending_callback_error_is_raised(monkeypatch):
    """clean-up after a dead peer must not leak the native solver / communicator: release, comm_destroy and solver_destroy EXECUTE and
    only then raise the stored callback error (every other command raises it without running)"""
    from prost_amd import _capi
    ran = []

    class FakeLib:
        def prost_command(self, cmd, nlhs, plhs, nrhs, prhs):
            ran.append(cmd.decode())
            return 0

        def prost_value_free(self, v):
            pass

    monkeypatch.setattr(_capi, "lib", lambda: FakeLib())
    for cmd in ("release", "comm_destroy", "solver_destroy"):
        del _capi._callback_error[:]
        _capi._callback_error.append(RuntimeError("peer 1 went away"))
        with pytest.raises(prost.ProstError, match="peer 1 went away"):
            _capi.command(cmd, (), 0)
        assert ran[-1] == cmd and not _capi._callback_error          # ran first, raised afterwards, once
    _capi._callback_error.append(RuntimeError("peer 1 went away"))
    n = len(ran)
    with pytest.raises(prost.ProstError):
        _capi.command("solver_iterate", (), 0)
    assert len(ran) == n                                            # a consuming command does not run


def test_example_rof_primal_description_matches_the_matlab_script():
    """examples/rof_primal_sub_variables.py builds what example_rof_primal.m:15-36 builds: a min_problem whose primal variable carries
    three sum_1d pieces on consecutive sub-variables (100, 500, the rest; slices of f as coefficient b), the norm2 'abs' regulariser on
    the constrained variable, one sparse block, boyd / residual_iter = 1 (no GPU needed for the description)"""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import rof_primal_sub_variables as ex
    nx, ny, nc = 40, 30, 3
    prob, backend, u, f, grad, lmb = ex.describe(nx, ny, nc)
    prob.finalize()
    n = nx * ny * nc
    assert prob.ncols == n and prob.nrows == 2 * n
    pieces = prob.data["prox_g"]
    assert len(pieces) == 3 and not prob.data["prox_fstar"] and len(prob.data["prox_f"]) == 1
    assert [(p[1], p[2]) for p in pieces] == [(0, 100), (100, 500), (600, n - 600)]
    assert [b[0] for b in prob.data["linop"]] == ["sparse"] and grad.shape == (2 * n, n)
    assert backend[0] == "pdhg" and backend[1]["stepsize"] == "boyd" and backend[1]["residual_iter"] == 1

