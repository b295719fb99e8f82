"""The reference's own MATLAB unit tests (matlab/+prost/+test/*.m) restated against the CPU oracle.
They pin the leaf operators the reference cannot run here (gradient / diags / sparse kernels need
a GPU launch or cuSPARSE): same formulas, same sizes, same tolerances (norm(diff) <= 1e-3)."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
import prost_amd as prost
from reference_matrices import label_first_perm, spdiags_const, spmat_gradient2d, spmat_gradient3d

DTYPES = [np.float32, np.float64]


@pytest.mark.parametrize("dtype", DTYPES)
def test_linop_gradient2d(dtype):
    """test_linop_gradient2d.m:3-51 (nx=307, ny=229, L=8)"""
    nx, ny, L = 307, 229, 8
    rng = np.random.default_rng(0)
    linop = [prost.block.gradient2d(nx, ny, L, False)(0, 0, nx * ny * 2 * L, nx * ny * L)[0]]
    K = spmat_gradient2d(nx, ny, L)
    inp, inp2 = rng.random(nx * ny * L), rng.random(nx * ny * L * 2)
    x, _, _ = oracle.eval_linop(linop, inp, False, dtype)
    y, rowsum, colsum = oracle.eval_linop(linop, inp2, True, dtype)
    assert np.linalg.norm(x - K @ inp) <= 1e-3
    assert np.linalg.norm(y - K.T @ inp2) <= 1e-3
    assert not (rowsum < np.asarray(abs(K).sum(axis=1)).ravel()).any()     # one-sided check of the reference (:40-49)
    assert not (colsum < np.asarray(abs(K).sum(axis=0)).ravel()).any()
    assert np.all(rowsum == 2) and np.all(colsum == 4)                      # block_gradient2d.cu:154-163


@pytest.mark.parametrize("dtype", DTYPES)
def test_linop_gradient3d(dtype):
    """test_linop_gradient3d.m (nx=151, ny=291, L=7), Dirichlet in z"""
    nx, ny, L = 151, 291, 7
    rng = np.random.default_rng(1)
    linop = [prost.block.gradient3d(nx, ny, L, False)(0, 0, nx * ny * 3 * L, nx * ny * L)[0]]
    K = spmat_gradient3d(nx, ny, L)
    inp, inp2 = rng.random(nx * ny * L), rng.random(nx * ny * L * 3)
    x, _, _ = oracle.eval_linop(linop, inp, False, dtype)
    y, rowsum, colsum = oracle.eval_linop(linop, inp2, True, dtype)
    assert np.linalg.norm(x - K @ inp) <= 1e-3
    assert np.linalg.norm(y - K.T @ inp2) <= 1e-3
    assert np.all(rowsum == 2) and np.all(colsum == 6)


@pytest.mark.parametrize("shape", [(7, 5, 3), (1, 9, 2), (9, 1, 2), (33, 13, 4)])
@pytest.mark.parametrize("d3", [False, True])
def test_gradient_label_first_is_a_permutation(shape, d3):
    """label_first only reorders entries: K_lf = P_out K P^T (block_gradient2d.cu:46-56)"""
    nx, ny, L = shape
    rng = np.random.default_rng(2)
    n = nx * ny * L
    k = 3 if d3 else 2
    K = (spmat_gradient3d if d3 else spmat_gradient2d)(nx, ny, L)
    P = label_first_perm(nx, ny, L)
    Pout = sp.block_diag([P] * k)
    v = rng.standard_normal(n)
    w = rng.standard_normal(k * n)
    og = oracle.grad3d if d3 else oracle.grad2d
    assert np.allclose(og(v, nx, ny, L, True), Pout @ (K @ (P.T @ v)), atol=1e-12)
    assert np.allclose(og(w, nx, ny, L, True, adjoint=True), P @ (K.T @ (Pout.T @ w)), atol=1e-12)
    assert np.allclose(og(v, nx, ny, L, False), K @ v, atol=1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linop_diags(dtype):
    """test_linop_diags.m:3-78: 3 x 9 grid of 5912 x 1131 blocks, 29 random diagonals each"""
    Ndiags, nrows, ncols, By, Bx = 29, 5912, 1131, 3, 9
    rng = np.random.default_rng(3)
    rows_K, linop, row = [], [], 0
    for i in range(By):
        col, krow = 0, []
        for j in range(Bx):
            factors = rng.random(Ndiags)
            offsets = rng.permutation(nrows + ncols - 2)[:Ndiags] - nrows + 1
            krow.append(spdiags_const(nrows, ncols, factors, offsets))
            linop.append(prost.block.diags(nrows, ncols, factors, offsets)(row, col, nrows, ncols)[0])
            col += ncols
        row += nrows
        rows_K.append(sp.hstack(krow))
    K = sp.vstack(rows_K).tocsr()
    inp, inp2 = rng.standard_normal(ncols * Bx), rng.standard_normal(nrows * By)
    x, _, _ = oracle.eval_linop(linop, inp, False, dtype)
    y, rowsum, colsum = oracle.eval_linop(linop, inp2, True, dtype)
    tol = 1e-3 if dtype == np.float64 else 2e-2      # the reference test runs the double build
    assert np.linalg.norm(x - K @ inp) <= tol
    assert np.linalg.norm(y - K.T @ inp2) <= tol
    assert np.allclose(rowsum, np.asarray(abs(K).sum(axis=1)).ravel(), rtol=1e-5)
    assert np.allclose(colsum, np.asarray(abs(K).sum(axis=0)).ravel(), rtol=1e-5)


def test_diags_adjoint_grid_quirk_is_documented():
    """block_diags.cu:211 sizes the adjoint grid from nrows: with ncols > ceil(nrows/256)*256 the
    reference leaves the trailing columns untouched; the oracle can reproduce both behaviours."""
    nrows, ncols = 40, 700
    ofs, fac = oracle.diags_sort([0, 300, 650], [1.0, 2.0, 3.0], np.float64)
    y = np.ones(nrows)
    full = oracle.diags(y, nrows, ncols, ofs, fac, adjoint=True, ref_grid_quirk=False)
    quirk = oracle.diags(y, nrows, ncols, ofs, fac, adjoint=True, ref_grid_quirk=True)
    K = spdiags_const(nrows, ncols, [1.0, 2.0, 3.0], [0, 300, 650])
    assert np.allclose(full, K.T @ y)
    assert np.array_equal(quirk[:256], full[:256]) and np.all(quirk[256:] == 0) and np.any(full[256:] != 0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_linop_sparse_zero(dtype):
    """test_linop_sparse_zero.m: block layouts of sprand(.,.,0.01) and zero blocks vs K*x, K'*y"""
    rng = np.random.default_rng(4)
    for trial in range(4):
        nr = rng.integers(50, 300, size=3); nc = rng.integers(50, 300, size=2)
        blocks, linop, row = [], [], 0
        for i in range(3):
            col, brow = 0, []
            for j in range(2):
                if rng.random() < 0.3:
                    brow.append(sp.csr_matrix((nr[i], nc[j])))
                    linop.append(prost.block.zero()(row, col, int(nr[i]), int(nc[j]))[0])
                else:
                    Kb = sp.random(nr[i], nc[j], density=0.01 + 0.05 * rng.random(), random_state=int(rng.integers(1 << 30)))
                    brow.append(Kb)
                    linop.append(prost.block.sparse(Kb)(row, col, int(nr[i]), int(nc[j]))[0])
                col += nc[j]
            row += nr[i]
            blocks.append(sp.hstack(brow))
        K = sp.vstack(blocks).tocsr()
        inp, inp2 = rng.standard_normal(K.shape[1]), rng.standard_normal(K.shape[0])
        x, _, _ = oracle.eval_linop(linop, inp, False, dtype)
        y, rowsum, colsum = oracle.eval_linop(linop, inp2, True, dtype)
        assert np.linalg.norm(x - K @ inp) <= 1e-3
        assert np.linalg.norm(y - K.T @ inp2) <= 1e-3
        assert np.allclose(rowsum, np.asarray(abs(K).sum(axis=1)).ravel(), rtol=1e-5, atol=1e-7)
        assert np.allclose(colsum, np.asarray(abs(K).sum(axis=0)).ravel(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_sum_norm2(dtype):
    """test_prox_sum_norm2.m:3-29: ind_leq0 of ||.||-1 == projection onto the unit ball"""
    N, d = 6000, 7
    rng = np.random.default_rng(5)
    P = -2 + 4 * rng.random((N, d))
    Q = oracle.eval_prox(prost.function.sum_norm2(d, False, "ind_leq0", np.ones(N), 1, np.ones(N), 0, 0, 0, 0),
                         P.reshape(-1, order="F"), 1, np.ones(N * d), dtype).reshape((N, d), order="F")
    nrm = np.sqrt((P ** 2).sum(axis=1, keepdims=True))
    Q2 = np.where(nrm <= 1, P, P / nrm)
    assert np.abs(Q - Q2).max() < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_conjugate(dtype):
    """test_prox_conjugate.m:3-36: the biconjugate prox equals the prox"""
    rng = np.random.default_rng(6)
    for i in range(10):
        N = 5000
        a, b, c, d, e, y = (rng.random(N) for _ in range(6))
        tau, Tau = rng.random(), rng.random(N) + 1e-3
        f = prost.function.sum_1d("abs", a, b, c, d, e)
        x = oracle.eval_prox(f, y, tau, Tau, dtype)
        x2 = oracle.eval_prox(prost.function.conjugate(prost.function.conjugate(f)), y, tau, Tau, dtype)
        assert np.abs(x - x2).max() <= (1e-5 if dtype == np.float64 else 2e-3)


def test_moreau_identity_for_every_function():
    """x = prox_{tau f}(x) + tau prox_{f*/tau}(x/tau): the conjugate built by prost.function.conjugate
    composed with itself is the identity for every convex Function1D"""
    rng = np.random.default_rng(7)
    N = 400
    y = rng.uniform(-2, 2, N)
    for fn in ("zero", "abs", "square", "ind_leq0", "ind_geq0", "ind_eq0", "ind_box01", "max_pos0", "huber"):
        f = prost.function.sum_1d(fn, 1.5, 0.3, 2.0, 0.1, 0.2, 0.7, 0.0)
        x = oracle.eval_prox(f, y, 0.6, np.full(N, 0.8))
        x2 = oracle.eval_prox(prost.function.conjugate(prost.function.conjugate(f)), y, 0.6, np.full(N, 0.8))
        assert np.allclose(x, x2, atol=1e-10), fn


@pytest.mark.parametrize("d3", [False, True])
def test_adjointness_dot_test(d3):
    rng = np.random.default_rng(8)
    nx, ny, L = 37, 41, 3
    n = nx * ny * L
    k = 3 if d3 else 2
    og = oracle.grad3d if d3 else oracle.grad2d
    x, y = rng.standard_normal(n), rng.standard_normal(k * n)
    for lf in (False, True):
        assert np.isclose(np.dot(og(x, nx, ny, L, lf), y), np.dot(x, og(y, nx, ny, L, lf, adjoint=True)), rtol=1e-12)


def test_rof_primal_dual_gap_decreases():
    """example_rof_pdgap.m:1-17 on the oracle's PDHG iterates: the gap shrinks towards 0"""
    from prost_amd import synthetic
    nx = ny = 48
    lmb = 10.0
    prob, u, q, f = synthetic.rof_problem(nx, ny, 1, lmb)
    K = spmat_gradient2d(nx, ny, 1)
    gaps = []

    def cb(it, x, y):
        g = (K @ x).reshape(2, nx * ny)
        en_prim = 0.5 * lmb * ((x - f) ** 2).sum() + np.sqrt((g ** 2).sum(axis=0)).sum()
        div = K.T @ y
        en_dual = f @ div - (1 / (2 * lmb)) * (div ** 2).sum()
        gaps.append((en_prim - en_dual) / (nx * ny))
        return gaps[-1] < 1e-5
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * lmb)
    o = prost.options(max_iters=3000, num_cback_calls=30, verbose=False, interm_cb=cb, tol_rel_primal=0, tol_rel_dual=0,
                      tol_abs_primal=0, tol_abs_dual=0)
    r = oracle.solve(prob, b, o, np.float64)
    assert r["result"] == "Converged." and gaps[-1] < 1e-5 and gaps[-1] < gaps[0] * 1e-3
