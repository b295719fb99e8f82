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


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_transform(dtype):
    """test_prox_transform.m:3-36: transform(sum_1d('abs'), a..e) == sum_1d('abs', a..e) under conjugation"""
    rng = np.random.default_rng(16)
    for i in range(10):
        N = 5000
        a, b, c, d, e, y = (rng.random(N) for _ in range(6))
        a = a + 1e-3
        tau, Tau = rng.random() + 1e-3, rng.random(N) + 1e-3
        x = oracle.eval_prox(prost.function.conjugate(prost.function.sum_1d("abs", a, b, c, d, e)), y, tau, Tau, dtype)
        x2 = oracle.eval_prox(prost.function.conjugate(prost.function.transform(prost.function.sum_1d("abs", 1, 0, 1, 0, 0), a, b, c, d, e)),
                              y, tau, Tau, dtype)
        assert np.abs(x - x2).max() <= (1e-5 if dtype == np.float64 else 5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_conj_trans(dtype):
    """test_prox_conj_trans.m:3-36: conjugate shifting, (f(. - b))^* = f^* + <b, .>, i.e.
    conjugate(sum_1d('abs', 1, b, 1, 0, 0)) == transform(conjugate(sum_1d('abs')), 1, 0, 1, b, 0)"""
    rng = np.random.default_rng(21)
    for i in range(10):
        N = 500
        b, y = rng.random(N), rng.random(N)
        tau, Tau = rng.random() + 1e-3, rng.random(N) + 1e-3
        x = oracle.eval_prox(prost.function.conjugate(prost.function.sum_1d("abs", 1, b, 1, 0, 0)), y, tau, Tau, dtype)
        x2 = oracle.eval_prox(prost.function.transform(prost.function.conjugate(prost.function.sum_1d("abs", 1, 0, 1, 0, 0)), 1, 0, 1, b, 0), y, tau, Tau, dtype)
        assert np.abs(x - x2).max() <= (1e-5 if dtype == np.float64 else 5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_transform_equals_coefficients(dtype):
    """transform.m:7-12: transform(sum_1d(fn), a..e) is the same function as sum_1d(fn, a..e), scalar and vector coefficients"""
    rng = np.random.default_rng(17)
    N = 3000
    y = rng.standard_normal(N); Tau = rng.random(N) + 0.1
    for coeffs in ((2.0, 0.5, 3.0, 0.1, 0.2), tuple(rng.random(N) + 0.2 for _ in range(5))):
        for fn in ("abs", "square", "huber"):
            alpha = 0.3
            x = oracle.eval_prox(prost.function.sum_1d(fn, *coeffs, alpha), y, 0.7, Tau, dtype)
            x2 = oracle.eval_prox(prost.function.transform(prost.function.sum_1d(fn, 1, 0, 1, 0, 0, alpha), *coeffs), y, 0.7, Tau, dtype)
            assert np.abs(x - x2).max() <= (1e-9 if dtype == np.float64 else 2e-4), (fn, np.abs(x - x2).max())


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_permute(dtype):
    """test_prox_permute.m:3-27"""
    rng = np.random.default_rng(18)
    n = 34
    y = 10 * rng.standard_normal(n)
    perm = rng.permutation(n)
    inv_perm = np.empty(n, dtype=int); inv_perm[perm] = np.arange(n)
    f = prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1)
    x1 = oracle.eval_prox(prost.function.permute(f, perm), y, 0.1, np.ones(n), dtype)
    x2 = oracle.eval_prox(f, y[perm], 0.1, np.ones(n), dtype)
    assert np.abs(x1 - x2[inv_perm]).max() <= 1e-5
    with pytest.raises(oracle.OracleError, match="Permutation vector has wrong size"):
        oracle.eval_prox(prost.function.permute(f, perm[:-2]), y, 0.1, np.ones(n), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_sum_ind_sum(dtype):
    """test_prox_sum_ind_sum.m:3-21 (elem_operation:ind_sum) and the index-family form sum_ind_sum2"""
    rng = np.random.default_rng(19)
    N, d = 21, 3
    y = rng.standard_normal(N)
    x = oracle.eval_prox(prost.function.sum_ind_sum(d, False), y, 1, np.ones(N), dtype).reshape((N // d, d), order="F")
    assert np.abs(x.sum(axis=1) - 1).max() <= 1e-5
    xi = oracle.eval_prox(prost.function.sum_ind_sum(d, True), y, 1, np.ones(N), dtype).reshape((N // d, d))
    assert np.abs(xi.sum(axis=1) - 1).max() <= 1e-5
    # index families: rows of a 5 x 4 array sum to 2, its first 3 columns (as a second family) to 0.5;
    # with uniform steps the result is the Euclidean projection, untouched entries stay
    inds = np.arange(20).reshape(5, 4)
    Tau = rng.random(24) + 0.5
    y = rng.standard_normal(24)
    x = oracle.eval_prox(prost.function.sum_ind_sum2(4, inds.ravel(), 2.0), y, 0.9, Tau, dtype)
    assert np.abs(x[:20].reshape(5, 4).sum(axis=1) - 2).max() <= 1e-5 and np.array_equal(x[20:], y[20:].astype(dtype).astype(np.float64))
    x_u = oracle.eval_prox(prost.function.sum_ind_sum2(4, inds.ravel(), 2.0), y, 0.9, np.ones(24), dtype)
    assert np.abs(x_u[:20].reshape(5, 4) - (y[:20].reshape(5, 4) - (y[:20].reshape(5, 4).sum(axis=1, keepdims=True) - 2) / 4)).max() <= 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_ind_halfspace_and_soc(dtype):
    """closed forms of prox_ind_halfspace.cu:31-50 and prox_ind_soc.cu:30-77 (no reference test exists for them)"""
    rng = np.random.default_rng(20)
    count, dim = 400, 3
    V = rng.standard_normal((count, dim)); A = rng.standard_normal((count, dim)); b = rng.standard_normal(count)
    x = oracle.eval_prox(prost.function.sum_ind_halfspace(dim, False, A.reshape(-1, order="F"), b), V.reshape(-1, order="F"), 1, np.ones(count * dim),
                         dtype).reshape((count, dim), order="F")
    exc = np.maximum(0, (A * V).sum(axis=1) - b) / (A * A).sum(axis=1)
    assert np.abs(x - (V - exc[:, None] * A)).max() <= 1e-5
    assert ((A * x).sum(axis=1) <= b + 1e-4).all()
    a1 = rng.standard_normal(dim)
    x1 = oracle.eval_prox(prost.function.sum_ind_halfspace(dim, False, a1, 0.3), V.reshape(-1, order="F"), 1, np.ones(count * dim),
                          dtype).reshape((count, dim), order="F")
    assert ((x1 @ a1) <= 0.3 + 1e-4).all()
    # second-order cone: (x, y) with ||x|| <= y; three regimes
    W = rng.standard_normal((count, dim)) * 2
    s = oracle.eval_prox(prost.function.sum_ind_soc(dim, False, 1), W.reshape(-1, order="F"), 1, np.ones(count * dim), dtype).reshape((count, dim), order="F")
    nx, y0 = np.sqrt((W[:, :-1] ** 2).sum(axis=1)), W[:, -1]
    fac = np.where(nx <= y0, 1.0, np.where(nx <= -y0, 0.0, (y0 + nx) / (2 * nx)))
    ref = np.concatenate([fac[:, None] * W[:, :-1], np.where(nx <= y0, y0, fac * nx)[:, None]], axis=1)
    assert np.abs(s - ref).max() <= 1e-5
    assert (np.sqrt((s[:, :-1] ** 2).sum(axis=1)) <= s[:, -1] + 1e-4).all()
    with pytest.raises(oracle.OracleError, match="Only alpha = 1"):
        oracle.eval_prox(prost.function.sum_ind_soc(dim, False, 2), W.reshape(-1, order="F"), 1, np.ones(count * dim), dtype)


def _projsplx(y):
    u = np.sort(y)[::-1]
    css = np.cumsum(u)
    rho = np.nonzero(u * np.arange(1, y.size + 1) > (css - 1))[0][-1]
    return np.maximum(y - (css[rho] - 1) / (rho + 1), 0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_prox_sum_ind_simplex(dtype):
    """test_prox_sum_ind_simplex.m:3-36 against the sort-based projection (projsplx)"""
    rng = np.random.default_rng(22)
    N, d = 300, 17 * 17
    P = -2 + 4 * rng.random((N, d))
    Q = oracle.eval_prox(prost.function.sum_ind_simplex(d, False), P.reshape(-1, order="F"), 1, np.ones(N * d), dtype).reshape((N, d), order="F")
    Q2 = np.array([_projsplx(P[i]) for i in range(N)])
    assert np.abs(Q - Q2).max() <= 1e-5
    Qi = oracle.eval_prox(prost.function.sum_ind_simplex(7, True), P[:, :7].reshape(-1), 1, np.ones(N * 7), dtype).reshape((N, 7))
    assert np.abs(Qi - np.array([_projsplx(P[i, :7]) for i in range(N)])).max() <= 1e-5
    one = oracle.eval_prox(prost.function.sum_ind_simplex(1, False), P[:, 0], 1, np.ones(N), dtype)
    assert np.abs(one - 1).max() <= 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["sparse_kron_id", "id_kron_sparse"])
def test_linop_kronecker_blocks(dtype, name):
    """test_linop_sparse_kron_id.m / test_linop_id_kron_sparse.m:3-95: four copies of kron(K, I) resp.
    kron(I, K) in a 2 x 2 arrangement against the explicit Kronecker product, incl. row / column sums"""
    rng = np.random.default_rng(23)
    diaglength, nrows, ncols = 64 * 12, 81, 64
    K_mat = sp.random(nrows, ncols, 0.01, random_state=3, format="csc")
    bf = getattr(prost.block, name)(K_mat, diaglength)
    m, n = nrows * diaglength, ncols * diaglength
    linop = [bf(0, 0, m, n)[0], bf(m, 0, m, n)[0], bf(m, n, m, n)[0], bf(0, n, m, n)[0]]
    full = sp.kron(K_mat, sp.eye(diaglength)) if name == "sparse_kron_id" else sp.kron(sp.eye(diaglength), K_mat)
    K = sp.bmat([[full, full], [full, full]]).tocsr()
    inp, inp_t = rng.standard_normal(2 * n), rng.standard_normal(2 * m)
    x, rowsum, colsum = oracle.eval_linop(linop, inp, False, dtype)[:3]
    x_t = oracle.eval_linop(linop, inp_t, True, dtype)[0]
    assert np.abs(x - K @ inp).max() <= 1e-4 and np.abs(x_t - K.T @ inp_t).max() <= 1e-4
    assert np.abs(rowsum - np.asarray(abs(K).sum(axis=1)).ravel()).max() <= 1e-4
    assert np.abs(colsum - np.asarray(abs(K).sum(axis=0)).ravel()).max() <= 1e-4


# ---------------------------------------------------------------------------------------------
# ADMM / CGLS (backend_admm.cu:355-665, cgls.hpp:222-371): the reference's own code needs cuBLAS + cuSPARSE and cannot be
# compiled here, and the reference holds no test or golden vector for it -- PARITY UNPINNED.  What can
# be checked is that the restatement solves the problems it claims to solve.
# ---------------------------------------------------------------------------------------------
def c4_shape_problem(nx, ny, seed=0):
    """SURVEY 8(d) C4 shape (TV-L1 flow-like): v = W u, W = [diag(Ix) diag(Iy)] (block.sparse); g = gradient2d(nx, ny, 2) u;
    f(v) = 5 |v - b|_1, f(g) = sum of 4-norms"""
    from prost_amd import synthetic
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
    G = spmat_gradient2d(nx, ny, 2)

    def energy(x):
        return 5.0 * np.abs(W @ x - bvec).sum() + np.sqrt(((G @ x).reshape(4, -1) ** 2).sum(0)).sum()
    return prob, energy


def test_admm_and_pdhg_agree_at_convergence_on_the_c4_shape():
    """two different algorithms, one minimiser: the oracle's ADMM (graph projection by CGLS on the preconditioned operator)
    and its PDHG (pinned bit-exactly by the real reference build) reach the same energy on the C4 shape -- the survey's
    own probe of the reference saw sum(x) = 465.796 (ADMM) vs 465.885 (PDHG-boyd) on a 32 x 32 TV-L1 problem"""
    prob, energy = c4_shape_problem(32, 32)
    prob.finalize()
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    out = {}
    for name, b, its in (("admm", prost.backend.admm(rho0=1), 1500), ("pdhg", prost.backend.pdhg(stepsize="boyd", residual_iter=10), 4000)):
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float64)
        s.initialize()
        s.iterate(50)
        e_early = energy(s.state()["x"])
        s.iterate(its - 50)
        x = s.state()["x"]
        out[name] = (energy(x), x.sum(), e_early)
        assert out[name][0] < e_early                               # the energy falls
    (ea, sa, _), (ep, sp_, _) = out["admm"], out["pdhg"]
    assert abs(ea - ep) / ep < 1e-3, (ea, ep)
    assert abs(sa - sp_) / abs(sp_) < 1e-4, (sa, sp_)
