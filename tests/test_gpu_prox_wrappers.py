"""GPU parity of the wrapper / projection proxes: ProxTransform, ProxPermute,
ProxIndHalfspace, ProxIndSOC, ProxIndSum and elem_operation:ind_sum -- product path
(prost.eval_prox -> prost_command -> host C++ -> kernels_prox_wrap.hip) against the CPU oracle.
Bar: bit-exact (same expressions, no FMA contraction); lq-based inner functions excluded."""
import numpy as np
import pytest

import oracle
import prost_amd as prost
from prost_amd import synthetic

pytestmark = pytest.mark.gpu
PRECISIONS = [("single", np.float32), ("double", np.float64)]
F = prost.function


@pytest.fixture(autouse=True)
def _gpu(hip):
    prost.set_gpu(0)
    yield
    prost.set_precision("double")


def both(fn, arg, tau, Tau, dtype):
    got, _ = prost.eval_prox(fn, arg, tau, Tau)
    want = oracle.eval_prox(fn, arg, tau, Tau, dtype)
    return np.asarray(got, dtype=np.float64), want


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("n", [1, 7, 1024, 4099])
def test_transform(prec, dtype, n):
    prost.set_precision(prec)
    rng = np.random.default_rng(n)
    y = rng.standard_normal(n); Tau = rng.random(n) + 0.1
    vec = [rng.random(n) + 0.2 for _ in range(5)]
    for coeffs in ((2.0, 0.5, 3.0, 0.1, 0.2), tuple(vec), (vec[0], 0.5, vec[2], 0.0, vec[4]), (1.0, vec[1], 1.0, 0.0, 0.0)):
        for inner in (F.sum_1d("abs", 1, 0, 1, 0, 0), F.sum_1d("square", 1.5, 0.2, 2.0, 0, 0), F.conjugate(F.sum_1d("huber", 1, 0, 1, 0, 0, 0.3)), F.zero()):
            for fn in (F.transform(inner, *coeffs), F.conjugate(F.transform(inner, *coeffs))):
                got, want = both(fn, y, 0.7, Tau, dtype)
                assert np.array_equal(got, want), (n, float(np.abs(got - want).max()))
    if n % 2 == 0:
        fn = F.transform(F.sum_norm2(2, False, "abs", 1, 0, 1, 0, 0), vec[0], vec[1], 2.0, 0.1, 0.0)
        got, want = both(fn, y, 0.7, np.full(n, 0.6), dtype)
        assert np.array_equal(got, want)
    with pytest.raises(prost.ProstError, match="isn't allowed to contain zero element"):
        prost.eval_prox(F.transform(F.zero(), 0.0), y, 0.7, Tau)


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
def test_permute(prec, dtype):
    prost.set_precision(prec)
    rng = np.random.default_rng(2)
    for n in (34, 2048, 5000):
        y = 10 * rng.standard_normal(n); Tau = rng.random(n) + 0.1
        perm = rng.permutation(n)
        for inner in (F.sum_norm2(2, False, "ind_leq0", 1, 1, 1), F.sum_1d("abs", 1, rng.random(n), 2.0), F.conjugate(F.sum_norm2(2, True, "abs"))):
            got, want = both(F.permute(inner, perm), y, 0.1, Tau, dtype)
            assert np.array_equal(got, want), n
        # test_prox_permute.m:3-27 on the product path
        inv = np.empty(n, dtype=int); inv[perm] = np.arange(n)
        f = F.sum_norm2(2, False, "ind_leq0", 1, 1, 1)
        x1, _ = prost.eval_prox(F.permute(f, perm), y, 0.1, np.ones(n))
        x2, _ = prost.eval_prox(f, y[perm], 0.1, np.ones(n))
        assert np.abs(x1 - x2[inv]).max() <= 1e-5
    with pytest.raises(prost.ProstError, match="Permutation vector has wrong size"):
        prost.eval_prox(F.permute(F.sum_1d("abs"), np.arange(5)), np.zeros(8), 0.1, np.ones(8))


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("count,dim", [(1, 1), (5, 3), (1000, 2), (4099, 4)])
def test_halfspace_soc_ind_sum(prec, dtype, count, dim):
    prost.set_precision(prec)
    rng = np.random.default_rng(count + dim)
    n = count * dim
    V = rng.standard_normal(n); Tau = rng.random(n) + 0.2
    A = rng.standard_normal(n); b = rng.standard_normal(count)
    for fn in (F.sum_ind_halfspace(dim, False, A, b), F.sum_ind_halfspace(dim, True, A, 0.25), F.sum_ind_halfspace(dim, False, A[:dim], b),
               F.sum_ind_halfspace(dim, False, A[:dim], -0.5), F.conjugate(F.sum_ind_halfspace(dim, False, A, b))):
        got, want = both(fn, V, 0.8, Tau, dtype)
        assert np.array_equal(got, want)
    for fn in (F.sum_ind_soc(dim, False, 1), F.conjugate(F.sum_ind_soc(dim, True, 1))):
        got, want = both(fn, 2 * V, 0.8, Tau, dtype)
        assert np.array_equal(got, want)
    for il in (False, True):
        got, want = both(F.sum_ind_sum(dim, il), V, 1.0, Tau, dtype)
        assert np.array_equal(got, want)
    # index families over a prox range with 3 untouched trailing entries
    inds = rng.permutation(n).reshape(count, dim)
    y = rng.standard_normal(n + 3); T3 = rng.random(n + 3) + 0.2
    for fn in (F.sum_ind_sum2(dim, inds.ravel(), 2.0), F.conjugate(F.sum_ind_sum2(dim, inds.ravel(), -1.0))):
        got, want = both(fn, y, 0.9, T3, dtype)
        assert np.array_equal(got, want)
    if count % 2 == 0:      # two families: rows and (half as many) double rows
        fn = F.sum_ind_sum2(dim, inds.ravel(), 1.0, 2 * dim, inds.ravel(), 0.5)
        got, want = both(fn, y, 0.9, T3, dtype)
        assert np.array_equal(got, want)
    with pytest.raises(prost.ProstError, match="Only alpha = 1"):
        prost.eval_prox(F.sum_ind_soc(dim, False, 2), V, 1, Tau)
    with pytest.raises(prost.ProstError, match="Coefficient b has to have dimension"):
        prost.eval_prox(F.sum_ind_halfspace(dim, False, A, np.zeros(count + 2)), V, 1, Tau)


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
def test_rof_through_transform_matches_oracle_and_coefficient_form(prec, dtype):
    """example_rof_primaldual.m:22-23: the data term as transform(sum_1d('square'), 1, f, lmb).  The
    solver takes the generic path (prox_g is not an elem operation); iterates equal the oracle's bit
    for bit and agree with the coefficient form to rounding."""
    prost.set_precision(prec)
    nx, ny = 24, 20
    f = np.asarray(synthetic.rof_image(nx, ny, 1, 7)).ravel()
    iters = {}
    for form in ("transform", "coeff"):
        u = prost.variable(nx * ny); q = prost.variable(2 * nx * ny)
        prob = prost.min_max_problem([u], [q])
        g = F.transform(F.sum_1d("square"), 1, f, 10.0) if form == "transform" else F.sum_1d("square", 1, f, 10.0)
        prob.add_function(u, g)
        prob.add_function(q, F.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
        prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, 1))
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=5, alg2_gamma=0.5)
        o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
        s = prost.Solver(prob, b, o); s.iterate(30); st = s.state(); s.destroy()
        so = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, dtype); so.initialize(); so.iterate(30); ost = so.state()
        for v in "xy":
            assert np.array_equal(st[v], ost[v]), (form, v)
        iters[form] = st
    assert np.abs(iters["transform"]["x"] - iters["coeff"]["x"]).max() <= (1e-10 if dtype == np.float64 else 1e-4)


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
def test_ind_simplex(prec, dtype):
    prost.set_precision(prec)
    rng = np.random.default_rng(31)
    for (N, d, il) in ((300, 289, False), (1000, 5, False), (1000, 5, True), (64, 1, False), (3, 1000, False)):
        P = -2 + 4 * rng.random(N * d)
        got, want = both(F.sum_ind_simplex(d, il), P, 1.0, np.ones(N * d), dtype)
        assert np.array_equal(got, want), (N, d, il)
        g = got.reshape((N, d), order="C" if il else "F")
        assert np.abs(g.sum(axis=1) - 1).max() <= 1e-4 and g.min() >= 0
    got, want = both(F.conjugate(F.sum_ind_simplex(4, False)), rng.standard_normal(400), 0.7, np.full(400, 0.9), dtype)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
@pytest.mark.parametrize("name", ["sparse_kron_id", "id_kron_sparse"])
def test_kronecker_blocks(prec, dtype, name):
    """product eval_linop (value, transposed value, row / column sums) == oracle bit for bit, and the
    reference's own check against the explicit Kronecker product (test_linop_sparse_kron_id.m:3-95)"""
    import scipy.sparse as sp
    prost.set_precision(prec)
    rng = np.random.default_rng(32)
    # (the last five: 16 bytes per lane with a ragged last tile; K too large for the LDS copy of its arrays (3000 non-zeros), then for the
    #  tiled kernel altogether (1100 rows); a single row; an identity shorter than a vector)
    shapes = ((64 * 12, 81, 64, 0.01), (5, 3, 4, 0.6), (1, 9, 7, 0.3), (700, 2, 2, 1.0), (4100, 5, 3, 0.5), (40, 300, 200, 0.05), (8, 1100, 30, 0.01),
              (1028, 1, 6, 0.7), (2, 12, 16, 0.2))
    for (diaglength, nrows, ncols, dens) in shapes:
        K_mat = sp.random(nrows, ncols, dens, random_state=5, format="csc")
        bf = getattr(prost.block, name)(K_mat, diaglength)
        m, n = nrows * diaglength, ncols * diaglength
        linop = [bf(0, 0, m, n)[0], bf(m, 0, m, n)[0], bf(m, n, m, n)[0], bf(0, n, m, n)[0]]
        inp, inp_t = rng.standard_normal(2 * n), rng.standard_normal(2 * m)
        x, rowsum, colsum, _ = prost.eval_linop(linop, inp, False)
        x_t = prost.eval_linop(linop, inp_t, True)[0]
        ox, orow, ocol = oracle.eval_linop(linop, inp, False, dtype)[:3]
        ox_t = oracle.eval_linop(linop, inp_t, True, dtype)[0]
        assert np.array_equal(x, ox) and np.array_equal(x_t, ox_t)
        assert np.allclose(rowsum, orow, rtol=1e-6) and np.allclose(colsum, ocol, rtol=1e-6)
        full = sp.kron(K_mat, sp.eye(diaglength)) if name == "sparse_kron_id" else sp.kron(sp.eye(diaglength), K_mat)
        K = sp.bmat([[full, full], [full, full]]).tocsr()
        assert np.abs(x - K @ inp).max() <= 1e-4 and np.abs(x_t - K.T @ inp_t).max() <= 1e-4


@pytest.mark.parametrize("prec,dtype", PRECISIONS)
def test_fused_moreau_equals_three_kernel_sequence(prec, dtype):
    """conjugate(elem operation) as ONE kernel (prost_hip_prox_elem_moreau) == MoreauPrescale + elem operation +
    MoreauPostscale (prox_moreau.cu:98-134) bit for bit, and both == the oracle; every function, both
    operations, both layouts, per-element coefficients, and the doubly conjugated (inverted step) form"""
    prost.set_precision(prec)
    rng = np.random.default_rng(77)
    fns = [f for f in F.FUNCTIONS_1D if f not in ("lq", "lq_plus_eps")]
    for n, dim in ((6, 1), (1027, 1), (4096, 2), (3000, 3), (35, 7)):
        arg = rng.standard_normal(n); Tau = rng.random(n) + 0.2
        vec = [rng.random(n // dim) + 0.3 for _ in range(3)]
        for fn in fns:
            if dim == 1:
                cases = [F.sum_1d(fn, 1.5, 0.2, 2.0, 0.1, 0.3, 0.4, 0.6), F.sum_1d(fn, rng.random(n) + 0.3, rng.random(n), 2.0, 0.0, 0.0, 0.4, 0.6)]
            else:
                cases = [F.sum_norm2(dim, il, fn, vec[0], 0.2, vec[2], 0.0, 0.1, 0.4, 0.6) for il in (False, True)]
            for f in cases:
                for g in (F.conjugate(f), F.conjugate(F.conjugate(f))):
                    prost.set_quirks(fuse_moreau=1)
                    fused, _ = prost.eval_prox(g, arg, 0.8, Tau)
                    prost.set_quirks(fuse_moreau=0)
                    plain, _ = prost.eval_prox(g, arg, 0.8, Tau)
                    prost.set_quirks(fuse_moreau=1)
                    assert np.array_equal(fused, plain, equal_nan=True), (fn, n, dim)
                    want = oracle.eval_prox(g, arg, 0.8, Tau, dtype)
                    assert np.array_equal(np.asarray(fused, dtype=np.float64), want, equal_nan=True), (fn, n, dim)
