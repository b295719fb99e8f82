"""scipy.sparse restatements of the MATLAB helper formulas the reference's own tests compare
against (matlab/+prost/+test/private/spmat_gradient2d.m:7-13, spmat_gradient3d.m:8-20)."""
import numpy as np
import scipy.sparse as sp


def spmat_gradient2d(nx, ny, L):
    dy = sp.diags([np.r_[-np.ones(ny - 1), 0.0], np.ones(ny - 1)], [0, 1], shape=(ny, ny))
    dy = sp.kron(sp.eye(nx), dy)
    dx = sp.diags([np.r_[-np.ones(ny * (nx - 1)), np.zeros(ny)], np.ones(nx * ny - ny)], [0, ny], shape=(nx * ny, nx * ny))
    return sp.vstack([sp.kron(sp.eye(L), dx), sp.kron(sp.eye(L), dy)]).tocsr()


def spmat_gradient3d(nx, ny, L):
    dy = sp.diags([np.r_[-np.ones(ny - 1), 0.0], np.ones(ny - 1)], [0, 1], shape=(ny, ny))
    dy = sp.kron(sp.eye(nx), dy)
    dx = sp.diags([np.r_[-np.ones(ny * (nx - 1)), np.zeros(ny)], np.ones(nx * ny - ny)], [0, ny], shape=(nx * ny, nx * ny))
    n = nx * ny * L
    dz = sp.diags([-np.ones(n), np.ones(n - nx * ny)], [0, nx * ny], shape=(n, n))    # no zeroed last slab: Dirichlet at z = L
    return sp.vstack([sp.kron(sp.eye(L), dx), sp.kron(sp.eye(L), dy), dz]).tocsr()


def label_first_perm(nx, ny, L):
    """P with (P v)[l + y*L + x*ny*L] = v[y + x*ny + l*nx*ny]"""
    idx_lf = np.arange(nx * ny * L)
    l = idx_lf % L
    y = (idx_lf // L) % ny
    x = idx_lf // (ny * L)
    src = y + x * ny + l * nx * ny
    return sp.csr_matrix((np.ones(nx * ny * L), (idx_lf, src)), shape=(nx * ny * L, nx * ny * L))


def spdiags_const(nrows, ncols, factors, offsets):
    """spdiags(ones(nrows,1) * factors', offsets, nrows, ncols) as used by test_linop_diags.m:26-27"""
    K = sp.lil_matrix((nrows, ncols))
    for f, o in zip(factors, offsets):
        r = np.arange(max(0, -int(o)), min(nrows, ncols - int(o)))
        if r.size:
            K[r, r + int(o)] = K[r, r + int(o)].toarray() + f
    return K.tocsr()
