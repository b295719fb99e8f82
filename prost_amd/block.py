"""Linear-operator block builders -- Python mirror of matlab/+prost/+block/*.m.

Each builder returns ``func(row, col, nrows, ncols) -> [[name, row, col, data], [nrows, ncols]]``
exactly like the MATLAB closures (gradient2d.m:12-14, gradient3d.m:12-14, sparse.m:6-9,
diags.m:17-20, identity.m:11-12, zero.m:3-4).  Cells are Python lists.
"""
import numpy as np


def gradient2d(nx, ny, L, label_first=False):
    sz = [nx * ny * L * 2, nx * ny * L]
    data = [nx, ny, L, bool(label_first)]
    return lambda row, col, nrows, ncols: [["gradient2d", row, col, data], sz]


def gradient3d(nx, ny, L, label_first=False):
    sz = [nx * ny * L * 3, nx * ny * L]
    data = [nx, ny, L, bool(label_first)]
    return lambda row, col, nrows, ncols: [["gradient3d", row, col, data], sz]


def sparse(K):
    import scipy.sparse as sp
    K = sp.csc_matrix(K, dtype=np.float64, copy=True)      # (copy: the two calls below work in place and K may BE the caller's matrix)
    K.eliminate_zeros()                 # a MATLAB sparse matrix holds no explicit zeros (scipy constructions such as diags / kron do)
    K.sort_indices()
    sz = [K.shape[0], K.shape[1]]
    return lambda row, col, nrows, ncols: [["sparse", row, col, [K]], sz]


def _kron(name, K, diaglength):
    import scipy.sparse as sp
    K = sp.csc_matrix(K, dtype=np.float64, copy=True)
    K.eliminate_zeros()
    K.sort_indices()
    d = int(diaglength)
    sz = [K.shape[0] * d, K.shape[1] * d]
    return lambda row, col, nrows, ncols: [[name, row, col, [K, d]], sz]


def sparse_kron_id(K, diaglength):
    """sparse_kron_id.m:1-14: kron(K, speye(diaglength)) without forming it"""
    return _kron("sparse_kron_id", K, diaglength)


def id_kron_sparse(K, diaglength):
    """id_kron_sparse.m:1-14: kron(speye(diaglength), K) without forming it"""
    return _kron("id_kron_sparse", K, diaglength)


def diags(nrows, ncols, factors, offsets):
    sz = [nrows, ncols]
    data = [nrows, ncols, np.atleast_1d(np.asarray(factors, dtype=np.float64)),
            np.atleast_1d(np.asarray(offsets, dtype=np.float64))]
    return lambda row, col, nrows_, ncols_: [["diags", row, col, data], sz]


def identity(scal=1):
    return lambda row, col, nrows, ncols: [
        ["diags", row, col, [nrows, ncols, np.array([float(scal)]), np.array([0.0])]], [nrows, ncols]]


def zero():
    return lambda row, col, nrows, ncols: [["zero", row, col, [nrows, ncols]], [nrows, ncols]]
