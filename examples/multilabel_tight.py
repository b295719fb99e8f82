"""matlab/examples/example_multilabel_tight.m on the MI355X build, line for line: the tight relaxation of the Potts model with pairwise
Lagrange multipliers v -- linear operators instead of matrices (:78-87): gradient2d(nx, ny, L), sparse_kron_id(ones(1, L), ny nx) for
the simplex constraint, identity() between v and p, sparse_kron_id(pair_local', ny nx) between v and q -- PDHG, boyd, residual_iter 10.
Synthetic RGB image instead of images/junction_gray.png.  usage: python examples/multilabel_tight.py [nx ny]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import prost_amd as prost
from multilabel_fast import unary_potentials


def describe(nx, ny, lmb=1.0, L=3, tol=2e-6, max_iters=20000, num_cback_calls=25):
    f, im = unary_potentials(nx, ny)
    k = L * (L - 1) // 2                                                                      # :20 number of pairwise terms
    pair_local = sp.lil_matrix((2 * k, 2 * L))                                                # :29-39
    idx = 0
    for i in range(L):
        for j in range(i + 1, L):
            pair_local[idx, i] = 1
            pair_local[idx, j] = -1
            pair_local[idx + k, i + L] = 1
            pair_local[idx + k, j + L] = -1
            idx += 1
    u = prost.variable(nx * ny * L)                                                           # :43
    v = prost.variable(2 * nx * ny * k)                                                       # :44 Lagrange multipliers of the pairwise constraints
    q = prost.variable(2 * nx * ny * L)                                                       # :47
    p = prost.variable(2 * nx * ny * k)                                                       # :48
    s = prost.variable(nx * ny)                                                               # :49
    prob = prost.min_max_problem([u, v], [q, p, s])                                           # :51
    prob.add_function(u, prost.function.sum_1d("ind_geq0", 1, 0, 1, f, 0))                    # :54
    prob.add_function(p, prost.function.sum_norm2(2, False, "ind_leq0", 1 / lmb, 1, 1, 0, 0))  # :57-58
    prob.add_function(s, prost.function.sum_1d("zero", 1, 0, 1, 1, 0))                        # :61
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, L))                               # :78
    prob.add_dual_pair(u, s, prost.block.sparse_kron_id(sp.csr_matrix(np.ones((1, L))), ny * nx))   # :81
    prob.add_dual_pair(v, p, prost.block.identity())                                          # :84
    prob.add_dual_pair(v, q, prost.block.sparse_kron_id(pair_local.T.tocsr(), ny * nx))       # :87
    backend = prost.backend.pdhg(stepsize="boyd", residual_iter=10)                           # :93-94
    opts = prost.options(max_iters=max_iters, tol_rel_primal=tol, tol_abs_primal=tol, tol_rel_dual=tol, tol_abs_dual=tol,
                         num_cback_calls=num_cback_calls, verbose=False)                      # :98-106
    return prob, backend, opts, u, f, im


def main(nx=256, ny=256, max_iters=20000, verbose=True, backend_opts=None):
    prob, backend, opts, u, f, im = describe(nx, ny, max_iters=max_iters)
    if backend_opts:
        backend[1].update(backend_opts)
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                                 # :109
    elapsed = time.perf_counter() - t0
    lab = np.asarray(u.val).reshape(3, nx, ny)                                                # :112
    if verbose:
        print("%s after %d iterations, %.3f s on %s" % (result["result"], result["iters"], elapsed, result["path"]))
    return result, lab, f


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
