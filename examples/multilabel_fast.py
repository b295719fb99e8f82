"""matlab/examples/example_multilabel_fast.m on the MI355X build, line for line: a 3-label segmentation relaxed to the simplex --
the gradient handed over as a SPARSE MATRIX (spmat_gradient2d(nx, ny, L), :21, :45), the simplex constraint as a second dual
variable s coupled through sum_op = kron(ones(1, L), speye(ny nx)) (:22, :48), vectorial TV over all labels (Lellmann et al., :39-40),
unary potentials f = squared distance to the label means (:9-14), PDHG with Boyd's residual balancing, residual_iter 10 (:52-53).
Synthetic RGB image instead of images/junction_gray.png (stored as RGB, 256 x 256).  usage: python examples/multilabel_fast.py [nx ny]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scipy.sparse as sp

import prost_amd as prost
from prost_amd import synthetic


def spmat_gradient2d(nx, ny, L):
    """matlab/+prost/+test/private/spmat_gradient2d.m:7-14 (forward differences, zero rows at the far borders, [all dx ; all dy])"""
    dy = sp.kron(sp.eye(nx), sp.diags([np.r_[-np.ones(ny - 1), 0.0], np.ones(ny - 1)], [0, 1], shape=(ny, ny)))
    dx = sp.diags([np.r_[-np.ones(ny * (nx - 1)), np.zeros(ny)], np.ones(nx * ny - ny)], [0, ny], shape=(nx * ny, nx * ny))
    return sp.vstack([sp.kron(sp.eye(L), dx), sp.kron(sp.eye(L), dy)]).tocsc()


def unary_potentials(nx, ny, seed=6):
    """:2-14: im (ny, nx, 3) in [0, 1]; f(:, :, i) = sum_c (im_c - means(c, i))^2 with means = eye(3); f = f(:) (column-major, label slowest)"""
    im = synthetic.rof_image(nx, ny, 3, seed).reshape(3, nx, ny)          # channel planes, column-major pixels (y fastest)
    means = np.eye(3)
    f = np.stack([((im - means[:, i].reshape(3, 1, 1)) ** 2).sum(axis=0) for i in range(3)])      # (label, x, y)
    return f.reshape(-1), im


def describe(nx, ny, lmb=1.0, L=3, tol=1e-5, max_iters=5000, num_cback_calls=250):
    f, im = unary_potentials(nx, ny)
    grad = spmat_gradient2d(nx, ny, L)                                                        # :21
    sum_op = sp.kron(np.ones((1, L)), sp.eye(ny * nx))                                        # :22
    u = prost.variable(nx * ny * L)                                                           # :24
    q = prost.variable(2 * nx * ny * L)                                                       # :25
    s = prost.variable(nx * ny)                                                               # :26
    prob = prost.min_max_problem([u], [q, s])                                                 # :28
    prob.add_function(u, prost.function.sum_1d("ind_geq0", 1, 0, 1, f, 0))                    # :31  I(u >= 0) + <u, f>
    prob.add_function(q, prost.function.sum_norm2(2 * L, False, "ind_leq0", 1 / lmb, 1, 1, 0, 0))   # :39-40
    prob.add_function(s, prost.function.sum_1d("zero", 1, 0, 1, 1, 0))                        # :43  <s, -1>
    prob.add_dual_pair(u, q, prost.block.sparse(grad))                                        # :46
    prob.add_dual_pair(u, s, prost.block.sparse(sum_op))                                      # :49
    backend = prost.backend.pdhg(stepsize="boyd", residual_iter=10)                           # :53-54
    opts = prost.options(max_iters=max_iters, tol_rel_primal=tol, tol_abs_primal=tol, tol_rel_dual=tol, tol_abs_dual=tol,
                         num_cback_calls=num_cback_calls, verbose=False)                      # :56-64
    return prob, backend, opts, u, f, im


def main(nx=256, ny=256, max_iters=5000, verbose=True, backend_opts=None):
    prob, backend, opts, u, f, im = describe(nx, ny, max_iters=max_iters)
    if backend_opts:
        backend[1].update(backend_opts)
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                                 # :68
    elapsed = time.perf_counter() - t0
    lab = np.asarray(u.val).reshape(3, nx, ny)                                                # :72
    if verbose:
        print("%s after %d iterations, %.3f s on %s; sum over labels in [%.4f, %.4f]" % (result["result"], result["iters"], elapsed, result["path"],
                                                                                     lab.sum(axis=0).min(), lab.sum(axis=0).max()))
    return result, lab, f


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
