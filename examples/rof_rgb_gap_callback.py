"""matlab/examples/example_rof_primaldual.m, line for line, on the MI355X build (reference lines cited).

The MATLAB script reads images/lion.png (700 x 464 RGB); no image file travels with this repo, so a synthetic RGB
image of the same size stands in (prost_amd.synthetic.rof_image).  Everything from `u = prost.variable(...)` on is the
reference's problem description unchanged; the primal-dual gap callback is example_rof_pdgap.m with the sparse
gradient matrix of +test/private/spmat_gradient2d.m.
usage: python examples/rof_rgb_gap_callback.py [nx ny nc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic


def spmat_gradient2d(nx, ny, nc):
    """matlab/examples/spmat_gradient2d.m: forward differences, Neumann boundary, [all d/dx | all d/dy]"""
    import scipy.sparse as sp
    dy = sp.diags([-np.ones(ny), np.ones(ny - 1)], [0, 1], shape=(ny, ny)).tolil(); dy[ny - 1, :] = 0
    dx = sp.diags([-np.ones(nx), np.ones(nx - 1)], [0, 1], shape=(nx, nx)).tolil(); dx[nx - 1, :] = 0
    gx = sp.kron(dx.tocsr(), sp.identity(ny)); gy = sp.kron(sp.identity(nx), dy.tocsr())
    return sp.vstack([sp.kron(sp.identity(nc), gx), sp.kron(sp.identity(nc), gy)]).tocsr()


def describe(nx=700, ny=464, nc=3, max_iters=10000, num_cback_calls=250):
    """the problem description, backend and options of example_rof_primaldual.m:15-46 as written, without the gap callback (which can
    stop the run): -> (prob, backend, opts)"""
    f = synthetic.rof_image(nx, ny, nc, seed=1).astype(np.float64)
    lmb = 10
    u = prost.variable(nx * ny * nc)
    q = prost.variable(2 * nx * ny * nc)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", 1, f, lmb))
    prob.add_function(q, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, nc))
    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * lmb)
    return prob, backend, prost.options(max_iters=max_iters, num_cback_calls=num_cback_calls, verbose=False)


def main(nx=700, ny=464, nc=3, max_iters=10000, verbose=True):
    f = synthetic.rof_image(nx, ny, nc, seed=1).astype(np.float64)           # :3-6  f in [0, 1], y fastest, then x, then channel
    grad = spmat_gradient2d(nx, ny, nc)                                       # :10
    lmb = 10                                                                  # :11

    u = prost.variable(nx * ny * nc)                                          # :15
    q = prost.variable(2 * nx * ny * nc)                                      # :16
    prob = prost.min_max_problem([u], [q])                                    # :18
    prob.add_function(u, prost.function.sum_1d("square", 1, f, lmb))          # :19
    prob.add_function(q, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1))   # :25-26
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, nc))              # :28

    backend = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.05 * lmb)   # :34-36
    gaps = []

    def pd_gap_callback(it, x, y):                                            # example_rof_pdgap.m:1-17
        g = (grad @ x).reshape(2 * nc, ny * nx).T
        en_prim = 0.5 * lmb * np.sum((x - f) ** 2) + np.sum(np.sqrt(np.sum(g ** 2, axis=1)))
        div = grad.T @ y
        en_dual = f @ div - (1 / (2 * lmb)) * np.sum(div ** 2)
        gaps.append((en_prim - en_dual) / (nx * ny))
        if verbose:
            print("it %5d primal_dual_gap=%.2e." % (it, gaps[-1]))
        return gaps[-1] < 1e-5

    opts = prost.options(max_iters=max_iters, interm_cb=pd_gap_callback, num_cback_calls=250, verbose=False)   # :43-46
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                 # :49
    elapsed = time.perf_counter() - t0
    prost.release()                                                           # :52
    if verbose:
        print("%s after %d iterations, %.3f s; final primal-dual gap %.2e" % (result["result"], result["iters"], elapsed, gaps[-1] if gaps else float("nan")))
    return result, gaps, u.val.reshape(nc, nx, ny)                            # :56 (imshow(reshape(u.val, [ny nx nc])))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
