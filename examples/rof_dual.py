"""matlab/examples/example_rof_dual.m, line for line, on the MI355X build (reference lines cited): ROF written as its DUAL problem --
min over the dual variable q of  I(|q| <= 1) + (1 / (2 lmb)) |w + lmb f|^2  subject to  w = -grad' q  (prost.min_problem with a sparse
block), goldstein steps with residual_iter = 100 and the primal-dual gap callback of example_rof_pdgap.m.  The image is read back
from the DUAL variables of the dual problem (get_all_variables, :46-52).  Synthetic image instead of images/dog.png.
usage: python examples/rof_dual.py [nx ny nc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic
from rof_rgb_gap_callback import spmat_gradient2d


def describe(nx=256, ny=192, nc=3, max_iters=20000, num_cback_calls=100):
    """-> (prob, backend, opts, f, grad, lmb): the description of example_rof_dual.m:10-41 without the callback"""
    f = synthetic.rof_image(nx, ny, nc, seed=4).astype(np.float64)            # :3-5
    grad = spmat_gradient2d(nx, ny, nc)                                       # :9
    lmb = 0.3                                                                 # :10
    q = prost.variable(2 * nx * ny * nc)                                      # :14
    w = prost.variable(nx * ny * nc)                                          # :15
    prob = prost.min_problem([q], [w])                                        # :17
    prob.add_function(q, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1, 0, 0))    # :18-19
    prob.add_function(w, prost.function.sum_1d("square", 1, -f * lmb, 1 / lmb, 0, 0))          # :20-21
    prob.add_constraint(q, w, prost.block.sparse((-grad.T).tocsc()))          # :22
    backend = prost.backend.pdhg(stepsize="goldstein", residual_iter=100)     # :26-27
    opts = prost.options(max_iters=max_iters, num_cback_calls=num_cback_calls, verbose=False)   # :32-35
    return prob, backend, opts, f, grad, lmb


def main(nx=256, ny=192, nc=3, max_iters=20000, verbose=True):
    prob, backend, opts, f, grad, lmb = describe(nx, ny, nc, max_iters)
    gaps = []

    def pd_gap_callback(it, x, y):                                            # :29-30: example_rof_pdgap(it, y, x, ...) -- the roles are exchanged
        u_, q_ = y[:nx * ny * nc], x[:2 * nx * ny * nc]
        g = (grad @ u_).reshape(2 * nc, ny * nx).T
        en_prim = 0.5 * lmb * np.sum((u_ - f) ** 2) + np.sum(np.sqrt(np.sum(g ** 2, axis=1)))
        div = grad.T @ q_
        en_dual = f @ div - (1 / (2 * lmb)) * np.sum(div ** 2)
        gaps.append((en_prim - en_dual) / (nx * ny))
        if verbose:
            print("it %5d primal_dual_gap=%.2e." % (it, gaps[-1]))
        return gaps[-1] < 1e-5

    opts["interm_cb"] = pd_gap_callback
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                 # :38
    elapsed = time.perf_counter() - t0
    u = prost.variable(nx * ny * nc)                                          # :46-47: the dual variables of the dual problem are the image
    prost.get_all_variables(result, [], [], [u], [])
    if verbose:
        print("%s after %d iterations, %.3f s on %s; last gap %s" % (result["result"], result["iters"], elapsed, result["path"], gaps[-1] if gaps else None))
    return result, gaps, np.asarray(u.val).reshape(nc, nx, ny), f             # :51


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
