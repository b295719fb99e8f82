"""matlab/examples/example_rof_primal.m, line for line, on the MI355X build (reference lines cited).

The MATLAB script reads images/lion.png (700 x 464 RGB); no image file travels with this repo, so a synthetic RGB image of the same
size stands in (prost_amd.synthetic.rof_image).  Everything from `u = prost.variable(...)` on is the reference's description
unchanged: the problem in its PRIMAL form (prost.min_problem: the regulariser sits on the constrained variable g = grad u, the PDHG
backend derives prox_f* from it by Moreau's identity), the data term spread over three sub-variables, the gradient handed over as
a sparse matrix, the backend options boyd / residual_iter = 1, the primal-dual gap callback of example_rof_pdgap.m.
On this build the description runs the one-kernel PDHG iterations (result["path"] = "pdhg:fused-grad2d(sparse)") with the step-size
rule evaluated on the device.
usage: python examples/rof_primal_sub_variables.py [nx ny nc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic
from rof_rgb_gap_callback import spmat_gradient2d


def describe(nx=700, ny=464, nc=3, backend_opts=None):
    """the problem and the backend of example_rof_primal.m:1-36"""
    f = synthetic.rof_image(nx, ny, nc, seed=1).astype(np.float64)           # :3-6  f in [0, 1], y fastest, then x, then channel
    grad = spmat_gradient2d(nx, ny, nc)                                       # :10
    lmb = 10                                                                  # :11

    u = prost.variable(nx * ny * nc)                                          # :15
    g = prost.variable(2 * nx * ny * nc)                                      # :16
    u1 = prost.sub_variable(u, 100)                                           # :19  "Example on how to use sub-variables"
    u2 = prost.sub_variable(u, 500)                                           # :20
    u3 = prost.sub_variable(u, nx * ny * nc - 600)                            # :21

    prob = prost.min_problem([u], [g])                                        # :23
    prob.add_function(u1, prost.function.sum_1d("square", 1, f[:100], lmb, 0, 0))        # :24
    prob.add_function(u2, prost.function.sum_1d("square", 1, f[100:600], lmb, 0, 0))     # :25
    prob.add_function(u3, prost.function.sum_1d("square", 1, f[600:], lmb, 0, 0))        # :26
    prob.add_function(g, prost.function.sum_norm2(2 * nc, False, "abs", 1, 0, 1, 0, 0))  # :27
    prob.add_constraint(u, g, prost.block.sparse(grad))                       # :28

    backend = prost.backend.pdhg(stepsize="boyd", residual_iter=1, alg2_gamma=0.05 * lmb, tau0=1, sigma0=1)   # :32-36
    if backend_opts:
        backend[1].update(backend_opts)
    return prob, backend, u, f, grad, lmb


def main(nx=700, ny=464, nc=3, max_iters=10000, verbose=True, backend_opts=None):
    prob, backend, u, f, grad, lmb = describe(nx, ny, nc, backend_opts)
    gaps = []

    def pd_gap_callback(it, x, y):                                            # :38-39, example_rof_pdgap.m:1-17
        gx = (grad @ x).reshape(2 * nc, ny * nx).T
        en_prim = 0.5 * lmb * np.sum((x - f) ** 2) + np.sum(np.sqrt(np.sum(gx ** 2, axis=1)))
        div = grad.T @ y
        en_dual = f @ div - (1 / (2 * lmb)) * np.sum(div ** 2)
        gaps.append((en_prim - en_dual) / (nx * ny))
        if verbose:
            print("it %5d primal_dual_gap=%.2e." % (it, gaps[-1]))
        return False                                                          # (:41-44: the example lets the tolerances stop the run)

    opts = prost.options(max_iters=max_iters, interm_cb=pd_gap_callback, num_cback_calls=250, verbose=False)   # :41-44
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                 # :47
    elapsed = time.perf_counter() - t0
    if verbose:
        print("%s after %d iterations, %.3f s (%s); final primal-dual gap %.2e" % (result["result"], result["iters"], elapsed, result.get("path"), gaps[-1] if gaps else float("nan")))
    return result, gaps, u.val.reshape(nc, nx, ny), elapsed                   # :52 (imshow(reshape(u.val, [ny nx nc])))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
