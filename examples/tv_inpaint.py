"""matlab/examples/example_tv_inpaint.m on the MI355X build: TV inpainting -- the 0 / 1 mask enters as coefficient a of the square data
term (sum_1d('square', m, f, lmb): ElemOperation1D skips the function where a == 0), vectorial TV, PDHG with Boyd's residual
balancing.  Synthetic image and mask instead of images/lion.png / maske2.png.  usage: python examples/tv_inpaint.py [nx ny nc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic


def main(nx=700, ny=464, nc=3, max_iters=50000, verbose=True, tol=1e-7):
    rng = np.random.default_rng(42)                                           # :1
    f = synthetic.rof_image(nx, ny, nc, seed=2).astype(np.float64)            # :5-9
    hole = np.zeros((nx, ny), dtype=bool)                                     # :6,10 the mask image: text-like strokes, here stripes + dots
    hole[:, ::9] = True
    hole[rng.integers(0, nx, 400), rng.integers(0, ny, 400)] = True
    m = np.tile(1.0 - hole.reshape(-1), nc)                                   # :10-11  m = 1 - (mask > 0), per channel
    lmb = 7                                                                   # :16

    u = prost.variable(nx * ny * nc)                                          # :20
    q = prost.variable(2 * nx * ny * nc)                                      # :21
    prob = prost.min_max_problem([u], [q])                                    # :23
    prob.add_function(u, prost.function.sum_1d("square", m, f, lmb))          # :24
    prob.add_function(q, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1))   # :30-31
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, nc))              # :33

    backend = prost.backend.pdhg(stepsize="boyd", residual_iter=10)           # :39-40
    opts = prost.options(max_iters=max_iters, num_cback_calls=250, verbose=False, tol_rel_primal=tol, tol_rel_dual=tol,
                         tol_abs_dual=tol, tol_abs_primal=tol)                # :44-50
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                 # :53
    elapsed = time.perf_counter() - t0
    # :61-63 the energy the script prints: (lmb / 2) sum (m (u - f))^2 + sum |grad u| (norm over channels and directions)
    img = np.asarray(u.val).reshape(nc, nx, ny)
    gx = np.diff(img, axis=1, append=img[:, -1:, :]); gy = np.diff(img, axis=2, append=img[:, :, -1:])
    energy = 0.5 * lmb * float(np.sum((m * (np.asarray(u.val) - f)) ** 2)) + float(np.sqrt((gx ** 2 + gy ** 2).sum(axis=0)).sum())
    if verbose:
        print("%s after %d iterations, %.3f s (%d launches of two iterations); energy %.4f" % (result["result"], result["iters"], elapsed, int(result["pair_launches"]), energy))
    return result, energy, m, f, np.asarray(u.val)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
