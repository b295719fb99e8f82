"""matlab/examples/example_tvl1.m on the MI355X build: TV-L1 denoising of an image with 25 % salt & pepper noise
(sum_1d('abs') data term, vectorial TV), PDHG with Boyd's residual balancing.  Synthetic image instead of images/fisch.jpg.
usage: python examples/tvl1_salt_and_pepper.py [nx ny nc]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic


def describe(nx=512, ny=384, nc=1, max_iters=50000):
    """the problem description, backend and options of example_tvl1.m:21-53 as written -> (prob, backend, opts, u, f, clean)"""
    rng = np.random.default_rng(42)                                           # :1
    clean = synthetic.rof_image(nx, ny, nc, seed=2).astype(np.float64)        # :5-8
    f = clean.copy()
    pix = rng.permutation(nx * ny * nc)                                       # :11-14  salt & pepper noise
    num_bad_pix = round(0.25 * nx * ny * nc)
    f[pix[:num_bad_pix]] = 1
    f[pix[num_bad_pix:2 * num_bad_pix]] = 0
    lmb = 1                                                                   # :19

    u = prost.variable(nx * ny * nc)                                          # :23
    q = prost.variable(2 * nx * ny * nc)                                      # :24
    prob = prost.min_max_problem([u], [q])                                    # :26
    prob.add_function(u, prost.function.sum_1d("abs", 1, f, lmb))             # :27
    prob.add_function(q, prost.function.sum_norm2(2 * nc, False, "ind_leq0", 1, 1, 1))   # :33-34
    prob.add_dual_pair(u, q, prost.block.gradient2d(nx, ny, nc))              # :36

    backend = prost.backend.pdhg(stepsize="boyd", residual_iter=10)           # :42-43
    opts = prost.options(max_iters=max_iters, num_cback_calls=250, verbose=False, tol_rel_primal=1e-7, tol_rel_dual=1e-7,
                         tol_abs_dual=1e-7, tol_abs_primal=1e-7)             # :47-53
    return prob, backend, opts, u, f, clean


def main(nx=512, ny=384, nc=1, max_iters=50000, verbose=True):
    prob, backend, opts, u, f, clean = describe(nx, ny, nc, max_iters)
    t0 = time.perf_counter()
    result = prost.solve(prob, backend, opts)                                 # :56
    elapsed = time.perf_counter() - t0
    err_noisy, err_denoised = float(np.mean(np.abs(f - clean))), float(np.mean(np.abs(u.val - clean)))
    if verbose:
        print("%s after %d iterations, %.3f s; mean |error| noisy %.4f -> denoised %.4f" % (result["result"], result["iters"], elapsed, err_noisy, err_denoised))
    return result, err_noisy, err_denoised


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:4]])
