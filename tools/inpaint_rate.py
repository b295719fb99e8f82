"""Rate of the inpainting shape (matlab/examples/example_tv_inpaint.m: RGB, 0 / 1 mask as coefficient a of the square data term) with and
without the two-iterations-per-launch kernels.  usage: inpaint_rate.py [N [L]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic


def main(N=2048, L=3):
    prost.set_precision("single")
    rng = np.random.default_rng(1)
    n = N * N * L
    mask = (rng.random(n) < 0.7).astype(np.float64)
    f = synthetic.rof_image(N, N, L, 3)
    u, q = prost.variable(n), prost.variable(2 * n)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", mask, f, 7.0))
    prob.add_function(q, prost.function.sum_norm2(2 * L, False, "ind_leq0", 1, 1, 1))
    prob.add_dual_pair(u, q, prost.block.gradient2d(N, N, L))
    o = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    for pair in (False, True):
        b = prost.backend.pdhg(stepsize="boyd", residual_iter=10)
        b[1]["allow_pair_kernel"] = pair
        s = prost.Solver(prob, b, o)
        s.iterate(400)
        info = s.iterate(2000)
        s.destroy()
        print("inpainting %dx%d L=%d fp32, pairs %s: %.0f it/s (%.4f ms/iteration)" % (N, N, L, "on " if pair else "off", 2000 / (info["ms"] * 1e-3), info["ms"] / 2000), flush=True)


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:3]))
