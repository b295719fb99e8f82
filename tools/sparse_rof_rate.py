"""ROF with the gradient given as a SPARSE MATRIX (block.sparse(spmat_gradient2d), as example_rof_primal.m / example_deblurring.m /
example_nonconvex_rof.m do) -- the generic PDHG path: iteration rate at N x N, fp32.   usage: sparse_rof_rate.py [N] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import prost_amd as prost
from prost_amd import synthetic
from reference_matrices import spmat_gradient2d


def main(n=2048, iters=300, stencils=1, stepsize="alg2", residual_iter=10):
    """stencils = 0: the generic path (row-pattern products), the A/B of the recognition (set_quirks(sparse_stencils=0))"""
    prost.set_gpu(0); prost.set_precision("single")
    prost.set_quirks(sparse_stencils=int(stencils) & 1)
    f = synthetic.rof_image(n, n, 1, 42)
    u, q = prost.variable(n * n), prost.variable(2 * n * n)
    prob = prost.min_max_problem([u], [q])
    prob.add_function(u, prost.function.sum_1d("square", 1, f, 10.0))
    prob.add_function(q, prost.function.sum_norm2(2, False, "ind_leq0", 1, 1, 1))
    # stencils = 2: the same description with prost.block.gradient2d (other preconditioners at the border: the reference's constants)
    prob.add_dual_pair(u, q, prost.block.gradient2d(n, n, 1) if int(stencils) == 2 else prost.block.sparse(spmat_gradient2d(n, n, 1)))
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    b = prost.backend.pdhg(stepsize=stepsize, residual_iter=int(residual_iter), alg2_gamma=0.5)
    t0 = time.time(); s = prost.Solver(prob, b, o); t1 = time.time()
    s.iterate(200); s.state(vectors=False)
    t2 = time.time(); s.iterate(iters); st = s.state(vectors=False); t3 = time.time()
    print("sparse-gradient ROF %dx%d fp32 %s R=%s, path %s: setup %.2f s, %.1f it/s (%.3f ms per iteration), pair launches %d" % (n, n, stepsize, residual_iter, st["path"], t1 - t0, iters / (t3 - t2), (t3 - t2) / iters * 1e3, st["pair_launches"]))
    s.destroy()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(*([int(v) for v in a[:3]] + a[3:]))
