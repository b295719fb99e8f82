"""401 iterations of the 2048 x 2048 x 64 volume with and without the double-iteration kernel: every element of x, y and the previous
iterate compared ON THE DEVICE (solver_compare).  usage: soak3d_fullsize.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prost_amd as prost
from prost_amd import synthetic
prost.set_precision("single")
nx, ny, L, k = 2048, 2048, 64, 401
f = synthetic.rof_image(nx, ny, L, 7)
o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
sol = {}
for pair in (True, False):
    prob, u, q, _ = synthetic.tv3d_problem(nx, ny, L, f=f)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    b[1]["allow_pair_kernel"] = pair
    s = prost.Solver(prob, b, o)
    info = s.iterate(k)
    sol[pair] = s
    print("pair" if pair else "single", "%.1f it/s" % (k / (info["ms"] * 1e-3)), s.state(vectors=False)["primal_res"], flush=True)
    del prob
print(sol[True].compare(sol[False]))
