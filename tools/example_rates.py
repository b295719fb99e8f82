import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prost_amd as prost
from prost_amd import synthetic
prost.set_precision("single")
for (nx, ny, L) in ((700, 464, 3), (700, 464, 1), (1024, 1024, 3), (2048, 2048, 3)):
    prob, u, q, f = synthetic.rof_problem(nx, ny, L)
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    s = prost.Solver(prob, b, prost.options(max_iters=10**9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0))
    s.iterate(2000); info = s.iterate(10000)
    print(nx, ny, L, "%.1f it/s, %.2f us/iteration" % (10000 / (info["ms"] * 1e-3), info["ms"] * 1e3 / 10000), {k: round(v["avg_ms"] * 1e3, 2) for k, v in info["kernels"].items()})
    s.destroy()
