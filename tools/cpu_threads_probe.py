import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, prost_amd as prost
from prost_amd import synthetic
n = 4096
prob, u, q, f = synthetic.rof_problem(n, n)
prob.finalize()
b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
o = prost.options(max_iters=10**6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
print("cpus", os.cpu_count())
for t in (1, 8, 16, 32, 64, 128, 256):
    oracle.set_num_threads(t)
    s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float32); s.initialize(); s.iterate(1)
    t0 = time.time(); s.iterate(4); el = time.time() - t0
    print(t, "threads:", 4 / el, "it/s", flush=True)
