"""Wall time of problem setup (finalize + solver_create: upload, preconditioners, normest, backend init)
for the ROF configurations.  usage: setup_time.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prost_amd as prost
from prost_amd import synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prost.set_precision("single")
t0 = time.perf_counter()
prob, u, q, f = synthetic.rof_problem(N, N)
t1 = time.perf_counter()
prob.finalize()
t2 = time.perf_counter()
b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
s = prost.Solver(prob, b, prost.options(max_iters=100, num_cback_calls=0, verbose=False))
t3 = time.perf_counter()
s.iterate(10)
t4 = time.perf_counter()
print('--- second solver_create ---', file=sys.stderr, flush=True)
s2 = prost.Solver(prob, b, prost.options(max_iters=100, num_cback_calls=0, verbose=False))
t5 = time.perf_counter()
print('--- state ---', file=sys.stderr, flush=True)
st = s.state()
t6 = time.perf_counter()
print("N=%d synthetic image %.3fs, finalize %.3fs, solver_create (first, incl. context) %.3fs, 10 iterations %.4fs, second solver_create %.3fs, state readback %.3fs"
      % (N, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5))
