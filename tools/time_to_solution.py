"""Wall time of a complete prost.solve() call (setup + iterations to the reference example's tolerance + read-back)
for ROF at N x N, fp32, with the settings of example_rof_primaldual.m (alg2, residual_iter 10, tol 1e-4, max 10000).
usage: time_to_solution.py [N]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prost_amd as prost
from prost_amd import synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prost.set_precision("single")
prob, u, q, f = synthetic.rof_problem(N, N)
b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
o = prost.options(max_iters=10000, num_cback_calls=0, verbose=False, tol_rel_primal=1e-4, tol_rel_dual=1e-4, tol_abs_primal=1e-4, tol_abs_dual=1e-4)
prost.solve(synthetic.rof_problem(64, 64)[0], b, prost.options(max_iters=10, num_cback_calls=0, verbose=False))     # context creation outside the timing
os.environ["PROST_TRACE_SOLVE"] = "1"          # native stage times on stderr
from prost_amd import _capi
_cmd = _capi.command
def _timed_command(name, *a, **k):
    t = time.perf_counter(); out = _cmd(name, *a, **k)
    if name == "solve_problem":
        print("python: prost_command('solve_problem') incl. marshalling both ways %.3f s" % (time.perf_counter() - t), file=sys.stderr)
    return out
_capi.command = _timed_command
t0 = time.perf_counter()
r = prost.solve(prob, b, o)
t1 = time.perf_counter()
print("ROF %dx%d fp32: prost.solve -> '%s' after %d iterations, %.3f s wall (setup + iterations + read-back of x, y, z, w)"
      % (N, N, r["result"], r["iters"], t1 - t0))
