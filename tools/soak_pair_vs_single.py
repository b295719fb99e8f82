"""Long trajectories: pair launches vs single launches must stay bit-identical while the iterates converge
(saturated duals, flat regions with exact zeros, tiny residuals) -- exercises the rare fallback paths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prost_amd as prost
from prost_amd import synthetic

# usage: soak_pair_vs_single.py [n] [iters] [single|double] [planes] [channels]   (planes > 0: volumetric TV n x n x planes, the 3-D
# double-iteration kernel; channels > 1: vectorial TV with that many channels, the multi-channel double-iteration kernel)
prost.set_precision(sys.argv[3] if len(sys.argv) > 3 else "single")
planes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
channels = int(sys.argv[5]) if len(sys.argv) > 5 else 1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
o = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
for step in ("alg2", "alg1", "boyd"):
    st = {}
    for pair in (True, False):
        prob, u, q, f = synthetic.tv3d_problem(n, n, planes) if planes else synthetic.rof_problem(n, n, channels)
        b = prost.backend.pdhg(stepsize=step, residual_iter=10, alg2_gamma=0.5)
        b[1]["allow_pair_kernel"] = pair
        s = prost.Solver(prob, b, o)
        s.iterate(iters)
        st[pair] = s.state(); s.destroy()
    same = all(np.array_equal(st[True][v], st[False][v]) for v in "xyzw")
    x = st[True]["x"]
    print(step, "n=%d planes=%d channels=%d iters=%d" % (n, planes, channels, iters), "identical:", same, "finite:", bool(np.isfinite(x).all()), "primal_res", st[True]["primal_res"], st[False]["primal_res"],
          "zeros in y: %.3f" % float((st[True]["y"] == 0).mean()), flush=True)
    assert same
