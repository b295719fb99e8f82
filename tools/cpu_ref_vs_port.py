"""CPU timing of the REAL reference (oracle/_ref: BackendPDHG, Problem, prox functors compiled from /root/reference,
thrust host backend; gradient stencils delegated to the oracle through the Block plugin interface) next to the
restatement (oracle/prost_oracle.cpp) on the same ROF problem -- build container only (shows that the
restatement used as bench.py's cpu_baseline is not slower than the reference's own CPU path).
usage: OMP_NUM_THREADS=1 python tools/cpu_ref_vs_port.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
import prost_amd as prost
from oracle import ref
from prost_amd import synthetic


def main(sizes):
    assert ref.available() or ref.build(), "oracle/_ref is not built and /root/reference is absent"
    b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
    o = prost.options(max_iters=10 ** 6, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    for N in sizes:
        iters = max(10, int(4e7 / (N * N)))
        prob, u, q, f = synthetic.rof_problem(N, N)
        prob.finalize()
        R = ref.RefProblem(prob.data, prob.nrows, prob.ncols, np.float32)
        t0 = time.perf_counter(); r = R.pdhg(b[1], o, iters); t_ref = time.perf_counter() - t0
        s = oracle.Solver(prob.data, prob.nrows, prob.ncols, b, o, np.float32)
        s.initialize()
        t0 = time.perf_counter(); s.iterate(iters); t_port = time.perf_counter() - t0
        same = np.array_equal(r["x"].astype(np.float32), s.state()["x"].astype(np.float32))
        print("N=%d fp32, %d iterations (incl. setup for the reference run): reference build %.1f it/s, restatement (OMP_NUM_THREADS=%s) %.1f it/s, iterates identical: %s"
              % (N, iters, iters / t_ref, os.environ.get("OMP_NUM_THREADS", "all"), iters / t_port, same), flush=True)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [256, 1024])
