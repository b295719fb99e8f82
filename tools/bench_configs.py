#!/usr/bin/env python3
"""Supplementary measurements for the non-headline BASELINE configs (one JSON line each):
  c3  : 2048x2048x64 volumetric TV (gradient3d + sum_norm2(3) + sum_1d square), PDHG alg2
  c4  : TV-L1 flow-like (block.sparse W + gradient2d(L=2), sum_1d abs + sum_norm2(4) abs), ADMM
  c1  : 256x256 ROF (plumbing size)
usage: python tools/bench_configs.py c3 [nx ny L] | c4 [N [device|host|pdhg]] | c1
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import prost_amd as prost
from prost_amd import synthetic

ZERO_TOL = dict(tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)


def run(prob, backend, steps, warmup, floats_per_unit, units, name):
    s = prost.Solver(prob, backend, prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, **ZERO_TOL))
    s.iterate(warmup)
    info = s.iterate(steps)
    st = s.state()
    s.destroy()
    it_s = steps / (info["ms"] * 1e-3)
    out = {"config": name, "path": st["path"], "it_per_s": it_s, "ms_per_it": info["ms"] / steps, "steps": steps,
           "algorithmic_GBps": it_s * floats_per_unit * 4 * units / 1e9 if floats_per_unit else None,
           "finite": bool(np.isfinite(st["x"]).all()), "cg_iterations": st.get("cg_iterations")}
    print(json.dumps(out), flush=True)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c3"
    prost.set_precision("single")
    if which == "c3":
        nx, ny, L = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (2048, 2048, 64)
        prob, u, q, f = synthetic.tv3d_problem(nx, ny, L)
        b = prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5)
        run(prob, b, 100, 10, 14, nx * ny * L, "TV-3D %dx%dx%d fp32 PDHG alg2" % (nx, ny, L))
    elif which == "c4":
        N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
        n = N * N
        Ix = synthetic.rof_image(N, N, 1, 1) - 0.5
        Iy = synthetic.rof_image(N, N, 1, 2) - 0.5
        bvec = synthetic.rof_image(N, N, 1, 3) - 0.5
        W = sp.hstack([sp.diags(Ix), sp.diags(Iy)]).tocsc()
        u = prost.variable(2 * n)
        v, g = prost.variable(n), prost.variable(4 * n)
        prob = prost.min_problem([u], [v, g])
        prob.add_function(v, prost.function.sum_1d("abs", 1, bvec, 5.0))
        prob.add_function(g, prost.function.sum_norm2(4, False, "abs"))
        prob.add_constraint(u, v, prost.block.sparse(W))
        prob.add_constraint(u, g, prost.block.gradient2d(N, N, 2))
        modes = (True, False) if len(sys.argv) <= 3 else () if sys.argv[3] == "pdhg" else (sys.argv[3] == "device",)
        for device_cg in modes:
            b = prost.backend.admm(rho0=1)
            b[1]["device_cg"] = device_cg
            b[1]["cg_graph"] = os.environ.get("PROST_CG_GRAPH", "0") == "1"
            run(prob, b, 300, 50, None, n, "TV-L1 flow-like %dx%d fp32 ADMM (block.sparse + gradient2d L=2), %s CG scalars" % (N, N, "device" if device_cg else "host"))
        if len(sys.argv) <= 3 or sys.argv[3] == "pdhg":
            run(prob, prost.backend.pdhg(stepsize="boyd", residual_iter=10), 5000, 1000, None, n, "TV-L1 flow-like %dx%d fp32 PDHG generic path" % (N, N))
    else:
        prob, u, q, f = synthetic.rof_problem(256, 256)
        run(prob, prost.backend.pdhg(stepsize="alg2", residual_iter=10, alg2_gamma=0.5), 50000, 10000, 11, 256 * 256, "ROF 256x256 fp32 PDHG alg2")


if __name__ == "__main__":
    main()
