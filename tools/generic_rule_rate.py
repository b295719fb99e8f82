"""The reference's default backend options (boyd, residual_iter = 1) on the GENERIC path: example_deblurring.m's shape -- min_problem,
two sparse constraint blocks on u (a blur operator with a square data term, the gradient with the TV norm), no function on u -- with
the step-size rule on the device (batches of iterations, one host wait each) and on the host (one round trip per iteration).
usage: generic_rule_rate.py [nx ny] [iters] [residual_iter] [warmup] [variant mask: 1 op+device, 2 op+host, 4 separate+device, 8 separate+host]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scipy.sparse as sp

import prost_amd as prost
from prost_amd import synthetic
from reference_matrices import spmat_gradient2d


def problem(nx, ny, seed=3):
    n = nx * ny
    k1 = sp.diags([np.ones(ny - 1), np.ones(ny), np.ones(ny - 1)], [-1, 0, 1]) / 3.0
    k2 = sp.diags([np.ones(nx - 1), np.ones(nx), np.ones(nx - 1)], [-1, 0, 1]) / 3.0
    B = sp.kron(k2, k1).tocsr()                                       # 3 x 3 box blur (example_deblurring.m:12-22 uses a motion kernel)
    shape = os.environ.get("PROST_RATE_BLUR", "xy")                   # experiments: "x" = 3 taps across columns (offsets -ny, 0, ny: 16-byte aligned operands), "y" = 3 taps along a column
    if shape == "x":
        B = sp.kron(k2, sp.identity(ny)).tocsr()
    elif shape == "y":
        B = sp.kron(sp.identity(nx), k1).tocsr()
    f = synthetic.rof_image(nx, ny, 1, seed=seed)
    fb = B @ f + 0.02 * np.random.default_rng(seed).standard_normal(n)
    u, v, g = prost.variable(n), prost.variable(n), prost.variable(2 * n)
    prob = prost.min_problem([u], [v, g])                             # :33
    prob.add_function(v, prost.function.sum_1d("square", 1, fb, 20.0, 0, 0))          # :34
    prob.add_function(g, prost.function.sum_norm2(2, False, "abs", 1, 0, 1, 0, 0))    # :35
    prob.add_constraint(u, v, prost.block.sparse(B))                  # :36
    prob.add_constraint(u, g, prost.block.sparse(spmat_gradient2d(nx, ny, 1)))        # :37
    return prob


def main(nx=700, ny=464, iters=3000, residual_iter=1, warmup=300, mask=15):
    prost.set_gpu(0); prost.set_precision("single")
    o = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    prob = problem(nx, ny)
    for name, dev, opf in (("operator in the prox kernels, rule on the device", True, True), ("operator in the prox kernels, rule on the host", False, True),
                           ("separate products, rule on the device", True, False), ("separate products, rule on the host", False, False)):
        mask, skip = mask >> 1, not (mask & 1)
        if skip:
            continue
        b = prost.backend.pdhg(stepsize="boyd", residual_iter=int(residual_iter))      # :40-41
        b[1]["allow_device_rules"] = dev
        b[1]["allow_op_fusion"] = 2 if opf else 0
        s = prost.Solver(prob, b, o)
        s.iterate(warmup)
        info = s.iterate(iters)
        st = s.state(vectors=False)
        print("deblurring-like %dx%d fp32, boyd R=%d, %-48s: %.0f it/s (%.4f ms per iteration), path %s, operator in prox kernels %s, device rule batches %s" % (
            nx, ny, residual_iter, name, iters / (info["ms"] * 1e-3), info["ms"] / iters, st["path"], st.get("operator_in_prox_kernels"), st.get("device_rule_batches")), flush=True)
        s.destroy()


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:7]])
