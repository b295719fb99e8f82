"""example_rof_primal.m as written (examples/rof_primal_sub_variables.py: primal form, three sub-variables, sparse gradient, boyd
with residual_iter = 1): iteration rate on the fused path (rule on the device / on the host) and on the generic path.
usage: rof_primal_rate.py [nx ny nc] [iters] [stepsize residual_iter]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import prost_amd as prost
import rof_primal_sub_variables as ex


def main(nx=700, ny=464, nc=3, iters=3000, stepsize="boyd", residual_iter=1):
    prost.set_gpu(0); prost.set_precision("single")
    o = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
    for name, bo in (("fused, rule on the device", {}), ("fused, rule on the host", {"allow_device_rules": False}), ("generic path", {"allow_fused": False})):
        prob, backend, *_ = ex.describe(nx, ny, nc, dict(bo, stepsize=stepsize, residual_iter=int(residual_iter)))
        s = prost.Solver(prob, backend, o)
        s.iterate(300)
        info = s.iterate(iters)
        st = s.state(vectors=False)
        print("example_rof_primal %dx%dx%d fp32 %s R=%s, %-26s: %.0f it/s (%.4f ms per iteration), path %s, pair launches %s, device rule batches %s" % (
            nx, ny, nc, stepsize, residual_iter, name, iters / (info["ms"] * 1e-3), info["ms"] / iters, st["path"], st.get("pair_launches"), st.get("device_rule_batches")), flush=True)
        s.destroy()


if __name__ == "__main__":
    main(*([int(a) for a in sys.argv[1:5]] + sys.argv[5:7]))
