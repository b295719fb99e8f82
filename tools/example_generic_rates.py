"""Iteration rates of the reference's examples whose operator is not a lone gradient, built AS WRITTEN (examples/multilabel_fast.py,
multilabel_tight.py, deblurring.py), fp32, their own backend options: operator inside the prox launches (allow_op_fusion 2) against
separate products (0), and the default.   usage: example_generic_rates.py [n] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
import prost_amd as prost
import deblurring
import multilabel_fast
import multilabel_tight


def main(n=512, iters=1500):
    prost.set_gpu(0); prost.set_precision("single")
    cases = [("example_multilabel_fast.m %dx%d, 3 labels" % (n, n), lambda: multilabel_fast.describe(n, n)[:3]),
             ("example_multilabel_tight.m %dx%d, 3 labels" % (n, n), lambda: multilabel_tight.describe(n, n)[:3]),
             ("example_deblurring.m %dx%dx3, 15-tap motion blur" % (n, n), lambda: deblurring.describe(n, n, 3)[:3])]
    for name, make in cases:
        for label, opf in (("default", None), ("operator inside the prox launches", 2), ("separate products", 0)):
            prob, backend, opts = make()
            if opf is not None:
                backend[1]["allow_op_fusion"] = opf
            o = prost.options(max_iters=10 ** 9, num_cback_calls=0, verbose=False, tol_rel_primal=0, tol_rel_dual=0, tol_abs_primal=0, tol_abs_dual=0)
            s = prost.Solver(prob, backend, o)
            s.iterate(200)
            info = s.iterate(iters)
            st = s.state(vectors=False)
            print("%-52s %-36s %7.0f it/s (%.4f ms per iteration), path %s, operator in prox kernels %s, device rule batches %s" % (
                name, label, iters / (info["ms"] * 1e-3), info["ms"] / iters, st["path"], st.get("operator_in_prox_kernels"), st.get("device_rule_batches")), flush=True)
            s.destroy()


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:3]])
