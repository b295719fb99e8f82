"""Bandwidth of the generic ProxElemOperation<T, OP> kernel (include/prost/prox/prox_elem_operation.inl, instantiated in
tests/plugins/elem_operations.hip) against the library's run-time dispatched prox kernel on the same operands.
eval_prox times ONE synchronous launch (host clock around enqueue + wait), so large operands are used."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prost_amd as prost  # noqa: E402
from test_plugins import PLUGIN_DIR, plugin_1d, plugin_norm2  # noqa: E402

subprocess.check_call(["make", "-C", PLUGIN_DIR], stdout=subprocess.DEVNULL)
prost.load_plugin(os.path.join(PLUGIN_DIR, "build", "libprost_test_plugins.so"))
prost.set_gpu(0)
rng = np.random.default_rng(1)
for prec, size in (("single", 4), ("double", 8)):
    prost.set_precision(prec)
    count = 1 << 24
    for dim, inter in [(1, False), (2, False), (2, True), (3, False), (7, False), (7, True)]:
        n = count * dim
        arg = rng.standard_normal(n)
        Tau = np.ones(n)
        cs = (1, 0, 1, 0, 0, 0.25, 0)
        rows = []
        for label, fn in (("plugin", plugin_norm2("test:tpl:norm2:huber", dim, inter, *cs) if dim > 1 else plugin_1d("test:tpl:1d:huber", *cs)),
                          ("library", prost.function.sum_norm2(dim, inter, "huber", *cs) if dim > 1 else prost.function.sum_1d("huber", *cs))):
            ms = min(prost.eval_prox(fn, arg, 0.4, Tau)[1] for _ in range(4))
            # bytes the operation touches: arg + res (dim each) + tau_diag[0] per group
            b = count * (2 * dim + 1) * size
            rows.append("%s %.3f ms %.2f TB/s" % (label, ms, b / ms / 1e9))
        print("%s dim %d %s: %s" % (prec, dim, "interleaved" if inter else "planar", " | ".join(rows)), flush=True)
