"""Runs only the single-kernel iteration (or the two-pass kernels) a few times: target for rocprofv3 --pmc."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prost_amd import _hip as hip
if os.environ.get("PROST_HIP_LIB"):          # A/B runs of kernel variants on one box
    hip.LIB_PATH = os.environ["PROST_HIP_LIB"]

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "iter"
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dtype = np.float32
n, m = N * N, 2 * N * N
rng = np.random.default_rng(0)
f = hip.DeviceArray.from_host(rng.random(n).astype(dtype))
x = [hip.DeviceArray.from_host(rng.random(n).astype(dtype)), hip.DeviceArray.zeros(n, dtype)]
y = [hip.DeviceArray.from_host((rng.random(m) - 0.5).astype(dtype)), hip.DeviceArray.zeros(m, dtype)]
d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = N, N, 1
d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
for i, (g, fv) in enumerate(zip([1, 0, 10, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0])):
    d.g_coeff_val[i] = g; d.f_coeff_val[i] = fv
d.g_coeff_ptr[1] = f.ptr.value
d.T_val, d.S_val = 0.25, 0.5
d.arith = int(os.environ.get("PROST_ARITH", "0"))
r4 = hip.DeviceArray.zeros(4, np.float64)
ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
for i in range(reps):
    a, b = i % 2, (i + 1) % 2
    if mode == "iter2res":
        t2 = (C.c_double * 2)(0.3, 0.29); s2 = (C.c_double * 2)(1.0, 1.03); th2 = (C.c_double * 2)(0.9, 0.91)
        hip.check(hip.fn("fused_iteration2", dtype)(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, None, t2, s2, th2, cols, r4.ptr, ws.ptr, None))
    elif mode == "iter2":
        t2 = (C.c_double * 2)(0.3, 0.29); s2 = (C.c_double * 2)(1.0, 1.03); th2 = (C.c_double * 2)(0.9, 0.91)
        hip.check(hip.fn("fused_iteration2", dtype)(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, None, t2, s2, th2, cols, None, None, None))
    elif mode.startswith("iterk"):        # iterk4 / iterk3res ...
        K = int(mode[5]); tk = (C.c_double * 4)(0.3, 0.29, 0.28, 0.27); sk = (C.c_double * 4)(1.0, 1.03, 1.06, 1.09); thk = (C.c_double * 4)(0.9, 0.91, 0.92, 0.93)
        res = mode.endswith("res")
        hip.check(hip.lib().prost_hip_fused_iterationk_f32(C.byref(d), K, x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, tk, sk, thk, cols, r4.ptr if res else None, ws.ptr if res else None, None))
    elif mode == "iter":
        hip.check(hip.fn("fused_iteration", dtype)(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 0, cols, None, None, None))
    else:
        hip.check(hip.fn("fused_primal", dtype)(C.byref(d), x[b].ptr, x[a].ptr, y[a].ptr, None, hip.dbl(0.3), 1, 0, None, None, None))
        hip.check(hip.fn("fused_dual", dtype)(C.byref(d), y[b].ptr, y[a].ptr, x[b].ptr, x[a].ptr, hip.dbl(1.0), hip.dbl(0.9), 1, None, None, None))
hip.sync()
print("done", mode, cols)
