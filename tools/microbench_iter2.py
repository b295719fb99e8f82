"""Micro-benchmark at the kernel C ABI: single-iteration kernel vs the two-iterations-per-launch kernel
(ROF 4096^2 fp32 by default).  usage: microbench_iter2.py [N] [cols,cols,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from prost_amd import _hip as hip

if os.environ.get("PROST_HIP_LIB"):          # A/B runs of kernel variants on one box
    hip.LIB_PATH = os.environ["PROST_HIP_LIB"]


def main(N=4096, cols_list=(0, 12, 18, 24, 30, 36, 42, 54, 66), iters=200, dtype=np.float32, gfn="square", ffn="ind_leq0", quick=False):
    hip.require_device()
    n, m = N * N, 2 * N * N
    rng = np.random.default_rng(0)
    f = hip.DeviceArray.from_host(rng.random(n).astype(dtype))
    x = [hip.DeviceArray.from_host(rng.random(n).astype(dtype)), hip.DeviceArray.zeros(n, dtype)]
    y = [hip.DeviceArray.from_host((rng.random(m) - 0.5).astype(dtype)), hip.DeviceArray.zeros(m, dtype)]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = N, N, 1
    d.g_fn = hip.FN_ID[gfn]; d.f_fn = hip.FN_ID[ffn]
    gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
    for i in range(7):
        d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
    d.g_coeff_ptr[1] = f.ptr.value
    d.T_val, d.S_val = 0.25, 0.5
    I1 = hip.fn("fused_iteration", dtype); I2 = hip.fn("fused_iteration2", dtype)
    L_ = hip.lib()
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        hip.check(L_.prost_hip_event_create(C.byref(e)))
    tau = (C.c_double * 2)(0.3, 0.29); sigma = (C.c_double * 2)(1.0, 1.03); theta = (C.c_double * 2)(0.9, 0.91)
    esz = np.dtype(dtype).itemsize

    def timed(run):
        run(10); hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters); hip.check(L_.prost_hip_event_record(ev[1], None))
        hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        return ms.value / iters

    def run1(k):
        for i in range(k):
            a, b = i % 2, (i + 1) % 2
            hip.check(I1(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 0, 0, None, None, None))
    t = timed(run1) if not quick else 1.0
    print("single  N=%d %s cols=auto: %.4f ms/launch = %.4f ms/iteration, %.0f it/s, algorithmic (11 floats/iter) %.0f GB/s"
          % (N, np.dtype(dtype).name, t, t, 1e3 / t, 11 * n * esz / 1e9 / (t * 1e-3)), flush=True)
    xm = hip.DeviceArray.zeros(n, dtype); ym = hip.DeviceArray.zeros(m, dtype); r4 = hip.DeviceArray.zeros(4, np.float64)
    ws = hip.DeviceArray(L_.prost_hip_reduce_workspace_bytes() // 8, np.float64)
    for cols, mode in [(c, 0) for c in cols_list] + ([] if quick else [(cols_list[-1], 1), (cols_list[-1], 2), (cols_list[-1], 3)]):
        def run2(k):
            for i in range(k):
                a, b = i % 2, (i + 1) % 2
                hip.check(I2(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, xm.ptr if mode & 1 else None, ym.ptr if mode & 1 else None, tau, sigma, theta, cols,
                             r4.ptr if mode & 2 else None, ws.ptr if mode & 2 else None, None))
        t = timed(run2)
        if quick:
            t = min(t, timed(run2), timed(run2))
        print("double  mode=%d N=%d %s cols=%-4d: %.4f ms/launch = %.4f ms/iteration, %.0f it/s, algorithmic (22 floats/launch) %.0f GB/s, kernel moves (7 floats/launch) %.0f GB/s"
              % (mode, N, np.dtype(dtype).name, cols, t, t / 2, 2e3 / t, 22 * n * esz / 1e9 / (t * 1e-3), 7 * n * esz / 1e9 / (t * 1e-3)), flush=True)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dt = np.float64 if len(sys.argv) > 3 and sys.argv[3] == "f64" else np.float32
    if len(sys.argv) > 2:
        main(N, tuple(int(c) for c in sys.argv[2].split(",")), dtype=dt, gfn=sys.argv[4] if len(sys.argv) > 4 else "square",
             ffn=sys.argv[5] if len(sys.argv) > 5 else "ind_leq0", quick=os.environ.get("QUICK", "0") == "1")
    else:
        main(N)
