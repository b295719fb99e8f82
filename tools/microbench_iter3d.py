"""Micro-benchmark at the kernel C ABI: gradient3d two-pass kernels vs the one-kernel iteration (TV-3D, fp32).
usage: microbench_iter3d.py [nx ny L] [cols,cols,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from prost_amd import _hip as hip

if os.environ.get("PROST_HIP_LIB"):          # A/B runs of kernel variants on one box
    hip.LIB_PATH = os.environ["PROST_HIP_LIB"]


def main(nx=2048, ny=2048, Lp=64, cols_list=(0, 6, 12, 18, 24, 36, 48), iters=30, dtype=np.float32):
    hip.require_device()
    n, m = nx * ny * Lp, 3 * nx * ny * Lp
    rng = np.random.default_rng(0)
    f = hip.DeviceArray.from_host(rng.random(n, dtype=np.float32).astype(dtype))
    x = [hip.DeviceArray.from_host(rng.random(n, dtype=np.float32).astype(dtype)), hip.DeviceArray.zeros(n, dtype)]
    y = [hip.DeviceArray.from_host((rng.random(m, dtype=np.float32) - 0.5).astype(dtype)), hip.DeviceArray.zeros(m, dtype)]
    d = hip.FusedDesc(); d.is3d = 1; d.nx, d.ny, d.L = nx, ny, Lp
    d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
    gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
    for i in range(7):
        d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
    d.g_coeff_ptr[1] = f.ptr.value
    d.T_val, d.S_val = 1.0 / 6.0, 0.5
    L_ = hip.lib()
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        hip.check(L_.prost_hip_event_create(C.byref(e)))
    ws = hip.DeviceArray(L_.prost_hip_reduce_workspace_bytes() // 8, np.float64)
    esz = np.dtype(dtype).itemsize

    def timed(run):
        run(3); hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters); hip.check(L_.prost_hip_event_record(ev[1], None))
        hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        return ms.value / iters

    P, D, I3 = hip.fn("fused_primal", dtype), hip.fn("fused_dual", dtype), hip.fn("fused_iteration3d", dtype)

    def run2(k):
        for i in range(k):
            a, b = i % 2, (i + 1) % 2
            hip.check(P(C.byref(d), x[b].ptr, x[a].ptr, y[a].ptr, None, hip.dbl(0.3), 1, 0, None, ws.ptr, None))
            hip.check(D(C.byref(d), y[b].ptr, y[a].ptr, x[b].ptr, x[a].ptr, hip.dbl(1.0), hip.dbl(0.9), 1, None, ws.ptr, None))
    only = os.environ.get("X2_ONLY") == "1"
    if only:
        cols_list = ()
    t = 1.0 if only else timed(run2)
    print("two-pass   %dx%dx%d %s: %.4f ms/iteration, %.1f it/s, algorithmic (14 values/voxel) %.0f GB/s" % (nx, ny, Lp, np.dtype(dtype).name, t, 1e3 / t, 14 * n * esz / 1e9 / (t * 1e-3)), flush=True)
    for cols in cols_list:
        def run1(k):
            for i in range(k):
                a, b = i % 2, (i + 1) % 2
                hip.check(I3(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 1, cols, None, None, None))
        t = timed(run1)
        print("one-kernel cols=%-3d: %.4f ms/iteration, %.1f it/s, algorithmic %.0f GB/s, kernel moves (9 values/voxel) %.0f GB/s"
              % (cols, t, 1e3 / t, 14 * n * esz / 1e9 / (t * 1e-3), 9 * n * esz / 1e9 / (t * 1e-3)), flush=True)
    PW = hip.fn("fused_iteration3d_pw", dtype)
    for waves in (() if only else (4, 8)):
        for cols in (6, 12, 18):
            def runp(k):
                for i in range(k):
                    a, b = i % 2, (i + 1) % 2
                    hip.check(PW(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, cols, waves, None))
            t = timed(runp)
            print("planes-across-waves waves=%d cols=%-3d: %.4f ms/iteration, %.1f it/s, algorithmic %.0f GB/s" % (waves, cols, t, 1e3 / t, 14 * n * esz / 1e9 / (t * 1e-3)), flush=True)
    if L_.prost_hip_fused_iteration3d_x2_supported(C.byref(d), 0 if dtype == np.float32 else 1) == 1:
        two = lambda v: (C.c_double * 2)(v, v)
        r4x = hip.DeviceArray.zeros(4, np.float64)
        for xres, cols in [(False, int(c)) for c in os.environ.get("X2_COLS", "0,8,16,24,32,48,64").split(",")] + [(True, 0)]:
            def runx(k):
                for i in range(k):
                    a, b = i % 2, (i + 1) % 2
                    hip.check(hip.fn("fused_iteration3d_x2", dtype)(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, two(0.3), two(1.0), two(0.9), cols, r4x.ptr if xres else None, ws.ptr if xres else None, None))
            t = timed(runx) / 2
            print("two iterations per launch%s cols=%-3d: %.4f ms/iteration, %.1f it/s, algorithmic %.0f GB/s" % (" + residual sums" if xres else "", cols, t, 1e3 / t, 14 * n * esz / 1e9 / (t * 1e-3)), flush=True)
    yp = hip.DeviceArray.from_host((rng.random(m, dtype=np.float32) - 0.5).astype(dtype)); r4 = hip.DeviceArray.zeros(4, np.float64)

    def run_res2(k):
        for i in range(k):
            hip.check(P(C.byref(d), x[1].ptr, x[0].ptr, y[0].ptr, yp.ptr, hip.dbl(0.3), 1, 1, r4.ptr, ws.ptr, None))
            hip.check(D(C.byref(d), y[1].ptr, y[0].ptr, x[1].ptr, x[0].ptr, hip.dbl(1.0), hip.dbl(0.9), 1, r4.ptr, ws.ptr, None))

    def run_res1(k):
        for i in range(k):
            hip.check(I3(C.byref(d), x[1].ptr, y[1].ptr, x[0].ptr, y[0].ptr, yp.ptr, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 1, 0, r4.ptr, ws.ptr, None))
    if only:
        return
    print("residual iteration: two-pass %.4f ms, one-kernel %.4f ms" % (timed(run_res2), timed(run_res1)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 4:
        shape = tuple(int(v) for v in sys.argv[1:4])
        cl = tuple(int(c) for c in sys.argv[4].split(",")) if len(sys.argv) > 4 else (0, 6, 12, 18, 24, 36, 48)
        main(*shape, cols_list=cl, dtype=np.float64 if len(sys.argv) > 5 and sys.argv[5] == "f64" else np.float32)
    else:
        main()
