"""Micro-benchmark of the fused ROF passes straight at the kernel C ABI (development aid)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prost_amd import _hip as hip

def main(N=4096, iters=200, dtype=np.float32, L=1):
    hip.require_device()
    n, m = N * N * L, 2 * N * N * L
    rng = np.random.default_rng(0)
    f = hip.DeviceArray.from_host(rng.random(n).astype(dtype))
    x = [hip.DeviceArray.from_host(rng.random(n).astype(dtype)), hip.DeviceArray.zeros(n, dtype)]
    y = [hip.DeviceArray.from_host((rng.random(m) - 0.5).astype(dtype)), hip.DeviceArray.zeros(m, dtype)]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = N, N, L
    d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
    gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
    for i in range(7):
        d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
    d.g_coeff_ptr[1] = f.ptr.value
    d.T_val, d.S_val = 0.25, 0.5
    ws = hip.DeviceArray(hip.lib().prost_hip_reduce_workspace_bytes() // 8, np.float64)
    out2 = hip.DeviceArray.zeros(2, np.float64)
    P = hip.fn("fused_primal", dtype); D = hip.fn("fused_dual", dtype)
    L_ = hip.lib()
    ev = [C.c_void_p() for _ in range(4)]
    for e in ev: hip.check(L_.prost_hip_event_create(C.byref(e)))
    def run(k, which, res=False):
        for i in range(k):
            a, b = i % 2, (i + 1) % 2
            if which in ("p", "both"):
                hip.check(P(C.byref(d), x[b].ptr, x[a].ptr, y[a].ptr, y[b].ptr, hip.dbl(0.3), 1, 1, out2.ptr if res else None, ws.ptr, None))
            if which in ("d", "both"):
                hip.check(D(C.byref(d), y[b].ptr, y[a].ptr, x[b].ptr, x[a].ptr, hip.dbl(1.0), hip.dbl(0.9), 1, out2.ptr if res else None, ws.ptr, None))
    esz = np.dtype(dtype).itemsize
    for which, floats in (("p", 5), ("d", 6), ("both", 11)):
        run(10, which); hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters, which); hip.check(L_.prost_hip_event_record(ev[1], None))
        hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        t = ms.value / iters
        gb = floats * N * N * L * esz / 1e9
        print("N=%d L=%d %s %-5s: %.3f ms/iter  %.1f GB/s (alg. bytes)  %.1f it/s" % (N, L, np.dtype(dtype).name, which, t, gb / (t * 1e-3), 1e3 / t), flush=True)
    run(10, "both", True); hip.sync()
    hip.check(L_.prost_hip_event_record(ev[0], None)); run(50, "both", True); hip.check(L_.prost_hip_event_record(ev[1], None))
    hip.check(L_.prost_hip_event_synchronize(ev[1]))
    ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
    print("  with residual sums: %.3f ms/iter" % (ms.value / 50), flush=True)

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    main(N)
    if len(sys.argv) > 2:
        main(N, dtype=np.float64)
        main(N // 2, L=3)


def bench_iter(N=4096, iters=200, dtype=np.float32, L=1, cols_list=(0, 4, 6, 8, 12, 16, 18, 24, 32)):
    hip.require_device()
    n, m = N * N * L, 2 * N * N * L
    rng = np.random.default_rng(0)
    f = hip.DeviceArray.from_host(rng.random(n).astype(dtype))
    x = [hip.DeviceArray.from_host(rng.random(n).astype(dtype)), hip.DeviceArray.zeros(n, dtype)]
    y = [hip.DeviceArray.from_host((rng.random(m) - 0.5).astype(dtype)), hip.DeviceArray.zeros(m, dtype)]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = N, N, L
    d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
    gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
    for i in range(7):
        d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
    d.g_coeff_ptr[1] = f.ptr.value
    d.T_val, d.S_val = 0.25, 0.5
    I = hip.fn("fused_iteration", dtype)
    L_ = hip.lib()
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev: hip.check(L_.prost_hip_event_create(C.byref(e)))
    esz = np.dtype(dtype).itemsize
    for cols in cols_list:
        def run(k):
            for i in range(k):
                a, b = i % 2, (i + 1) % 2
                hip.check(I(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, hip.dbl(0.3), hip.dbl(1.0), hip.dbl(0.9), 1, 1, 0, cols, None, None, None))
        run(10); hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters); hip.check(L_.prost_hip_event_record(ev[1], None))
        hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        t = ms.value / iters
        print("single-kernel iteration N=%d L=%d %s cols=%-3d: %.3f ms/iter  %.1f it/s  actual 7-float traffic %.0f GB/s, algorithmic (11 floats) %.0f GB/s"
              % (N, L, np.dtype(dtype).name, cols, t, 1e3 / t, 7 * N * N * L * esz / 1e9 / (t * 1e-3), 11 * N * N * L * esz / 1e9 / (t * 1e-3)), flush=True)


if __name__ == "__main__" and len(sys.argv) > 3:
    bench_iter(int(sys.argv[1]))
    bench_iter(int(sys.argv[1]), dtype=np.float64, cols_list=(0, 16))
