"""Tolerance-class (FMAD) pair kernel against the exact one at the kernel C ABI: deviation after one launch (two iterations)
and launch times over chunk lengths.  usage: fmad_probe.py [N] [cols,cols,...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from prost_amd import _hip as hip
from prost_amd import synthetic

if os.environ.get("PROST_HIP_LIB"):
    hip.LIB_PATH = os.environ["PROST_HIP_LIB"]


def ulps(a, b):
    """distance in units in the last place of the larger magnitude of each pair"""
    s = np.maximum(np.abs(a), np.abs(b)).astype(np.float32)
    u = np.spacing(np.maximum(s, np.float32(1e-30)))
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / u


def main(N=4096, cols_list=(0, 12, 18, 24, 30, 36, 48), iters=200):
    hip.require_device()
    dtype = np.float32
    n, m = N * N, 2 * N * N
    rng = np.random.default_rng(0)
    fh = synthetic.rof_image(N, N, seed=42).astype(dtype)
    xh = (fh + 0.05 * (rng.random(n).astype(dtype) - 0.5)).astype(dtype)
    yh = ((rng.random(m) - 0.5) * 1.2).astype(dtype)
    f = hip.DeviceArray.from_host(fh)
    x0 = hip.DeviceArray.from_host(xh); y0 = hip.DeviceArray.from_host(yh)
    x = [hip.DeviceArray.from_host(xh), hip.DeviceArray.zeros(n, dtype)]
    y = [hip.DeviceArray.from_host(yh), hip.DeviceArray.zeros(m, dtype)]

    def desc(arith):
        d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = N, N, 1
        d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
        gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
        for i in range(7):
            d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
        d.g_coeff_ptr[1] = f.ptr.value
        d.T_val, d.S_val = 0.25, 0.5
        d.arith = arith
        return d
    I2 = hip.fn("fused_iteration2", dtype)
    L_ = hip.lib()
    tau = (C.c_double * 2)(0.3, 0.29); sigma = (C.c_double * 2)(1.0, 1.03); theta = (C.c_double * 2)(0.9, 0.91)
    out = {}
    r4 = hip.DeviceArray.zeros(4, np.float64)
    ws = hip.DeviceArray(L_.prost_hip_reduce_workspace_bytes() // 8, np.float64)
    for arith in (0, 1):
        d = desc(arith)
        print("arith asked %d -> runs %d" % (arith, L_.prost_hip_fused_iteration2_arith(C.byref(d), 0)))
        hip.check(I2(C.byref(d), x[1].ptr, y[1].ptr, x0.ptr, y0.ptr, None, None, tau, sigma, theta, 0, r4.ptr, ws.ptr, None)); hip.sync()
        out[arith] = (x[1].to_host().copy(), y[1].to_host().copy(), r4.to_host().copy())
    for k, name in ((0, "x"), (1, "y")):
        a, b = out[0][k], out[1][k]
        u = ulps(a, b)
        scale = np.abs(a).max()
        print("%s after one pair launch: max |d| = %.3e (%.2f ulp at the vector's scale %.3f), elementwise ulp: max %.1f, 99.9%% %.2f, mean %.3f, share > 2 ulp %.2e, equal %.4f"
              % (name, np.abs(a - b).max(), np.abs(a.astype(np.float64) - b).max() / np.spacing(np.float32(scale)), scale, u.max(), np.quantile(u, 0.999), u.mean(),
                 (u > 2).mean(), (a == b).mean()), flush=True)
    print("residual sums exact", out[0][2], "fmad", out[1][2], "rel", np.abs(out[0][2] - out[1][2]) / np.abs(out[0][2]))
    # 100 launches each from the same start: drift
    for arith in (0, 1):
        d = desc(arith)
        hip.check(L_.prost_hip_memcpy_d2d(x[0].ptr, x0.ptr, n * 4, None)); hip.check(L_.prost_hip_memcpy_d2d(y[0].ptr, y0.ptr, m * 4, None))
        for i in range(100):
            a, b = i % 2, (i + 1) % 2
            hip.check(I2(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, None, tau, sigma, theta, 0, None, None, None))
        hip.sync()
        out[arith] = (x[0].to_host().copy(), y[0].to_host().copy())
    for k, name in ((0, "x"), (1, "y")):
        a, b = out[0][k], out[1][k]
        print("%s after 200 iterations: rel-inf %.3e" % (name, np.abs(a.astype(np.float64) - b).max() / np.abs(a).max()), flush=True)

    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        hip.check(L_.prost_hip_event_create(C.byref(e)))

    def timed(run):
        run(10); hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters); hip.check(L_.prost_hip_event_record(ev[1], None))
        hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        return ms.value / iters
    for mode in (0, 2):
        for cols in cols_list:
            line = "mode=%d cols=%-3d" % (mode, cols)
            for arith in (0, 1):
                d = desc(arith)

                def run2(k):
                    for i in range(k):
                        a, b = i % 2, (i + 1) % 2
                        hip.check(I2(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, None, tau, sigma, theta, cols,
                                     r4.ptr if mode & 2 else None, ws.ptr if mode & 2 else None, None))
                t = min(timed(run2), timed(run2), timed(run2))
                line += "  %s %.4f ms/launch %6.0f it/s frac(7 floats) %.3f" % ("fmad " if arith else "exact", t, 2e3 / t, 7 * n * 4 / 1e9 / (t * 1e-3) / 8000)
            print(line, flush=True)


def main_k(N=4096, iters=200):
    """K iterations per launch: equality with pair launches of the same class, launch times over chunk lengths"""
    hip.require_device()
    dtype = np.float32
    n, m = N * N, 2 * N * N
    rng = np.random.default_rng(0)
    fh = synthetic.rof_image(N, N, seed=42).astype(dtype)
    xh = (fh + 0.05 * (rng.random(n).astype(dtype) - 0.5)).astype(dtype)
    yh = ((rng.random(m) - 0.5) * 1.2).astype(dtype)
    f = hip.DeviceArray.from_host(fh)
    x0 = hip.DeviceArray.from_host(xh); y0 = hip.DeviceArray.from_host(yh)
    x = [hip.DeviceArray.zeros(n, dtype), hip.DeviceArray.zeros(n, dtype)]
    y = [hip.DeviceArray.zeros(m, dtype), hip.DeviceArray.zeros(m, dtype)]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = N, N, 1
    d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
    gv = [1, 0, 10, 0, 0, 0, 0]; fv = [1, 1, 1, 0, 0, 0, 0]
    for i in range(7):
        d.g_coeff_val[i] = gv[i]; d.f_coeff_val[i] = fv[i]
    d.g_coeff_ptr[1] = f.ptr.value
    d.T_val, d.S_val = 0.25, 0.5
    d.arith = int(os.environ.get("PROST_ARITH", "1"))          # 0: the exact class (K <= 4), 1: the tolerance class
    L_ = hip.lib()
    I2 = hip.fn("fused_iteration2", dtype)
    IK = L_.prost_hip_fused_iterationk_f32
    print("K max", L_.prost_hip_fused_iterationk_max(C.byref(d), 0))
    taus = [0.3, 0.29, 0.28, 0.27, 0.26, 0.25, 0.24, 0.23, 0.22, 0.21, 0.2, 0.19]
    sigmas = [1.0, 1.03, 1.06, 1.09, 1.12, 1.15, 1.18, 1.21, 1.24, 1.27, 1.3, 1.33]
    thetas = [0.9, 0.91, 0.92, 0.93, 0.94, 0.95, 0.96, 0.97, 0.98, 0.985, 0.99, 0.995]
    r4 = hip.DeviceArray.zeros(4, np.float64)
    ws = hip.DeviceArray(L_.prost_hip_reduce_workspace_bytes() // 8, np.float64)

    def arr(v, i, k):
        return (C.c_double * k)(*v[i:i + k])

    def reset():
        hip.check(L_.prost_hip_memcpy_d2d(x[0].ptr, x0.ptr, n * 4, None)); hip.check(L_.prost_hip_memcpy_d2d(y[0].ptr, y0.ptr, m * 4, None))

    def run_seq(ks, res_last=False, cols=0):
        """from (x0, y0): launches of ks[i] iterations each; returns the iterate and the sums of the last launch"""
        reset(); it = 0
        for i, k in enumerate(ks):
            a, b = i % 2, (i + 1) % 2
            last = res_last and i == len(ks) - 1
            if k == -2:      # the pair kernel
                hip.check(I2(C.byref(d), x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, None, None, arr(taus, it, 2), arr(sigmas, it, 2), arr(thetas, it, 2), 0,
                             r4.ptr if last else None, ws.ptr if last else None, None)); it += 2
            else:
                hip.check(IK(C.byref(d), k, x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, arr(taus, it, k), arr(sigmas, it, k), arr(thetas, it, k), cols,
                             r4.ptr if last else None, ws.ptr if last else None, None)); it += k
        hip.sync()
        e = len(ks) % 2
        return x[e].to_host().copy(), y[e].to_host().copy(), r4.to_host().copy()
    ref12 = run_seq([-2] * 6, True)
    cases = (("6 x K=2", [2] * 6), ("4 x K=3", [3] * 4), ("3 x K=4", [4] * 3), ("K=4,3,3,2", [4, 3, 3, 2]), ("K=5,5,2", [5, 5, 2]), ("2 x K=6", [6, 6]),
             ("K=6 cols 5", None), ("K=5 cols 100", None), ("K=2 cols 7", None), ("K=4 cols 5", None), ("K=3 cols 1000", None))
    kmax = L_.prost_hip_fused_iterationk_max(C.byref(d), 0)
    cases = tuple(c for c in cases if (max(c[1]) if c[1] else int(c[0][2])) <= kmax)
    for name, ks in cases:
        if ks is None:
            k = int(name[2]); got = run_seq([k] * (12 // k) + ([12 % k] if 12 % k else []), True, cols=int(name.split()[-1]))
        else:
            got = run_seq(ks, True)
        print("%-14s vs 6 pair launches (12 iterations): x equal %s (max |d| %.2e), y equal %s (max |d| %.2e), sums rel %s" % (
            name, np.array_equal(got[0], ref12[0]), np.abs(got[0] - ref12[0]).max(), np.array_equal(got[1], ref12[1]), np.abs(got[1] - ref12[1]).max(),
            np.abs(got[2] - ref12[2]) / np.abs(ref12[2])), flush=True)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        hip.check(L_.prost_hip_event_create(C.byref(e)))

    def timed(run):
        run(10); hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters); hip.check(L_.prost_hip_event_record(ev[1], None))
        hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        return ms.value / iters
    reset()
    for k in [k for k in (int(c) for c in os.environ.get("PROBE_KS", "2,3,4,5,6").split(",")) if k <= kmax]:
        for mode in (0, 2):
            for cols in tuple(int(c) for c in os.environ.get("PROBE_COLS", "0,18,24,30,36,42,48,72").split(",")):
                def runk(cnt):
                    for i in range(cnt):
                        a, b = i % 2, (i + 1) % 2
                        hip.check(IK(C.byref(d), k, x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, arr(taus, 0, k), arr(sigmas, 0, k), arr(thetas, 0, k), cols,
                                     r4.ptr if mode else None, ws.ptr if mode else None, None))
                t = min(timed(runk), timed(runk), timed(runk))
                print("K=%d mode=%d cols=%-3d (auto %d): %.4f ms/launch %6.0f it/s frac(7 floats) %.3f" % (
                    k, mode, cols, L_.prost_hip_fused_iterationk_chunk_cols(C.byref(d), 0, k, mode), t, k * 1e3 / t, 7 * n * 4 / 1e9 / (t * 1e-3) / 8000), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "k":
        main_k(int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
        sys.exit(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if len(sys.argv) > 2:
        main(N, tuple(int(c) for c in sys.argv[2].split(",")))
    else:
        main(N)
