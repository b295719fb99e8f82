"""K-iteration kernel: launch time per pixel over image heights (column pitch) -- HBM channel spread"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from prost_amd import _hip as hip
hip.require_device()
L_ = hip.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ev = [C.c_void_p() for _ in range(2)]
for e in ev:
    hip.check(L_.prost_hip_event_create(C.byref(e)))
def timed(run, iters=100):
    run(10); hip.sync()
    hip.check(L_.prost_hip_event_record(ev[0], None)); run(iters); hip.check(L_.prost_hip_event_record(ev[1], None))
    hip.check(L_.prost_hip_event_synchronize(ev[1]))
    ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
    return ms.value / iters
SIZES = [(4096, 4096), (4096, 4100), (4096, 4104), (4096, 4112), (4096, 4128), (4096, 4160), (4096, 4224), (4096, 3968), (4096, 3720), (4096, 4216),
         (4096, 4464), (3968, 4216), (8192, 2048), (2048, 8192)]
for (nx, ny) in SIZES:
    n, m = nx * ny, 2 * nx * ny
    rng = np.random.default_rng(0)
    f = hip.DeviceArray.from_host(rng.random(n).astype(np.float32))
    x = [hip.DeviceArray.from_host(rng.random(n).astype(np.float32)), hip.DeviceArray.zeros(n, np.float32)]
    y = [hip.DeviceArray.from_host((rng.random(m) - 0.5).astype(np.float32)), hip.DeviceArray.zeros(m, np.float32)]
    d = hip.FusedDesc(); d.is3d = 0; d.nx, d.ny, d.L = nx, ny, 1
    d.g_fn = hip.FN_ID["square"]; d.f_fn = hip.FN_ID["ind_leq0"]
    for i, (g, fv) in enumerate(zip([1, 0, 10, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0])):
        d.g_coeff_val[i] = g; d.f_coeff_val[i] = fv
    d.g_coeff_ptr[1] = f.ptr.value; d.T_val, d.S_val = 0.25, 0.5; d.arith = 1
    tk = (C.c_double * 6)(0.3, 0.29, 0.28, 0.27, 0.26, 0.25); sk = (C.c_double * 6)(1.0, 1.03, 1.06, 1.09, 1.1, 1.1); thk = (C.c_double * 6)(0.9, 0.91, 0.92, 0.93, 0.94, 0.95)
    best = None
    for cols in (24, 30, 36, 42, 48):
        def run(cnt):
            for i in range(cnt):
                a, b = i % 2, (i + 1) % 2
                hip.check(L_.prost_hip_fused_iterationk_f32(C.byref(d), K, x[b].ptr, y[b].ptr, x[a].ptr, y[a].ptr, tk, sk, thk, cols, None, None, None))
        t = min(timed(run), timed(run))
        if best is None or t < best[0]:
            best = (t, cols)
    t, cols = best
    print("K=%d nx %5d ny %5d best cols %2d: %.4f ms/launch, %.3f ps/pixel, compulsory %.0f GB/s (frac %.3f)"
          % (K, nx, ny, cols, t, t * 1e9 / n, 28 * n / t / 1e6, 28 * n / t / 1e6 / 8000), flush=True)
    del f, x, y
