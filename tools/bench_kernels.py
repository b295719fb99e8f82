#!/usr/bin/env python3
"""Per-kernel achieved HBM bandwidth of the generic C-ABI kernels (include/prost_hip.h) against
their COMPULSORY bytes (each operand streamed once).  One JSON line per kernel."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from prost_amd import _hip as hip

L_ = None
EV = None


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    hip.sync()
    hip.check(L_.prost_hip_event_record(EV[0], None))
    for _ in range(reps):
        fn()
    hip.check(L_.prost_hip_event_record(EV[1], None))
    hip.check(L_.prost_hip_event_synchronize(EV[1]))
    ms = C.c_float()
    hip.check(L_.prost_hip_event_elapsed_ms(EV[0], EV[1], C.byref(ms)))
    return ms.value / reps


def report(name, ms, nbytes):
    print(json.dumps({"kernel": name, "ms": round(ms, 5), "compulsory_MB": round(nbytes / 1e6, 1),
                      "achieved_GBps": round(nbytes / 1e9 / (ms * 1e-3), 1), "frac_of_8TBps": round(nbytes / 1e9 / (ms * 1e-3) / 8000, 3)}), flush=True)


def main():
    global L_, EV
    hip.require_device()
    L_ = hip.lib()
    EV = [C.c_void_p(), C.c_void_p()]
    for e in EV:
        hip.check(L_.prost_hip_event_create(C.byref(e)))
    dt = np.float32
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192     # 8192^2 floats = 268 MB per operand: past the 256 MB Infinity Cache
    n = N * N
    rng = np.random.default_rng(0)
    h = rng.random(2 * n).astype(dt)
    A = [hip.DeviceArray.from_host(h[:n]) for _ in range(6)]          # distinct buffers: no operand is aliased below
    B = [hip.DeviceArray.from_host(h) for _ in range(6)]
    ws = hip.DeviceArray(L_.prost_hip_reduce_workspace_bytes() // 8, np.float64)
    out2 = hip.DeviceArray.zeros(4, np.float64)
    sz, dbl = hip.sz, hip.dbl
    f = lambda name: hip.fn(name, dt)
    report("grad2d_fwd N^2 (overwrite)", timeit(lambda: hip.check(f("grad2d_fwd")(B[0].ptr, A[0].ptr, sz(N), sz(N), sz(1), 0, 0, None))), 3 * n * 4)
    report("grad2d_fwd N^2 (accumulate)", timeit(lambda: hip.check(f("grad2d_fwd")(B[0].ptr, A[0].ptr, sz(N), sz(N), sz(1), 0, 1, None))), 5 * n * 4)
    report("grad2d_adj N^2 (overwrite)", timeit(lambda: hip.check(f("grad2d_adj")(A[1].ptr, B[0].ptr, sz(N), sz(N), sz(1), 0, 0, None))), 3 * n * 4)
    report("grad2d_fwd (N/2)^2 x 4 label_first", timeit(lambda: hip.check(f("grad2d_fwd")(B[0].ptr, A[0].ptr, sz(N // 2), sz(N // 2), sz(4), 1, 0, None))), 3 * n * 4)
    # 3-D on 1024 x 1024 x 16 = n voxels; output needs 3n
    B3 = hip.DeviceArray.zeros(3 * n, dt)
    report("grad3d_fwd (N/4)x(N/4)x16 (overwrite)", timeit(lambda: hip.check(f("grad3d_fwd")(B3.ptr, A[0].ptr, sz(N // 4), sz(N // 4), sz(16), 0, 0, None))), 4 * n * 4)
    report("grad3d_adj (N/4)x(N/4)x16 (overwrite)", timeit(lambda: hip.check(f("grad3d_adj")(A[1].ptr, B3.ptr, sz(N // 4), sz(N // 4), sz(16), 0, 0, None))), 4 * n * 4)
    ofs = hip.DeviceArray.from_host(np.array([-N, -1, 0, 1, N], dtype=np.int64)); fac = hip.DeviceArray.from_host(np.array([1, 1, -4, 1, 1], dtype=np.float32))
    report("diags_fwd 5 diagonals", timeit(lambda: hip.check(f("diags_fwd")(A[1].ptr, A[0].ptr, sz(n), sz(n), sz(5), ofs.ptr, fac.ptr, None))), 3 * n * 4)
    W = sp.hstack([sp.diags(rng.random(n // 2)), sp.diags(rng.random(n // 2))]).tocsr()
    dv, dp, di = hip.DeviceArray.from_host(W.data.astype(dt)), hip.DeviceArray.from_host(W.indptr.astype(np.int32)), hip.DeviceArray.from_host(W.indices.astype(np.int32))
    rows = n // 2
    report("csr_spmv_acc 2 nnz/row", timeit(lambda: hip.check(f("csr_spmv_acc")(A[1].ptr, A[0].ptr, sz(rows), sz(W.nnz), dv.ptr, dp.ptr, di.ptr, None))),
           W.nnz * 8 + rows * 4 + rows * 8 + n * 4)
    ptrs = (C.c_void_p * 7)(); vals = (C.c_double * 7)(1, 0, 10, 0, 0, 0, 0); ptrs[1] = A[3].ptr.value
    report("prox_elem 1d:square, b per element", timeit(lambda: hip.check(f("prox_elem")(0, hip.FN_ID["square"], A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), sz(1), 0, ptrs, vals, None))), 4 * n * 4)
    p0 = (C.c_void_p * 7)(); v1 = (C.c_double * 7)(1, 0.1, 1, 0, 0, 0, 0)
    report("prox_elem 1d:abs, scalar coefficients", timeit(lambda: hip.check(f("prox_elem")(0, hip.FN_ID["abs"], A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), sz(1), 0, p0, v1, None))), 3 * n * 4)
    v2 = (C.c_double * 7)(1, 1, 1, 0, 0, 0, 0)
    report("prox_elem norm2:ind_leq0 dim 2 planar", timeit(lambda: hip.check(f("prox_elem")(1, hip.FN_ID["ind_leq0"], B[1].ptr, B[0].ptr, B[2].ptr, dbl(0.3), 0, sz(n), sz(2), 0, p0, v2, None))), (4 * n + n) * 4)
    report("moreau_prescale", timeit(lambda: hip.check(f("moreau_prescale")(A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), None))), 3 * n * 4)
    report("moreau_postscale", timeit(lambda: hip.check(f("moreau_postscale")(A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), None))), 4 * n * 4)
    report("pdhg_primal_arg", timeit(lambda: hip.check(f("pdhg_primal_arg")(A[1].ptr, A[0].ptr, A[2].ptr, A[3].ptr, dbl(0.3), sz(n), None))), 4 * n * 4)
    report("pdhg_dual_arg (m = 2n)", timeit(lambda: hip.check(f("pdhg_dual_arg")(B[3].ptr, B[0].ptr, B[1].ptr, B[2].ptr, B[4].ptr, dbl(0.3), dbl(0.9), sz(2 * n), None))), 5 * 2 * n * 4)
    report("pdhg_residual_primal (m = 2n)", timeit(lambda: hip.check(f("pdhg_residual_primal")(out2.ptr, B[0].ptr, B[1].ptr, B[2].ptr, B[3].ptr, B[4].ptr, dbl(0.3), dbl(0.9), sz(2 * n), ws.ptr, None))), 5 * 2 * n * 4)
    report("nrm2 (m = 2n)", timeit(lambda: hip.check(f("nrm2")(out2.ptr, B[0].ptr, sz(2 * n), ws.ptr, None))), 2 * n * 4)
    report("axpy (m = 2n)", timeit(lambda: hip.check(f("axpy")(B[1].ptr, B[0].ptr, dbl(0.5), sz(2 * n), None))), 3 * 2 * n * 4)
    report("admm_elem TEMP1 (n)", timeit(lambda: hip.check(f("admm_elem")(0, A[1].ptr, A[0].ptr, A[2].ptr, A[3].ptr, A[4].ptr, dbl(1.7), dbl(0), sz(n), None))), 5 * n * 4)


if __name__ == "__main__":
    main()
