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
    report("grad2d_adj (N/2)^2 x 4 label_first", timeit(lambda: hip.check(f("grad2d_adj")(A[1].ptr, B[0].ptr, sz(N // 2), sz(N // 2), sz(4), 1, 0, None))), 3 * n * 4)
    # 29 diagonals (the band count of test_linop_diags.m's larger case): offsets around 0 and around +-N
    o29 = np.array(sorted(set(list(range(-7, 8)) + [N + k for k in range(-3, 4)] + [-N + k for k in range(-3, 4)])), dtype=np.int64)
    ofs29 = hip.DeviceArray.from_host(o29); fac29 = hip.DeviceArray.from_host(rng.random(o29.size).astype(np.float32))
    report("diags_fwd %d diagonals" % o29.size, timeit(lambda: hip.check(f("diags_fwd")(A[1].ptr, A[0].ptr, sz(n), sz(n), sz(o29.size), ofs29.ptr, fac29.ptr, None))), 3 * n * 4)
    # a CSR block with ~30 non-zeros per row (a 5 x 6 neighbourhood, random weights: no row patterns at the C ABI): rows = n / 8
    rows30 = n // 8
    base = (np.arange(rows30, dtype=np.int64) * 8)[:, None]
    offs30 = np.array([dx * N + dy for dx in range(-2, 3) for dy in range(-3, 3)], dtype=np.int64)[None, :]
    cols30 = np.clip(base + offs30, 0, n - 1)
    cols30.sort(axis=1)
    W30 = sp.csr_matrix((rng.random(cols30.size).astype(dt), cols30.reshape(-1), np.arange(0, cols30.size + 1, offs30.size)), shape=(rows30, n))
    dv30, dp30, di30 = hip.DeviceArray.from_host(W30.data.astype(dt)), hip.DeviceArray.from_host(W30.indptr.astype(np.int32)), hip.DeviceArray.from_host(W30.indices.astype(np.int32))
    report("csr_spmv_acc 30 nnz/row", timeit(lambda: hip.check(f("csr_spmv_acc")(A[1].ptr, A[0].ptr, sz(rows30), sz(W30.nnz), dv30.ptr, dp30.ptr, di30.ptr, None))),
           W30.nnz * 8 + rows30 * 4 + rows30 * 8 + n * 4)
    # Kronecker blocks of the multilabel examples: kron(S, I_d) and kron(I_d, S) with a small sparse S (12 x 16, 3 non-zeros per row)
    S = sp.random(12, 16, density=3 / 16, random_state=1, format="csr"); S.data[:] = 1 + rng.random(S.nnz)
    sv, sp_, si = hip.DeviceArray.from_host(S.data.astype(np.float32)), hip.DeviceArray.from_host(S.indptr.astype(np.int32)), hip.DeviceArray.from_host(S.indices.astype(np.int32))
    dlen = n // 16
    report("sparse_kron_id (12 x 16, d = n/16)", timeit(lambda: hip.check(f("sparse_kron_id")(A[1].ptr, A[0].ptr, sz(dlen), sz(12), sv.ptr, sp_.ptr, si.ptr, None))), (12 + 16) * dlen * 4)
    report("id_kron_sparse (12 x 16, d = n/16)", timeit(lambda: hip.check(f("id_kron_sparse")(A[1].ptr, A[0].ptr, sz(dlen), sz(12), sz(16), sv.ptr, sp_.ptr, si.ptr, None))), (12 + 16) * dlen * 4)
    # epigraph projection (dim 3: two coordinates + the height), a and c scalar, b per element
    cnt = n // 3
    report("prox_epi_quad dim 3", timeit(lambda: hip.check(f("prox_epi_quad")(A[1].ptr, A[0].ptr, sz(cnt), sz(3), None, dbl(1.0), A[2].ptr, None, dbl(0.5), None))), (3 + 3 + 2) * cnt * 4)
    ptrs = (C.c_void_p * 7)(); vals = (C.c_double * 7)(1, 0, 10, 0, 0, 0, 0); ptrs[1] = A[3].ptr.value
    report("prox_elem 1d:square, b per element",
           timeit(lambda: hip.check(f("prox_elem")(0, hip.FN_ID["square"], A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), sz(1), 0, ptrs, vals, None))), 4 * n * 4)
    p0 = (C.c_void_p * 7)(); v1 = (C.c_double * 7)(1, 0.1, 1, 0, 0, 0, 0)
    report("prox_elem 1d:abs, scalar coefficients",
           timeit(lambda: hip.check(f("prox_elem")(0, hip.FN_ID["abs"], A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), sz(1), 0, p0, v1, None))), 3 * n * 4)
    v2 = (C.c_double * 7)(1, 1, 1, 0, 0, 0, 0)
    report("prox_elem norm2:ind_leq0 dim 2 planar",
           timeit(lambda: hip.check(f("prox_elem")(1, hip.FN_ID["ind_leq0"], B[1].ptr, B[0].ptr, B[2].ptr, dbl(0.3), 0, sz(n), sz(2), 0, p0, v2, None))), (4 * n + n) * 4)
    report("moreau_prescale", timeit(lambda: hip.check(f("moreau_prescale")(A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), None))), 3 * n * 4)
    report("moreau_postscale", timeit(lambda: hip.check(f("moreau_postscale")(A[1].ptr, A[0].ptr, A[2].ptr, dbl(0.3), 0, sz(n), None))), 4 * n * 4)
    report("pdhg_primal_arg", timeit(lambda: hip.check(f("pdhg_primal_arg")(A[1].ptr, A[0].ptr, A[2].ptr, A[3].ptr, dbl(0.3), sz(n), None))), 4 * n * 4)
    report("pdhg_dual_arg (m = 2n)",
           timeit(lambda: hip.check(f("pdhg_dual_arg")(B[3].ptr, B[0].ptr, B[1].ptr, B[2].ptr, B[4].ptr, dbl(0.3), dbl(0.9), sz(2 * n), None))), 5 * 2 * n * 4)
    report("pdhg_residual_primal (m = 2n)",
           timeit(lambda: hip.check(f("pdhg_residual_primal")(out2.ptr, B[0].ptr, B[1].ptr, B[2].ptr, B[3].ptr, B[4].ptr, dbl(0.3), dbl(0.9), sz(2 * n), ws.ptr, None))),
           5 * 2 * n * 4)
    report("pdhg_residual_dual (n)",
           timeit(lambda: hip.check(f("pdhg_residual_dual")(out2.ptr, A[0].ptr, A[1].ptr, A[2].ptr, A[3].ptr, A[4].ptr, dbl(0.3), sz(n), ws.ptr, None))), 5 * n * 4)
    report("nrm2 (m = 2n)", timeit(lambda: hip.check(f("nrm2")(out2.ptr, B[0].ptr, sz(2 * n), ws.ptr, None))), 2 * n * 4)
    report("axpy (m = 2n)", timeit(lambda: hip.check(f("axpy")(B[1].ptr, B[0].ptr, dbl(0.5), sz(2 * n), None))), 3 * 2 * n * 4)
    report("admm_elem TEMP1 (n)", timeit(lambda: hip.check(f("admm_elem")(0, A[1].ptr, A[0].ptr, A[2].ptr, A[3].ptr, A[4].ptr, dbl(1.7), dbl(0), sz(n), None))), 5 * n * 4)


if __name__ == "__main__":
    main()
