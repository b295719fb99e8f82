"""Kronecker block kernels at the C ABI: rate of kron(S, I_d) and kron(I_d, S) for small sparse S (the multilabel examples' shapes).
usage: kron_probe.py [log2 of d]   (PROST_KRON_PLAIN=1: the one-element-per-lane kernels)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

from prost_amd import _hip as hip


def main(logd=22):
    hip.require_device()
    L_ = hip.lib()
    rng = np.random.default_rng(3)
    d = 1 << logd
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        hip.check(L_.prost_hip_event_create(C.byref(e)))

    def timeit(run, iters=30):
        for _ in range(3):
            run()
        hip.sync()
        hip.check(L_.prost_hip_event_record(ev[0], None))
        for _ in range(iters):
            run()
        hip.check(L_.prost_hip_event_record(ev[1], None)); hip.check(L_.prost_hip_event_synchronize(ev[1]))
        ms = C.c_float(); hip.check(L_.prost_hip_event_elapsed_ms(ev[0], ev[1], C.byref(ms)))
        return ms.value / iters

    sz = C.c_size_t
    for dt, name in ((np.float32, "f32"), (np.float64, "f64")):
        for (m, n, per_row) in ((12, 16, 3), (3, 3, 3), (1, 8, 8), (32, 32, 2)):
            S = sp.random(m, n, density=per_row / n, random_state=1, format="csr"); S.data[:] = 1 + rng.random(S.nnz)
            sv, sp_, si = (hip.DeviceArray.from_host(S.data.astype(np.float32)), hip.DeviceArray.from_host(S.indptr.astype(np.int32)),
                           hip.DeviceArray.from_host(S.indices.astype(np.int32)))
            x = hip.DeviceArray.from_host(rng.standard_normal(n * d).astype(dt)); r = hip.DeviceArray.zeros(m * d, dt)
            item = np.dtype(dt).itemsize
            for op in ("sparse_kron_id", "id_kron_sparse"):
                for acc in (False, True):
                    fn = getattr(L_, "prost_hip_%s%s_%s" % (op, "_acc" if acc else "", name))
                    if op == "sparse_kron_id":
                        t = timeit(lambda: hip.check(fn(r.ptr, x.ptr, sz(d), sz(m), sv.ptr, sp_.ptr, si.ptr, None)))
                    else:
                        t = timeit(lambda: hip.check(fn(r.ptr, x.ptr, sz(d), sz(m), sz(n), sv.ptr, sp_.ptr, si.ptr, None)))
                    used = np.unique(S.indices).size
                    mb = (used + m * (2 if acc else 1)) * d * item / 1e6
                    print("%s %-14s%s S %2d x %2d (%d nnz): %.4f ms, %7.1f MB compulsory, %6.0f GB/s, frac %.3f" % (
                        name, op, " acc" if acc else "    ", m, n, S.nnz, t, mb, mb / t, mb / t / 8000), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 22)
