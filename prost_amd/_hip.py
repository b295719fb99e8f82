"""ctypes binding of the kernel C ABI (include/prost_hip.h -> prost_amd/lib/libprost_hip.so).

Only plumbing: device buffers are raw hipMalloc pointers wrapped by DeviceArray.  There is NO
CPU fallback -- every compute entry point raises HipError when no MI355X is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprost_hip.so")

FUNCTIONS = ("zero", "abs", "square", "ind_leq0", "ind_geq0", "ind_eq0", "ind_box01",
             "max_pos0", "l0", "huber", "lq", "lq_plus_eps", "trunclin", "truncquad")
FN_ID = {n: i for i, n in enumerate(FUNCTIONS)}
OP_1D, OP_NORM2 = 0, 1


class HipError(RuntimeError):
    pass


class FusedDesc(C.Structure):
    _fields_ = [("is3d", C.c_int), ("nx", C.c_size_t), ("ny", C.c_size_t), ("L", C.c_size_t),
                ("g_fn", C.c_int), ("g_coeff_ptr", C.c_void_p * 7), ("g_coeff_val", C.c_double * 7),
                ("f_fn", C.c_int), ("f_coeff_ptr", C.c_void_p * 7), ("f_coeff_val", C.c_double * 7),
                ("T_val", C.c_double), ("S_val", C.c_double), ("res_x0", C.c_size_t), ("res_x1", C.c_size_t), ("g_b_masked", C.c_int),
                ("var_T", C.c_int), ("T_cls", C.c_double * 3), ("f_moreau", C.c_int), ("arith", C.c_int)]


class ArgSpec(C.Structure):
    """prost_hip_arg_spec: how a prox obtains its argument (PROST_ARG_PLAIN / PDHG_PRIMAL / PDHG_DUAL)"""
    _fields_ = [("mode", C.c_int), ("v", C.c_void_p * 4), ("s", C.c_double * 2),
                ("op", C.c_void_p), ("op_rows", C.c_uint64), ("op_cols", C.c_uint64), ("base", C.c_uint64), ("w", C.c_void_p * 2), ("kty_out", C.c_void_p),
                ("use", C.c_int * 2), ("res_ws", C.c_void_p), ("res_slot", C.c_uint), ("res_slots_max", C.c_uint)]      # operator sources / residual sums: zero = none


class CglsDesc(C.Structure):
    """prost_hip_cgls_desc"""
    _fields_ = [("state", C.c_void_p), ("workspace", C.c_void_p), ("b", C.c_void_p), ("x", C.c_void_p), ("p", C.c_void_p), ("q", C.c_void_p),
                ("r", C.c_void_p), ("s", C.c_void_p), ("t", C.c_void_p), ("sigma", C.c_void_p), ("tau", C.c_void_p), ("m", C.c_uint64), ("n", C.c_uint64),
                ("shift", C.c_double), ("tol", C.c_double), ("host_done", C.c_void_p), ("epoch", C.c_int)]


class OpBlock(C.Structure):
    """prost_hip_op_block"""
    _fields_ = [("kind", C.c_int), ("row", C.c_uint64), ("col", C.c_uint64), ("nrows", C.c_uint64), ("ncols", C.c_uint64),
                ("nx", C.c_uint64), ("ny", C.c_uint64), ("L", C.c_uint64),
                ("val", C.c_void_p), ("ptr", C.c_void_p), ("ind", C.c_void_p), ("val_t", C.c_void_p), ("ptr_t", C.c_void_p), ("ind_t", C.c_void_p),
                ("ids", C.c_void_p), ("pptr", C.c_void_p), ("rel", C.c_void_p), ("pval", C.c_void_p),          # ABI 7: row patterns of K ...
                ("ids_t", C.c_void_p), ("pptr_t", C.c_void_p), ("rel_t", C.c_void_p), ("pval_t", C.c_void_p),    # ... and of K^T (NULL: CSR)
                ("anchor", C.c_void_p), ("anchor_t", C.c_void_p)]                                                # ABI 8: anchored tables (NULL: offsets from the row number)


class FusedOp(C.Structure):
    """prost_hip_fused_op"""
    _fields_ = [("nblocks", C.c_int), ("block", OpBlock * 4)]


OP_CSR, OP_GRAD2D, OP_GRAD3D = 1, 2, 3


class CglsResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("indefinite", C.c_int), ("flag", C.c_int),
                ("norms", C.c_double), ("norms0", C.c_double), ("normx", C.c_double), ("xmax", C.c_double)]


_lib = None


def lib():
    """Loads libprost_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or make -C prost_amd/csrc)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.prost_hip_last_error.restype = C.c_char_p
        if L.prost_hip_abi_version() != 10:
            raise HipError("libprost_hip.so has ABI version %d, this binding needs 10: rebuild (make -C prost_amd/csrc)" % L.prost_hip_abi_version())
        L.prost_hip_reduce_workspace_bytes.restype = C.c_size_t
        L.prost_hip_cgls_state_bytes.restype = C.c_size_t
        L.prost_hip_cgls_workspace_bytes.restype = C.c_size_t
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise HipError(lib().prost_hip_last_error().decode() or ("prost_hip error %d" % rc))


def device_count():
    n = C.c_int(0)
    rc = lib().prost_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_device():
    if device_count() < 1:
        raise HipError("no HIP device available: the prost hot path only runs on an MI355X (gfx950)")


_SUFFIX = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}


def suffix(dtype):
    return _SUFFIX[np.dtype(dtype)]


def fn(name, dtype):
    return getattr(lib(), "prost_hip_%s_%s" % (name, suffix(dtype)))


class DeviceArray:
    """1-D device buffer of a numpy dtype (hipMalloc/hipFree through the C ABI)."""

    def __init__(self, n, dtype):
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.ptr = C.c_void_p()
        check(lib().prost_hip_malloc(C.byref(self.ptr), C.c_size_t(max(self.n, 1) * self.dtype.itemsize)))

    @classmethod
    def from_host(cls, a, dtype=None, stream=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = cls(a.size, a.dtype)
        check(lib().prost_hip_memcpy_h2d(d.ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), stream))
        check(lib().prost_hip_stream_synchronize(stream))
        return d

    @classmethod
    def zeros(cls, n, dtype, stream=None):
        d = cls(n, dtype)
        check(lib().prost_hip_memset(d.ptr, 0, C.c_size_t(d.n * d.dtype.itemsize), stream))
        return d

    def to_host(self, stream=None):
        out = np.empty(self.n, dtype=self.dtype)
        check(lib().prost_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(out.nbytes), stream))
        check(lib().prost_hip_stream_synchronize(stream))
        return out

    def offset(self, elems):
        return C.c_void_p(self.ptr.value + int(elems) * self.dtype.itemsize)

    def free(self):
        if self.ptr is not None and self.ptr.value:
            lib().prost_hip_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sz(v):
    return C.c_size_t(int(v))


def dbl(v):
    return C.c_double(float(v))


def coeff_args(coeffs, dtype, count):
    """7 coefficients (scalar or length-count) -> (ptr array, val array, keepalive list)."""
    ptrs = (C.c_void_p * 7)()
    vals = (C.c_double * 7)()
    keep = []
    for i, c in enumerate(coeffs):
        c = np.atleast_1d(np.asarray(c, dtype=np.float64)).ravel()
        if c.size > 1:
            d = DeviceArray.from_host(c.astype(dtype))
            keep.append(d)
            ptrs[i] = d.ptr.value
        else:
            ptrs[i] = None
            vals[i] = float(c[0])
    return ptrs, vals, keep


def sync(stream=None):
    check(lib().prost_hip_stream_synchronize(stream))
