"""ctypes wrapper of oracle/_ref/libprost_ref.so -- the REAL reference code (built by
oracle/Makefile.ref from /root/reference where it lies).  TEST INFRASTRUCTURE ONLY.

available() is False on machines without the prebuilt library and without /root/reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import FN_ID, OP_1D, OP_NORM2, STEPSIZE, _dt, _p, grad2d as _orc_grad2d, grad3d as _orc_grad3d

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_ref", "libprost_ref.so")
REFERENCE_ROOT = "/root/reference"

BLOCK_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)


class RefPDHGOpts(C.Structure):
    _fields_ = [("tau0", C.c_double), ("sigma0", C.c_double), ("residual_iter", C.c_int),
                ("scale_steps_operator", C.c_int), ("alg2_gamma", C.c_double),
                ("arg_alpha0", C.c_double), ("arg_nu", C.c_double), ("arg_delta", C.c_double),
                ("arb_delta", C.c_double), ("arb_tau", C.c_double), ("stepsize", C.c_int)]


class RefTolOpts(C.Structure):
    _fields_ = [("tol_rel_primal", C.c_double), ("tol_rel_dual", C.c_double),
                ("tol_abs_primal", C.c_double), ("tol_abs_dual", C.c_double), ("solve_dual", C.c_int)]


def build():
    """(Re)build from /root/reference if it is present; no-op otherwise."""
    if os.path.isdir(REFERENCE_ROOT):
        subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref", "-j4"], stdout=subprocess.DEVNULL)
    return os.path.exists(_LIB_PATH)


def available():
    return os.path.exists(_LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_LIB_PATH)
        L.ref_last_error.restype = C.c_char_p
        L.ref_problem_create.restype = C.c_void_p
        L.ref_problem_create.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
        L.ref_problem_destroy.argtypes = [C.c_void_p]
        L.ref_problem_destroy.restype = None
        L.ref_prox_elem.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                    C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_project_epi_quad.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.ref_csr2csc.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
        L.ref_linspace.argtypes = [C.c_double, C.c_double, C.c_int, C.c_void_p]
        L.ref_problem_add_block_cb.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                               BLOCK_CB, C.c_void_p, C.c_double, C.c_double]
        L.ref_problem_add_prox_elem.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t,
                                                C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_problem_add_prox_zero.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t]
        L.ref_problem_set_scaling.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_size_t]
        L.ref_problem_initialize.argtypes = [C.c_void_p]
        L.ref_problem_get_scaling.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_problem_normest.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_prox_elem_eval.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.ref_pdhg_run.argtypes = [C.c_void_p, C.POINTER(RefPDHGOpts), C.POINTER(RefTolOpts), C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_size_t, C.c_int] + [C.c_void_p] * 5
        L.ref_rand.restype = C.c_int
        L.ref_srand.argtypes = [C.c_uint]
        L.ref_srand.restype = None
        _lib = L
    return _lib


class RefError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise RefError(lib().ref_last_error().decode())


def _coeff_arrays(coeffs, dtype):
    ptrs = (C.c_void_p * 7)()
    vals = (C.c_double * 7)()
    keep = []
    for i, c in enumerate(coeffs):
        c = np.atleast_1d(np.asarray(c, dtype=np.float64))
        if c.size > 1:
            a = np.ascontiguousarray(c.astype(dtype))
            keep.append(a)
            ptrs[i] = a.ctypes.data
        else:
            ptrs[i] = None
            vals[i] = float(c[0])
    return ptrs, vals, keep


def prox_elem(op, fn, arg, tau_diag, tau, count, dim, interleaved, coeffs, invert_tau=False):
    arg = np.ascontiguousarray(arg)
    tau_diag = np.ascontiguousarray(tau_diag, dtype=arg.dtype)
    res = np.zeros_like(arg)
    ptrs, vals, keep = _coeff_arrays(coeffs, arg.dtype)
    fn_id = FN_ID[fn] if isinstance(fn, str) else fn
    _chk(lib().ref_prox_elem(_dt(arg.dtype), op, fn_id, _p(res), _p(arg), _p(tau_diag), float(tau),
                             int(invert_tau), count, dim, int(interleaved), ptrs, vals))
    return res


def project_epi_quad(x0, y0, alpha):
    """x0: (dim, count) planar; returns (x, y) projected with helper::ProjectEpiQuadNd."""
    x0 = np.ascontiguousarray(x0)
    dim, count = x0.shape
    buf = np.concatenate([x0.reshape(-1), np.zeros(count, dtype=x0.dtype)])
    y0 = np.ascontiguousarray(y0, dtype=x0.dtype)
    alpha = np.ascontiguousarray(alpha, dtype=x0.dtype)
    _chk(lib().ref_project_epi_quad(_dt(x0.dtype), _p(buf), count, dim, _p(y0), _p(alpha)))
    return buf[:dim * count].reshape(dim, count), buf[dim * count:]


def csr2csc(n, m, val, col_idx, row_start):
    val = np.ascontiguousarray(val).copy()
    nz = len(val)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32).copy()
    row_start = np.ascontiguousarray(row_start, dtype=np.int32).copy()
    out_val = np.zeros(nz, dtype=val.dtype)
    row_idx = np.zeros(nz, dtype=np.int32)
    col_start = np.zeros(m + 1, dtype=np.int32)
    _chk(lib().ref_csr2csc(_dt(val.dtype), n, m, nz, _p(val), _p(col_idx), _p(row_start), _p(out_val),
                           _p(row_idx), _p(col_start)))
    return out_val, row_idx, col_start


def linspace(start, end, num):
    out = np.zeros(num + 8)
    k = lib().ref_linspace(float(start), float(end), int(num), _p(out))
    return out[:k]


def _elem_desc(desc):
    """[name, idx, size, diagsteps, data] -> (op, fn, idx, count, dim, interleaved, diagsteps, coeffs, moreau_depth)"""
    depth = 0
    while desc[0] == "moreau":
        desc = desc[4][0]
        depth += 1
    name, idx, size, diagsteps, data = desc
    if not name.startswith("elem_operation:"):
        raise RefError("only elementwise prox can be built on the reference side: " + name)
    _, kind, fn = name.split(":")
    count, dim, interleaved, coeffs = data
    return (OP_1D if kind == "1d" else OP_NORM2, FN_ID[fn], idx, int(count), int(dim), bool(interleaved),
            bool(diagsteps), coeffs, depth)


def _coeff_ptrs(coeffs):
    arrs = [np.ascontiguousarray(np.atleast_1d(np.asarray(c, dtype=np.float64)).ravel()) for c in coeffs]
    ptrs = (C.c_void_p * 7)(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * 7)(*[a.size for a in arrs])
    return ptrs, lens, arrs


def eval_prox(prox_fn, arg, tau, Tau, dtype=np.float64):
    arg = np.ascontiguousarray(np.asarray(arg, dtype=np.float64).astype(dtype))
    Tau = np.ascontiguousarray(np.asarray(Tau, dtype=np.float64).astype(dtype))
    op, fn, idx, count, dim, il, ds, coeffs, depth = _elem_desc(prox_fn(0, arg.size))
    ptrs, lens, keep = _coeff_ptrs(coeffs)
    res = np.zeros_like(arg)
    _chk(lib().ref_prox_elem_eval(_dt(dtype), op, fn, count, dim, int(il), ptrs, lens, depth, _p(res), _p(arg),
                                  _p(Tau), float(tau)))
    return res.astype(np.float64)


class RefProblem:
    """Reference Problem<T> + BackendPDHG<T>, leaf blocks delegated to the oracle's stencils
    through the reference's Block plugin interface (gradient kernels need a GPU launch)."""

    def __init__(self, data, nrows, ncols, dtype=np.float64):
        L = lib()
        self.dtype = np.dtype(dtype)
        self.nrows, self.ncols = int(nrows), int(ncols)
        self.h = L.ref_problem_create(_dt(dtype), self.nrows, self.ncols)
        self._keep = []
        for blk in data.get("linop", []):
            name, row, col, bd = blk
            if name not in ("gradient2d", "gradient3d"):
                raise RefError("reference-side leaf block not available: " + name)
            nx, ny, Lc, lf = bd
            n = nx * ny * Lc
            k = 3 if name == "gradient3d" else 2
            fn = _orc_grad3d if k == 3 else _orc_grad2d

            def cb(user, adjoint, res, rhs, fn=fn, n=n, k=k, nx=nx, ny=ny, Lc=Lc, lf=lf):
                nres, nrhs = (n, k * n) if adjoint else (k * n, n)
                r = np.ctypeslib.as_array(C.cast(res, C.POINTER(np.ctypeslib.as_ctypes_type(self.dtype))), (nres,))
                x = np.ctypeslib.as_array(C.cast(rhs, C.POINTER(np.ctypeslib.as_ctypes_type(self.dtype))), (nrhs,))
                fn(x, nx, ny, Lc, lf, adjoint=bool(adjoint), acc=r)
            cbo = BLOCK_CB(cb)
            self._keep.append(cbo)
            # row_sum = 2, col_sum = 4 / 6: block_gradient2d.cu:154-163, block_gradient3d.cu:165-174
            _chk(L.ref_problem_add_block_cb(self.h, row, col, k * n, n, cbo, None, 2.0, 4.0 if k == 2 else 6.0))
        for key, which in (("prox_g", 0), ("prox_f", 1), ("prox_gstar", 2), ("prox_fstar", 3)):
            for p in data.get(key, []):
                if p[0] == "zero":
                    _chk(L.ref_problem_add_prox_zero(self.h, which, p[1], p[2]))
                    continue
                op, fn, idx, count, dim, il, ds, coeffs, depth = _elem_desc(p)
                ptrs, lens, keep = _coeff_ptrs(coeffs)
                _chk(L.ref_problem_add_prox_elem(self.h, which, op, fn, idx, count, dim, int(il), int(ds), ptrs,
                                                 lens, depth))
        sc = data.get("scaling", "alpha")
        if sc == "alpha":
            _chk(L.ref_problem_set_scaling(self.h, 0, float(data.get("scaling_alpha", 1)), None, 0, None, 0))
        elif sc == "identity":
            _chk(L.ref_problem_set_scaling(self.h, 1, 0.0, None, 0, None, 0))
        else:
            l = np.ascontiguousarray(data["scaling_left"], dtype=np.float64)
            r = np.ascontiguousarray(data["scaling_right"], dtype=np.float64)
            _chk(L.ref_problem_set_scaling(self.h, 2, 0.0, _p(l), l.size, _p(r), r.size))

    def initialize(self):
        _chk(lib().ref_problem_initialize(self.h))

    def scaling(self):
        l, r = np.zeros(self.nrows), np.zeros(self.ncols)
        _chk(lib().ref_problem_get_scaling(self.h, _p(l), _p(r)))
        return l, r

    def normest(self):
        lib().ref_srand(1)       # fresh-process state of std::rand (problem.cu:435)
        out = C.c_double()
        _chk(lib().ref_problem_normest(self.h, C.byref(out)))
        return out.value

    def pdhg(self, backend_opts, opts, iters):
        """Runs `iters` BackendPDHG::PerformIteration calls from a fresh backend; returns state dict."""
        bo = backend_opts
        po = RefPDHGOpts(float(bo["tau0"]), float(bo["sigma0"]), int(bo["residual_iter"]),
                         int(bool(bo["scale_steps_operator"])), float(bo["alg2_gamma"]), float(bo["arg_alpha0"]),
                         float(bo["arg_nu"]), float(bo["arg_delta"]), float(bo["arb_delta"]), float(bo["arb_tau"]),
                         STEPSIZE[bo["stepsize"]])
        to = RefTolOpts(float(opts["tol_rel_primal"]), float(opts["tol_rel_dual"]), float(opts["tol_abs_primal"]),
                        float(opts["tol_abs_dual"]), int(bool(opts.get("solve_dual", False))))
        x0 = opts.get("x0")
        y0 = opts.get("y0")
        x0 = None if x0 is None else np.ascontiguousarray(x0, dtype=np.float64)
        y0 = None if y0 is None else np.ascontiguousarray(y0, dtype=np.float64)
        n, m = self.ncols, self.nrows
        x, z, y, w, sc = np.zeros(n), np.zeros(m), np.zeros(m), np.zeros(n), np.zeros(6)
        lib().ref_srand(1)
        _chk(lib().ref_pdhg_run(self.h, C.byref(po), C.byref(to), _p(x0), 0 if x0 is None else x0.size,
                                _p(y0), 0 if y0 is None else y0.size, int(iters), _p(x), _p(z), _p(y), _p(w), _p(sc)))
        keys = ("primal_res", "dual_res", "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual")
        out = dict(x=x, z=z, y=y, w=w)
        out.update(dict(zip(keys, sc)))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_problem_destroy(self.h)
            self.h = None
