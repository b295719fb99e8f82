"""ctypes binding of the command-level C ABI (include/prost_c.h -> prost_amd/lib/libprost.so).

Marshals the nested problem descriptions (Python lists / dicts / numpy / scipy.sparse -- the
MATLAB cells / structs / matrices) into prost_value trees and calls prost_command with the same
command names and argument order as the reference's MEX gateway (prost.cpp:305-313).
No compute happens in Python and there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libprost.so")

INTERM_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_double), C.c_size_t)
STOP_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLREDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)
OUTPUT_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_char), C.c_size_t)
P2P_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))

(VALUE_EMPTY, VALUE_MATRIX, VALUE_STRING, VALUE_CELL, VALUE_STRUCT, VALUE_SPARSE, VALUE_CALLBACK) = range(7)


class ProstError(RuntimeError):
    """what mexErrMsgTxt would have shown (prost.cpp:342-346)"""


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ProstError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(or make -C prost_amd/csrc)" % LIB_PATH)
        # libprost.so finds libprost_hip.so through its $ORIGIN rpath
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.prost_value_scalar.restype = vp
        L.prost_value_scalar.argtypes = [C.c_double]
        L.prost_value_matrix.restype = vp
        L.prost_value_matrix.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.prost_value_string.restype = vp
        L.prost_value_string.argtypes = [C.c_char_p]
        L.prost_value_cell.restype = vp
        L.prost_value_cell.argtypes = [C.c_size_t]
        L.prost_value_cell_set.argtypes = [vp, C.c_size_t, vp]
        L.prost_value_struct.restype = vp
        L.prost_value_struct_set.argtypes = [vp, C.c_char_p, vp]
        L.prost_value_sparse.restype = vp
        L.prost_value_sparse.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp]
        L.prost_value_callback.restype = vp
        L.prost_value_callback.argtypes = [INTERM_CB, vp]
        L.prost_value_free.argtypes = [vp]
        L.prost_value_free.restype = None
        L.prost_value_kind.argtypes = [vp]
        L.prost_value_rows.argtypes = [vp]
        L.prost_value_rows.restype = C.c_size_t
        L.prost_value_cols.argtypes = [vp]
        L.prost_value_cols.restype = C.c_size_t
        L.prost_value_data.argtypes = [vp]
        L.prost_value_data.restype = C.POINTER(C.c_double)
        L.prost_value_str.argtypes = [vp]
        L.prost_value_str.restype = C.c_char_p
        L.prost_value_count.argtypes = [vp]
        L.prost_value_count.restype = C.c_size_t
        L.prost_value_cell_get.argtypes = [vp, C.c_size_t]
        L.prost_value_cell_get.restype = vp
        L.prost_value_field.argtypes = [vp, C.c_char_p]
        L.prost_value_field.restype = vp
        L.prost_command.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.c_int, C.POINTER(vp)]
        L.prost_last_error.restype = C.c_char_p
        L.prost_set_stop_callback.argtypes = [STOP_CB, vp]
        L.prost_set_stop_callback.restype = None
        L.prost_value_field_count.argtypes = [vp]
        L.prost_value_field_count.restype = C.c_size_t
        L.prost_value_field_name.argtypes = [vp, C.c_size_t]
        L.prost_value_field_name.restype = C.c_char_p
        L.prost_set_output_callback.argtypes = [OUTPUT_CB, vp]
        L.prost_set_output_callback.restype = None
        L.prost_comm_init_host.argtypes = [ALLREDUCE_CB, vp, C.c_int]
        L.prost_comm_set_host_p2p.argtypes = [P2P_CB, vp]
        _lib = L
    return _lib


# ------------------------------------------------------------------------------------------
# Python <-> prost_value
# ------------------------------------------------------------------------------------------
def to_value(obj, keep):
    """MATLAB-like conversion: numbers/bools -> 1x1, 1-D arrays -> Nx1, None/[] -> empty matrix,
    str -> string, list/tuple -> cell, dict -> struct, scipy sparse -> sparse, callable -> handle."""
    L = lib()
    if obj is None:
        return L.prost_value_matrix(None, 0, 0)
    if isinstance(obj, (bool, np.bool_)):
        return L.prost_value_scalar(1.0 if obj else 0.0)
    if isinstance(obj, (int, float, np.integer, np.floating)):
        return L.prost_value_scalar(float(obj))
    if isinstance(obj, str):
        return L.prost_value_string(obj.encode())
    if isinstance(obj, dict):
        s = L.prost_value_struct()
        for k, v in obj.items():
            L.prost_value_struct_set(s, k.encode(), to_value(v, keep))
        return s
    if isinstance(obj, (list, tuple)):
        c = L.prost_value_cell(len(obj))
        for i, v in enumerate(obj):
            L.prost_value_cell_set(c, i, to_value(v, keep))
        return c
    if callable(obj):
        def _cb(user, it, x, nx, y, ny, fn=obj):
            r = fn(it, np.ctypeslib.as_array(x, (nx,)).copy(), np.ctypeslib.as_array(y, (ny,)).copy())
            return int(bool(r))
        cb = INTERM_CB(_cb)
        keep.append(cb)
        return L.prost_value_callback(cb, None)
    try:
        import scipy.sparse as sp
        if sp.issparse(obj):
            K = sp.csc_matrix(obj, dtype=np.float64)
            K.sort_indices()
            val = np.ascontiguousarray(K.data, dtype=np.float64)
            ir = np.ascontiguousarray(K.indices, dtype=np.int64)
            jc = np.ascontiguousarray(K.indptr, dtype=np.int64)
            return L.prost_value_sparse(K.shape[0], K.shape[1], K.nnz, val.ctypes.data, ir.ctypes.data, jc.ctypes.data)
    except ImportError:
        pass
    a = np.asarray(obj, dtype=np.float64)
    if a.ndim == 0:
        return L.prost_value_scalar(float(a))
    if a.ndim == 1:
        a = np.ascontiguousarray(a)
        return L.prost_value_matrix(a.ctypes.data, a.size, 1 if a.size else 0)
    if a.ndim == 2:
        f = np.asfortranarray(a)
        return L.prost_value_matrix(f.ctypes.data, f.shape[0], f.shape[1])
    raise ProstError("Cannot handle arrays with dim > 2.")


_VIEW_THRESHOLD = 1 << 20     # elements; larger result matrices are handed out without a copy


class _ValueOwner:
    """Frees a prost_value tree when the last numpy view into it is gone."""

    def __init__(self, ptr):
        self.ptr = ptr
        self.used = False

    def __del__(self):
        try:
            if self.ptr:
                lib().prost_value_free(self.ptr)
        except Exception:          # interpreter shutdown
            pass
        self.ptr = None


def from_value(v, owner=None):
    """Converts a prost_value to Python.  With `owner` (a _ValueOwner of the tree's root) large matrices are returned
    as numpy views of the value's storage -- the 10^7..10^8-element result vectors are not copied a second time;
    the tree is freed when the last view is collected."""
    L = lib()
    kind = L.prost_value_kind(v)
    if kind == VALUE_MATRIX:
        r, c = L.prost_value_rows(v), L.prost_value_cols(v)
        if r * c == 0:
            return np.zeros((r, c))
        if owner is not None and r * c >= _VIEW_THRESHOLD:
            addr = C.cast(L.prost_value_data(v), C.c_void_p).value
            buf = (C.c_double * (r * c)).from_address(addr)
            buf._prost_owner = owner          # numpy keeps `buf` (the buffer exporter) alive for every view
            owner.used = True
            a = np.frombuffer(buf, dtype=np.float64)
            return a if c == 1 else a.reshape((c, r)).T
        a = np.ctypeslib.as_array(L.prost_value_data(v), (r * c,)).copy()
        if r == 1 and c == 1:
            return float(a[0])
        return a if c == 1 else a.reshape((c, r)).T
    if kind == VALUE_STRING:
        return L.prost_value_str(v).decode()
    if kind == VALUE_CELL:
        return [from_value(L.prost_value_cell_get(v, i), owner) for i in range(L.prost_value_count(v))]
    if kind == VALUE_STRUCT:
        names = [L.prost_value_field_name(v, i).decode() for i in range(L.prost_value_field_count(v))]
        return _struct_fields(v, names, owner)
    return None


def _struct_fields(v, names, owner=None):
    L = lib()
    out = {}
    for n in names:
        f = L.prost_value_field(v, n.encode())
        if f:
            out[n] = from_value(f, owner)
    return out


# first exception raised inside a host-transport callback (they run on a HIP runtime thread, in stream order: ctypes can only
# print an exception there).  The callback poisons what it was asked to produce (NaN sums / untouched halo bytes would otherwise
# flow on silently) and the NEXT command raises it.
_callback_error = []


_TEARDOWN_COMMANDS = frozenset(["release", "comm_destroy", "solver_destroy"])


def _raise_callback_error():
    if _callback_error:
        e = _callback_error[0]
        del _callback_error[:]
        raise ProstError("host-transport callback failed: %s: %s" % (type(e).__name__, e)) from e


def command(cmd, args=(), nlhs=0, struct_fields=None):
    """prost_(cmd, args...) -- returns a list of nlhs converted results."""
    L = lib()
    # A host-transport callback failure left over from an earlier command aborts the next command that would CONSUME results.  Teardown
    # commands run first and raise the stored error afterwards: clean-up after a dead peer or a gloo timeout must not leak the native
    # solver / communicator (nor need to be issued twice).
    teardown = cmd in _TEARDOWN_COMMANDS
    if not teardown:
        _raise_callback_error()
    keep = []
    vals = [to_value(a, keep) for a in args]
    prhs = (C.c_void_p * max(len(vals), 1))(*vals)
    plhs = (C.c_void_p * max(nlhs, 1))()
    try:
        rc = L.prost_command(cmd.encode(), nlhs, plhs, len(vals), prhs)
        if rc != 0:
            del _callback_error[:]
            raise ProstError(L.prost_last_error().decode())
        _raise_callback_error()            # (a callback that failed DURING this command: its results are not to be trusted)
        out = []
        for i in range(nlhs):
            if not plhs[i]:
                out.append(None)
                continue
            owner = _ValueOwner(plhs[i])
            try:
                if L.prost_value_kind(plhs[i]) == VALUE_STRUCT and struct_fields is not None:
                    out.append(_struct_fields(plhs[i], struct_fields, owner))
                else:                         # structs: every field, enumerated through prost_value_field_name
                    out.append(from_value(plhs[i], owner))
            finally:
                if owner.used:
                    plhs[i] = None            # the views own the tree now
                else:
                    owner.ptr = None          # nothing references it: freed below
        return out
    finally:
        for v in vals:
            L.prost_value_free(v)
        for i in range(nlhs):
            if plhs[i]:
                L.prost_value_free(plhs[i])


# ------------------------------------------------------------------------------------------
# the +prost command surface (matlab/+prost/*.m)
# ------------------------------------------------------------------------------------------
_RESULT_FIELDS = ("x", "y", "z", "w", "result", "iters", "path", "pair_launches")
_STATE_FIELDS = ("x", "y", "z", "w", "tau", "sigma", "theta", "rho", "iteration", "primal_res", "dual_res",
                 "primal_var_norm", "dual_var_norm", "eps_primal", "eps_dual", "cg_iterations", "path", "pair_launches",
                 "speculative_launches", "speculative_adopted", "device_rule_batches", "arithmetic", "iterations_per_launch_max", "operator_in_prox_kernels",
                 "residual_sums_in_prox_launches", "sparse_pattern_products")


def _opts_struct(opts):
    o = dict(opts)
    for k in ("x0", "y0"):
        if o.get(k) is None:
            o[k] = None
    if o.get("interm_cb") is None:
        o.pop("interm_cb", None)
    return o


def init():
    command("init")


def release():
    command("release")


def set_gpu(gpu_id):
    command("set_gpu", [gpu_id])


def list_gpus():
    return int(command("list_gpus", nlhs=1)[0])


def set_precision(name):
    """'double' (the reference's shipped default, config.hpp:7) or 'single'"""
    command("set_precision", [name])


def get_precision():
    return command("get_precision", nlhs=1)[0]


def solve(prob, backend, opts):
    """prost.solve (solve.m:5-9)"""
    prob.finalize()
    result = command("solve_problem", [prob.data, prob.nrows, prob.ncols, backend, _opts_struct(opts)],
                     nlhs=1, struct_fields=_RESULT_FIELDS)[0]
    for k in ("x", "y", "z", "w"):                       # (a one-element vector comes back as a scalar)
        if k in result:
            result[k] = np.atleast_1d(result[k])
    prob.fill_variables(result)
    return result


def eval_linop(linop, rhs, transpose):
    """prost.eval_linop (eval_linop.m:3) -> (result, rowsum, colsum, time_ms)"""
    rhs = np.asarray(rhs, dtype=np.float64).reshape(-1, 1)
    return tuple(command("eval_linop", [list(linop), rhs, bool(transpose)], nlhs=4))


def eval_prox(prox, arg, tau, Tau, verbose=False):
    """prost.eval_prox (eval_prox.m:7): prox is a function builder, evaluated at (0, len(arg))"""
    arg = np.asarray(arg, dtype=np.float64).reshape(-1, 1)
    Tau = np.asarray(Tau, dtype=np.float64).reshape(-1, 1)
    res, ms = command("eval_prox", [prox(0, arg.shape[0]), arg, float(tau), Tau, bool(verbose)], nlhs=2)
    return np.atleast_1d(res), ms


def problem_info(prob):
    """host-side problem setup (coverage checks, zero-prox filling, preconditioners); no GPU."""
    prob.finalize()
    return command("problem_info", [prob.data, prob.nrows, prob.ncols], nlhs=1,
                   struct_fields=("scaling_left", "scaling_right", "nrows", "ncols", "linop_nrows", "linop_ncols",
                                  "prox_g", "prox_f", "prox_gstar", "prox_fstar"))[0]


def load_plugin(path):
    """loads a shared library of user-defined blocks / proxes / backends (custom.cpp:11-28); they register themselves"""
    command("load_plugin", [os.path.abspath(path)])


def registered():
    """{'prox': [...], 'block': [...], 'backend': [...]}: the names in the factory registries"""
    return command("registered", nlhs=1)[0]


def set_quirks(**kw):
    command("set_quirks", [kw])


class Solver:
    """Persistent solver handle: iterate K times without convergence tests (benchmark / tests)."""

    def __init__(self, prob, backend, opts, owned_columns=None):
        """owned_columns = (x0, x1, nx): column-sharded images, see prost_amd.distributed.ColumnShardedSolver"""
        prob.finalize()
        self.prob = prob
        args = [prob.data, prob.nrows, prob.ncols, backend, _opts_struct(opts)]
        if owned_columns is not None:
            args.append(np.asarray(owned_columns, dtype=np.float64).reshape(1, 3))
        self.handle = command("solver_create", args, nlhs=1)[0]

    def halo_exchange(self, ny, halo, left_halo, right_halo, left_rank, right_rank):
        command("solver_halo_exchange", [self.handle, int(ny), int(halo), int(left_halo), int(right_halo), int(left_rank), int(right_rank)])

    def iterate_sharded(self, iters, ny, halo, left_halo, right_halo, left_rank, right_rank, since_exchange):
        """`iters` iterations with the halo refresh every halo - 2 iterations, looped natively -> new since_exchange"""
        return int(command("solver_iterate_sharded", [self.handle, int(iters), int(ny), int(halo), int(left_halo), int(right_halo), int(left_rank),
                                                      int(right_rank), int(since_exchange)], nlhs=1)[0])

    def copy_columns_from(self, dst_col, src, src_col, ncols, ny):
        command("solver_copy_columns", [self.handle, int(dst_col), src.handle, int(src_col), int(ncols), int(ny)])

    def compare(self, other):
        """comparison with another solver's iterates ON THE DEVICE (no read-back):
        {"x" | "y" | "x_prev" | "y_prev": (elements that differ in value, sum |a - b|)}"""
        m = np.asarray(command("solver_compare", [self.handle, other.handle], nlhs=1)[0]).reshape(4, 2)
        return {k: (int(m[i, 0]), float(m[i, 1])) for i, k in enumerate(("x", "y", "x_prev", "y_prev"))}

    def read(self, which, offsets, count):
        """`count` consecutive entries of the device vector `which` ("x", "y", "x_prev", "y_prev") from each offset
        -> array (len(offsets), count); partial read-back for states too large to fetch whole"""
        offsets = np.asarray(offsets, dtype=np.float64).reshape(-1)
        out = np.asarray(command("solver_read", [self.handle, which, offsets, int(count)], nlhs=1)[0], dtype=np.float64)
        return out.T if out.ndim == 2 else out.reshape(1, -1)      # (count, segments) -> (segments, count)

    @staticmethod
    def _kernel_dict(cells):
        return {k[0]: {"avg_ms": k[1], "sampled": int(k[2]), "iterations_per_launch": int(k[3]), "launches": int(k[4]),
                       "chunk_cols": int(k[5])} for k in cells or []}

    def iterate(self, iters, time_kernels=False, sample_every=8, checked=False, defer_times=False):
        """-> {"ms": wall time, "converged", "kernels": {kernel name: {"avg_ms", "sampled", "launches",
        "iterations_per_launch", "chunk_cols"}}} (kernel launch times from HIP events on the solver's stream; one launch
        in `sample_every` is bracketed).  checked=True runs the loop of prost.solve -- the stopping test after every
        observable iteration -- without callbacks or read-out, and stops when the test fires.  defer_times=True leaves
        the recorded events to a later kernel_times() call (keeps their evaluation out of a caller's timed region)."""
        info = command("solver_iterate", [self.handle, int(iters), bool(time_kernels), int(sample_every), bool(checked), bool(defer_times)],
                       nlhs=1, struct_fields=("ms", "converged", "kernels"))[0]
        info["kernels"] = self._kernel_dict(info.get("kernels"))
        return info

    def kernel_times(self):
        """evaluates the event pairs recorded by iterate(..., time_kernels=True, defer_times=True)"""
        return self._kernel_dict(command("solver_kernel_times", [self.handle], nlhs=1)[0])

    def state(self, vectors=True):
        """vectors=False: step sizes, iteration count and residuals only (no read-back of x, y, z, w)"""
        st = command("solver_state", [self.handle, bool(vectors)], nlhs=1, struct_fields=_STATE_FIELDS)[0]
        for k in ("x", "y", "z", "w"):
            if k in st:
                st[k] = np.atleast_1d(st[k])
        return st

    def destroy(self):
        if self.handle is not None:
            command("solver_destroy", [self.handle])
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def comm_unique_id():
    return np.asarray(command("comm_unique_id", nlhs=1)[0]).reshape(-1)


def comm_init(unique_id, rank, world):
    command("comm_init", [np.asarray(unique_id, dtype=np.float64).reshape(1, -1), rank, world])


_host_allreduce_keep = []


def comm_init_host(allreduce, world, p2p=None):
    """Communicator with a host-side all-reduce: `allreduce(a)` sums the float64 numpy array `a` (a view of pinned host
    memory) over the ranks IN PLACE, e.g. `dist.all_reduce(torch.from_numpy(a))` on a gloo group.  It is called from a
    runtime thread, in stream order, wherever an RCCL all-reduce would run.  For several ranks on one GPU.

    p2p (optional): `p2p(ops)` with ops = [(is_send, peer, uint8 array view of pinned host memory), ...] performs ALL
    transfers of one halo exchange and returns when they are complete (gloo: isend / irecv on every entry, then wait) --
    what ncclSend / ncclRecv inside one group do on the RCCL transport (solver_halo_exchange, solver_iterate_sharded)."""
    def _cb(user, ptr, count, fn=allreduce):
        a = np.ctypeslib.as_array(ptr, (count,))
        try:
            fn(a)
        except BaseException as e:          # noqa: BLE001 -- a peer that died, a gloo timeout: must not vanish on the runtime thread
            a[:] = np.nan                   # the global sums are not global: residuals (and every step size derived from them) become NaN
            _callback_error.append(e)
    cb = ALLREDUCE_CB(_cb)
    _host_allreduce_keep.append(cb)
    if lib().prost_comm_init_host(cb, None, int(world)) != 0:
        raise ProstError(lib().prost_last_error().decode())
    if p2p is not None:
        def _p2p(user, nops, is_send, peers, bufs, nbytes, fn=p2p):
            ops = [(bool(is_send[i]), int(peers[i]), np.ctypeslib.as_array(C.cast(bufs[i], C.POINTER(C.c_uint8)), (nbytes[i],)))
                   for i in range(nops)]
            try:
                fn(ops)
            except BaseException as e:      # noqa: BLE001
                for snd, _, buf in ops:     # halo columns that never arrived: NaN bit patterns instead of stale staging bytes
                    if not snd:
                        buf[:] = 0xFF
                _callback_error.append(e)
        pcb = P2P_CB(_p2p)
        _host_allreduce_keep.append(pcb)
        if lib().prost_comm_set_host_p2p(pcb, None) != 0:
            raise ProstError(lib().prost_last_error().decode())


def gloo_p2p(dist):
    """the p2p function of comm_init_host over a torch.distributed (gloo) process group"""
    import torch

    def p2p(ops):
        reqs = [dist.isend(torch.from_numpy(a), peer) if send else dist.irecv(torch.from_numpy(a), peer) for send, peer, a in ops]
        for r in reqs:
            r.wait()
    return p2p


def comm_info():
    """{'nranks': ranks as the communicator counts them (ncclCommCount), 'transport': 'rccl' | 'host' | 'none'}"""
    return command("comm_info", nlhs=1)[0]


_stop_cb_keep = []


def set_stop_callback(fn):
    """Registers `fn() -> bool` (True = stop) as the stopping callback of prost.solve -- the MEX gateway's Ctrl-C poll
    (prost.cpp:58-66).  It is asked once per kernel launch, i.e. after every iteration or every second one; None removes it."""
    if fn is None:
        lib().prost_set_stop_callback(STOP_CB(), None)
        del _stop_cb_keep[:]
        return
    cb = STOP_CB(lambda user, f=fn: 1 if f() else 0)
    _stop_cb_keep.append(cb)
    lib().prost_set_stop_callback(cb, None)


_output_cb_keep = []


def set_output_callback(fn):
    """`fn(text)` receives what the library prints (verbose header, "It k: Feas_p=..." lines, list_gpus) instead of the
    process's stdout -- the mexPrintf redirect of the MEX gateway (prost.cpp:15-44); None restores stdout."""
    if fn is None:
        lib().prost_set_output_callback(OUTPUT_CB(), None)
        del _output_cb_keep[:]
        return
    cb = OUTPUT_CB(lambda user, text, n, f=fn: f(C.string_at(text, n).decode(errors="replace")))
    _output_cb_keep.append(cb)
    lib().prost_set_output_callback(cb, None)


def comm_destroy():
    command("comm_destroy")
