"""ctypes binding of libpmf_hip.so (include/pmf_hip.h).

This is the Python counterpart of the Julia `ccall` shim (julia/PathMatFacHIP.jl): it marshals numpy arrays
in the reference's conventions (column-major matrices, 1-based inclusive ranges) to the C ABI.  There is NO
CPU fallback: if the shared library is missing or a call fails, a PMFError is raised.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libpmf_hip.so"

NOISE = {"normal": 0, "bernoulli": 1, "poisson": 2}
OPT = {"adagrad": 0, "adam": 1}
STORE = {"f32": 0, "bf16": 1}
TERM = {0: "max_epochs", 1: "loss_increase", 2: "abs_tol", 3: "rel_tol", 4: "nonfinite"}
PARAM = {"X": 0, "Y": 1, "logsigma": 2, "mu": 3, "logdelta": 4, "theta": 5}


class PMFError(RuntimeError):
    pass


class FitOpts(C.Structure):
    _fields_ = [("update_X", C.c_int32), ("update_Y", C.c_int32), ("update_col_layers", C.c_int32),
                ("frozen_layers", C.c_int32), ("frozen_regs", C.c_int32), ("max_epochs", C.c_int32),
                ("epoch", C.c_int32), ("tol_max_iters", C.c_int32), ("keep_trace", C.c_int32),
                ("verbosity", C.c_int32), ("print_iter", C.c_int32), ("reserved", C.c_int32),
                ("abs_tol", C.c_double), ("rel_tol", C.c_double), ("capacity", C.c_int64)]


class FitResult(C.Structure):
    _fields_ = [("term_code", C.c_int32), ("epochs", C.c_int32), ("n_trace", C.c_int32),
                ("trace_cap", C.c_int32), ("final_loss", C.c_double), ("loss_trace", C.POINTER(C.c_double)),
                ("seconds", C.c_double)]


# every symbol include/pmf_hip.h declares (checked by tests/test_abi.py against the header and the .so)
EXPORTS = [
    "pmf_last_error", "pmf_version", "pmf_device_count", "pmf_create", "pmf_destroy", "pmf_set_stream", "pmf_synchronize",
    "pmf_set_data", "pmf_set_data_device", "pmf_set_factors", "pmf_set_X", "pmf_set_Y", "pmf_get_factors",
    "pmf_set_col_params", "pmf_get_col_params", "pmf_set_n_batch_views", "pmf_set_batch_view",
    "pmf_get_batch_view", "pmf_set_noise", "pmf_clear_xreg", "pmf_add_xreg_l2", "pmf_add_xreg_group",
    "pmf_clear_yreg", "pmf_add_yreg_l2", "pmf_add_yreg_group", "pmf_add_yreg_ard", "pmf_add_yreg_fsard",
    "pmf_set_layer_regs", "pmf_set_optimizer", "pmf_set_lr", "pmf_get_lr", "pmf_reset_optimizer_state",
    "pmf_fit", "pmf_epoch_begin", "pmf_epoch_step_local", "pmf_epoch_step_shared", "pmf_epoch_loss",
    "pmf_grad_device_ptr", "pmf_get_grad", "pmf_forward", "pmf_stats", "pmf_kernel_time", "pmf_synth_data",
    "pmf_set_precision", "pmf_get_precision",
    "pmf_comm_get_unique_id", "pmf_comm_init", "pmf_comm_init_host", "pmf_comm_destroy", "pmf_comm_set_chunks",
    "pmf_comm_info", "pmf_comm_allreduce", "pmf_get_opt_state", "pmf_fsard_update_A", "pmf_debug_last_path", "pmf_debug_last_kernel",
]

COMM_ID_BYTES = 128
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)


def comm_unique_id(lib_path=None):
    """ncclGetUniqueId through the library (call on ONE rank; hand the 128 bytes to every rank)."""
    lib = load_library(lib_path)
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if lib.pmf_comm_get_unique_id(buf) != 0:
        raise PMFError(lib.pmf_last_error().decode())
    return buf.raw



def device_count(lib_path=None):
    """HIP devices visible to this process (pmf_device_count)."""
    lib = load_library(lib_path)
    n = C.c_int(0)
    if lib.pmf_device_count(C.byref(n)) != 0:
        raise PMFError(lib.pmf_last_error().decode())
    return n.value


_lib = None


def load_library(path=None):
    """Loads libpmf_hip.so (does not touch the GPU).  Raises PMFError when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise PMFError(f"{p} not found: build it with pathmatfac.jl_amd/csrc/build.sh "
                       "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64; a process must run ONE HIP runtime.  If torch is installed, load it
    # first so that libpmf_hip.so binds to the same runtime (loading the library first and torch later leaves
    # torch.cuda unusable).  A host without torch (the Julia shim) simply uses the system ROCm runtime.
    # PMF_NO_TORCH=1: a process that never uses torch (bench.py's ranks) keeps to ONE runtime, the system ROCm's.
    if os.environ.get("PMF_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(str(p))
    lib.pmf_last_error.restype = C.c_char_p
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if a declared symbol is not exported
    _lib = lib
    return lib


def _f32(a, order="F"):
    return np.require(np.asarray(a, dtype=np.float32), requirements=["F" if order == "F" else "C", "A"])


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else C.POINTER(C.c_float)()


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _flat(a):
    """None, a flat vector, or a list of per-view vectors -> one contiguous float32 vector."""
    if a is None:
        return None
    if isinstance(a, (list, tuple)) and len(a) and np.ndim(a[0]) > 0:
        a = np.concatenate([np.asarray(x, dtype=np.float32).ravel() for x in a])
    return _f32(np.asarray(a, dtype=np.float32).ravel())


def _ranges(ranges):
    s = np.ascontiguousarray([r[0] for r in ranges], dtype=np.int64)
    e = np.ascontiguousarray([r[1] for r in ranges], dtype=np.int64)
    return s, e


class Context:
    """One pmf_ctx = one GPU.  Mirrors `gpu(model)`: owns the device copies of data and parameters."""

    def __init__(self, device=0, lib_path=None):
        self.lib = load_library(lib_path)
        self._h = C.c_void_p()
        self._chk(self.lib.pmf_create(int(device), C.byref(self._h)))
        self.M = self.N = self.K = 0
        self.view_shapes = []

    def _chk(self, rc):
        if rc != 0:
            raise PMFError(self.lib.pmf_last_error().decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.pmf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data / parameters
    def set_stream(self, stream_ptr):
        """Runs the library's kernels on the caller's HIP stream, e.g. torch.cuda.current_stream().cuda_stream, so that
        they order with the caller's collectives.  torch reports its DEFAULT stream as handle 0, which pmf_set_stream
        reads as "back to the library's own (non-blocking) stream" -- kernels there would NOT be ordered with work torch
        or RCCL issue behind the default stream -- so 0 is passed on as hipStreamLegacy, the legacy default stream itself."""
        HIP_STREAM_LEGACY = 1   # hip_runtime_api.h: #define hipStreamLegacy ((hipStream_t)1)
        self._chk(self.lib.pmf_set_stream(self._h, C.c_void_p(stream_ptr if stream_ptr else HIP_STREAM_LEGACY)))
        self._stream_adopted = True

    def use_own_stream(self):
        """Back to the library's own non-blocking stream (pmf_set_stream(ctx, NULL))."""
        self._chk(self.lib.pmf_set_stream(self._h, C.c_void_p(0)))
        self._stream_adopted = False

    def synchronize(self):
        self._chk(self.lib.pmf_synchronize(self._h))

    def set_data(self, D, store="f32"):
        """store = "bf16": the device copy of D is kept as bfloat16 (PMF_STORE_BF16; read by the split-bf16 data pass)."""
        D = _f32(D)
        self.M, self.N = D.shape
        self._chk(self.lib.pmf_set_data(self._h, _fp(D), C.c_int64(self.M), C.c_int64(self.N), STORE[store]))

    def set_data_device(self, ptr, M, N, store="f32"):
        self.M, self.N = int(M), int(N)
        self._chk(self.lib.pmf_set_data_device(self._h, C.c_void_p(ptr), C.c_int64(M), C.c_int64(N), STORE[store]))

    def set_factors(self, X, Y):
        X, Y = _f32(X), _f32(Y)
        K = X.shape[0]
        if X.shape != (K, self.M) or Y.shape != (K, self.N):
            raise PMFError(f"factor shapes {X.shape}, {Y.shape} do not match data {self.M} x {self.N}")
        self.K = K
        self._chk(self.lib.pmf_set_factors(self._h, _fp(X), _fp(Y), K))

    def set_X(self, X):
        X = _f32(X)
        self._chk(self.lib.pmf_set_X(self._h, _fp(X), X.shape[0]))

    def set_Y(self, Y):
        Y = _f32(Y)
        self._chk(self.lib.pmf_set_Y(self._h, _fp(Y), Y.shape[0]))

    def get_factors(self):
        X = np.zeros((self.K, self.M), np.float32, order="F")
        Y = np.zeros((self.K, self.N), np.float32, order="F")
        self._chk(self.lib.pmf_get_factors(self._h, _fp(X), _fp(Y)))
        return X, Y

    def set_col_params(self, logsigma=None, mu=None):
        ls = None if logsigma is None else _f32(logsigma)
        m = None if mu is None else _f32(mu)
        self._chk(self.lib.pmf_set_col_params(self._h, _fp(ls), _fp(m)))

    def get_col_params(self):
        ls = np.zeros(self.N, np.float32)
        mu = np.zeros(self.N, np.float32)
        self._chk(self.lib.pmf_get_col_params(self._h, _fp(ls), _fp(mu)))
        return ls, mu

    def set_batch_views(self, views):
        """views: list of dict(start1, stop1, batch_of_row (M, 0-based), logdelta (nb x Nv), theta (nb x Nv))."""
        self._chk(self.lib.pmf_set_n_batch_views(self._h, len(views)))
        self.view_shapes = []
        for v, b in enumerate(views):
            ld, th = _f32(b["logdelta"]), _f32(b["theta"])
            nb, Nv = ld.shape
            bor = np.ascontiguousarray(b["batch_of_row"], dtype=np.int32)
            if bor.shape != (self.M,):
                raise PMFError("batch_of_row must have one entry per row")
            self._chk(self.lib.pmf_set_batch_view(self._h, v, C.c_int64(b["start1"]), C.c_int64(b["stop1"]), nb,
                                                  bor.ctypes.data_as(C.POINTER(C.c_int32)), _fp(ld), _fp(th)))
            self.view_shapes.append((nb, Nv))

    def update_batch_values(self, v, start1, stop1, batch_of_row, logdelta, theta):
        ld, th = _f32(logdelta), _f32(theta)
        bor = np.ascontiguousarray(batch_of_row, dtype=np.int32)
        self._chk(self.lib.pmf_set_batch_view(self._h, v, C.c_int64(start1), C.c_int64(stop1), ld.shape[0],
                                              bor.ctypes.data_as(C.POINTER(C.c_int32)), _fp(ld), _fp(th)))

    def get_batch_view(self, v):
        nb, Nv = self.view_shapes[v]
        ld = np.zeros((nb, Nv), np.float32, order="F")
        th = np.zeros((nb, Nv), np.float32, order="F")
        self._chk(self.lib.pmf_get_batch_view(self._h, v, _fp(ld), _fp(th)))
        return ld, th

    def set_noise(self, ranges, kinds, weights=None):
        s, e = _ranges(ranges)
        k = np.ascontiguousarray([NOISE[x] if isinstance(x, str) else int(x) for x in kinds], dtype=np.int32)
        w = None if weights is None else _f32(weights)
        self._chk(self.lib.pmf_set_noise(self._h, len(k), _i64p(s), _i64p(e), k.ctypes.data_as(C.POINTER(C.c_int32)),
                                         _fp(w)))

    # ---- regularizers
    def clear_xreg(self):
        self._chk(self.lib.pmf_clear_xreg(self._h))

    def clear_yreg(self):
        self._chk(self.lib.pmf_clear_yreg(self._h))

    def add_reg_l2(self, which, w, p=1.0):
        w = _f32(np.asarray(w).ravel())
        f = self.lib.pmf_add_xreg_l2 if which == "X" else self.lib.pmf_add_yreg_l2
        self._chk(f(self._h, _fp(w), C.c_float(p)))

    def add_reg_group(self, which, ranges, w, p=1.0):
        s, e = _ranges(ranges)
        w = np.ascontiguousarray(np.asarray(w, dtype=np.float32))  # (n_groups, K), K contiguous
        if w.shape != (len(s), self.K):
            raise PMFError(f"group weights must be (n_groups, K) = ({len(s)}, {self.K}); got {w.shape}")
        f = self.lib.pmf_add_xreg_group if which == "X" else self.lib.pmf_add_yreg_group
        self._chk(f(self._h, len(s), _i64p(s), _i64p(e), _fp(w), C.c_float(p)))

    def add_yreg_ard(self, ranges, alpha, beta, p=1.0):
        s, e = _ranges(ranges)
        a, b = _f32(np.asarray(alpha).ravel()), _f32(np.asarray(beta).ravel())
        self._chk(self.lib.pmf_add_yreg_ard(self._h, len(s), _i64p(s), _i64p(e), _fp(a), _fp(b), C.c_float(p)))

    def add_yreg_fsard(self, alpha, beta, p=1.0):
        a, b = _f32(np.asarray(alpha).ravel()), _f32(beta)
        if a.shape != (self.N,) or b.shape != (self.K, self.N):
            raise PMFError("FeatureSetARD: alpha must be N, beta K x N")
        self._chk(self.lib.pmf_add_yreg_fsard(self._h, _fp(a), _fp(b), C.c_float(p)))

    def set_layer_regs(self, ranges=None, w_logsigma=None, c_logsigma=None, w_mu=None, c_mu=None,
                       w_logdelta=None, c_logdelta=None, w_theta=None, c_theta=None):
        ranges = ranges or []
        s, e = _ranges(ranges) if ranges else (np.zeros(1, np.int64), np.zeros(1, np.int64))
        arrs = [_flat(a) for a in (w_logsigma, c_logsigma, w_mu, c_mu, w_logdelta, c_logdelta, w_theta, c_theta)]
        self._keep = arrs
        self._chk(self.lib.pmf_set_layer_regs(self._h, len(ranges), _i64p(s), _i64p(e), *[_fp(a) for a in arrs]))

    # ---- optimizer
    def set_optimizer(self, kind="adagrad", lr=1.0, eps=1e-8, beta1=0.9, beta2=0.999):
        self._chk(self.lib.pmf_set_optimizer(self._h, OPT[kind], C.c_float(lr), C.c_float(eps), C.c_float(beta1),
                                             C.c_float(beta2)))

    def set_lr(self, lr):
        self._chk(self.lib.pmf_set_lr(self._h, C.c_float(lr)))

    def get_lr(self):
        v = C.c_float(0)
        self._chk(self.lib.pmf_get_lr(self._h, C.byref(v)))
        return v.value

    def reset_optimizer_state(self):
        self._chk(self.lib.pmf_reset_optimizer_state(self._h))

    # ---- fitting
    @staticmethod
    def make_opts(update_X=False, update_Y=False, update_col_layers=False, frozen_layers=0, frozen_regs=0,
                  max_epochs=1000, epoch=1, tol_max_iters=3, keep_trace=True, verbosity=0, print_iter=10,
                  abs_tol=1e-9, rel_tol=1e-6, capacity=10 ** 8):
        o = FitOpts()
        o.update_X, o.update_Y, o.update_col_layers = int(update_X), int(update_Y), int(update_col_layers)
        o.frozen_layers, o.frozen_regs = int(frozen_layers), int(frozen_regs)
        o.max_epochs, o.epoch, o.tol_max_iters = int(max_epochs), int(epoch), int(tol_max_iters)
        o.keep_trace, o.verbosity, o.print_iter = int(keep_trace), int(verbosity), int(print_iter)
        o.abs_tol, o.rel_tol, o.capacity = float(abs_tol), float(rel_tol), int(capacity)
        return o

    def fit(self, **kw):
        o = self.make_opts(**kw)
        cap = max(o.max_epochs - o.epoch + 1, 1)
        trace = np.zeros(cap, np.float64)
        r = FitResult()
        r.trace_cap = cap
        r.loss_trace = trace.ctypes.data_as(C.POINTER(C.c_double))
        self._chk(self.lib.pmf_fit(self._h, C.byref(o), C.byref(r)))
        return {"term_code": TERM[r.term_code], "epochs": r.epochs, "loss": trace[:r.n_trace].copy(),
                "final_loss": r.final_loss, "seconds": r.seconds}

    def epoch_begin(self, o):
        self._chk(self.lib.pmf_epoch_begin(self._h, C.byref(o)))

    def epoch_step_local(self, o):
        self._chk(self.lib.pmf_epoch_step_local(self._h, C.byref(o)))

    def epoch_step_shared(self, o):
        self._chk(self.lib.pmf_epoch_step_shared(self._h, C.byref(o)))

    def epoch_loss(self):
        a, b = C.c_double(0), C.c_double(0)
        self._chk(self.lib.pmf_epoch_loss(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def grad_device_ptr(self, which):
        p, n = C.c_void_p(), C.c_int64(0)
        self._chk(self.lib.pmf_grad_device_ptr(self._h, PARAM[which], C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_grad(self, which, view=0):
        if which == "X":
            out = np.zeros((self.K, self.M), np.float32, order="F")
        elif which == "Y":
            out = np.zeros((self.K, self.N), np.float32, order="F")
        elif which in ("logsigma", "mu"):
            out = np.zeros(self.N, np.float32)
        else:
            out = np.zeros(self.view_shapes[view], np.float32, order="F")
        self._chk(self.lib.pmf_get_grad(self._h, PARAM[which], int(view), _fp(out)))
        return out

    def get_opt_state(self, which, view=0):
        """(acc, mom) of a parameter group in the parameter's shape (pmf_get_opt_state)."""
        if which == "X":
            shape = (self.K, self.M)
        elif which == "Y":
            shape = (self.K, self.N)
        elif which in ("logsigma", "mu"):
            shape = (self.N,)
        else:
            shape = self.view_shapes[view]
        acc, mom = np.zeros(shape, np.float32, order="F"), np.zeros(shape, np.float32, order="F")
        self._chk(self.lib.pmf_get_opt_state(self._h, PARAM[which], int(view), _fp(acc), _fp(mom)))
        return acc, mom

    def fsard_update_A(self, start1, stop1, S, alpha, lam, alpha0, v0, lr, ssq_grad, max_epochs=1000, term_iter=20,
                       atol=1e-5):
        """update_A! for one view on the device (pmf_fsard_update_A).  S: L x Nv, ssq_grad: L x K (updated in place).
        Returns (A [L x K], beta [K x Nv], best_loss, epochs_run)."""
        S = np.ascontiguousarray(S, dtype=np.float32)
        L, Nv = S.shape
        if Nv != stop1 - start1 + 1:
            raise PMFError("S must have one column per column of the view")
        al = np.ascontiguousarray(alpha, dtype=np.float32)
        lm = np.ascontiguousarray(lam, dtype=np.float32)
        if ssq_grad.dtype != np.float32 or not ssq_grad.flags.c_contiguous or ssq_grad.shape != (L, self.K):
            raise PMFError("ssq_grad must be a C-contiguous float32 L x K array")
        A = np.zeros((L, self.K), np.float32)
        beta = np.zeros((self.K, Nv), np.float32, order="F")
        best, ep = C.c_double(0), C.c_int(0)
        self._chk(self.lib.pmf_fsard_update_A(self._h, C.c_int64(start1), C.c_int64(stop1), int(L), _fp(S), _fp(al), _fp(lm),
                                              C.c_float(alpha0), C.c_float(v0), C.c_float(lr), _fp(ssq_grad), _fp(A),
                                              int(max_epochs), int(term_iter), C.c_double(atol), C.byref(best), C.byref(ep),
                                              _fp(beta)))
        return A, beta, best.value, ep.value

    def forward(self):
        Z = np.zeros((self.M, self.N), np.float32, order="F")
        self._chk(self.lib.pmf_forward(self._h, _fp(Z)))
        return Z

    def stats(self, use_factors=False):
        """Masked column and (batch, column) statistics (pmf_stats).  Returns a dict of float32 arrays."""
        N = self.N
        out = {k: np.zeros(N, np.float32) for k in ("n", "sum", "sumsq", "sqerr", "ssq_grad")}
        nbt = sum(nb * nv for nb, nv in self.view_shapes)
        bc, bs = np.zeros(max(nbt, 1), np.float32), np.zeros(max(nbt, 1), np.float32)
        self._chk(self.lib.pmf_stats(self._h, int(use_factors), _fp(out["n"]), _fp(out["sum"]), _fp(out["sumsq"]),
                                     _fp(out["sqerr"]), _fp(out["ssq_grad"]), _fp(bc), _fp(bs)))
        off, cnt, sq = 0, [], []
        for nb, nv in self.view_shapes:
            cnt.append(bc[off:off + nb * nv].reshape((nb, nv), order="F").copy())
            sq.append(bs[off:off + nb * nv].reshape((nb, nv), order="F").copy())
            off += nb * nv
        out["batch_count"], out["batch_sqerr"] = cnt, sq
        return out

    # ---- multi-GPU
    def comm_init(self, rank, nranks, unique_id):
        """Attaches an RCCL communicator (pmf_comm_init; collective over the ranks).  pmf_fit then shards by rows."""
        if len(unique_id) != COMM_ID_BYTES:
            raise PMFError("unique id must be 128 bytes (comm_unique_id())")
        self._chk(self.lib.pmf_comm_init(self._h, int(rank), int(nranks), C.c_char_p(bytes(unique_id))))

    def comm_init_host(self, rank, nranks, allreduce):
        """Host-staged transport: `allreduce(array)` must sum a numpy array in place over the ranks (tests: gloo)."""
        def _cb(_user, buf, count, dtype):
            try:
                ct = C.c_double if dtype == 1 else C.c_float
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ct)), shape=(int(count),))
                allreduce(arr)
                return 0
            except Exception:   # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1
        self._host_cb = HOST_ALLREDUCE_FN(_cb)   # keep alive as long as the communicator
        self._chk(self.lib.pmf_comm_init_host(self._h, int(rank), int(nranks), self._host_cb, None))

    def comm_destroy(self):
        self._chk(self.lib.pmf_comm_destroy(self._h))
        self._host_cb = None

    def comm_set_chunks(self, n):
        self._chk(self.lib.pmf_comm_set_chunks(self._h, int(n)))

    def comm_allreduce(self, arr, op="sum"):
        """In-place sum / max over the ranks of a float32 or float64 numpy array (pmf_comm_allreduce)."""
        if arr.dtype not in (np.float32, np.float64) or not arr.flags.c_contiguous:
            raise PMFError("comm_allreduce needs a contiguous float32 / float64 array")
        self._chk(self.lib.pmf_comm_allreduce(self._h, arr.ctypes.data_as(C.c_void_p), C.c_int64(arr.size),
                                              1 if arr.dtype == np.float64 else 0, {"sum": 0, "max": 1}[op]))
        return arr

    def comm_info(self):
        r, n, t, ch, cu = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        nc = C.c_int64(0)
        self._chk(self.lib.pmf_comm_info(self._h, C.byref(r), C.byref(n), C.byref(t), C.byref(ch), C.byref(cu), C.byref(nc)))
        return dict(rank=r.value, nranks=n.value, transport=("none", "rccl", "host")[t.value], n_chunks=ch.value,
                    reserved_cus=cu.value, n_collectives=nc.value)

    def last_path(self):
        """Batch-layer variants of the last launches: dict(bmode=0|1|2, layer_path=0|1|2, slots=...), see pmf_hip.h."""
        b, l, s = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(self.lib.pmf_debug_last_path(self._h, C.byref(b), C.byref(l), C.byref(s)))
        return dict(bmode=b.value, layer_path=l.value, slots=s.value)

    def last_kernel(self):
        """Kernel family of the last fused data pass: 0 exact, 1 / 2 / 4 / 8 the split kernels sb / sb2 / sb4 / sb8."""
        k = C.c_int(0)
        self._chk(self.lib.pmf_debug_last_kernel(self._h, C.byref(k)))
        return k.value

    def set_precision(self, mode):
        """'f32' (exact f32 MFMA, default) or 'bf16x3' (split-bf16 products where a kernel variant exists)."""
        self._chk(self.lib.pmf_set_precision(self._h, {"f32": 0, "bf16x3": 1}[mode]))

    def get_precision(self):
        mode, n = C.c_int(0), C.c_int64(0)
        self._chk(self.lib.pmf_get_precision(self._h, C.byref(mode), C.byref(n)))
        return ("f32", "bf16x3")[mode.value], n.value

    def kernel_time(self, reset=False):
        ms, n = C.c_double(0), C.c_int64(0)
        self._chk(self.lib.pmf_kernel_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def synth_data(self, seed=20260104, noise=0.1, frac_nan=0.0):
        self._chk(self.lib.pmf_synth_data(self._h, C.c_uint64(seed), C.c_float(noise), C.c_float(frac_nan)))
