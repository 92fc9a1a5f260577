"""ctypes wrapper around oracle/pmf_oracle.c  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (pathmatfac.jl_amd/) never does.

The wrapper takes plain numpy arrays in the reference's (Julia) conventions: column-major
matrices are passed as numpy arrays of shape (rows, cols) in any memory order (they are
converted to Fortran order), ranges are 1-based inclusive.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_BUILD = _HERE / "_build"

KIND = {"normal": 0, "bernoulli": 1, "poisson": 2}
REG = {"none": 0, "l2": 1, "group": 2, "ard": 3, "fsard": 4}
OPT = {"adagrad": 0, "adam": 1}
TERM = {0: "max_epochs", 1: "loss_increase", 2: "abs_tol", 3: "rel_tol", 4: "nonfinite"}


def build(force=False):
    """Compile both oracle builds with the committed Makefile (gcc)."""
    libs = [_BUILD / "liboracle64.so", _BUILD / "liboracle32.so"]
    src = _HERE / "pmf_oracle.c"
    if force or not all(p.exists() and p.stat().st_mtime >= src.stat().st_mtime for p in libs):
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return libs


def _mk(real):
    class RegTerm(C.Structure):
        _fields_ = [("kind", C.c_int32), ("n_ranges", C.c_int32), ("p", real),
                    ("start1", C.POINTER(C.c_int64)), ("stop1", C.POINTER(C.c_int64)),
                    ("w", C.POINTER(real)), ("a", C.POINTER(real)), ("b", C.POINTER(real))]

    class Model(C.Structure):
        _fields_ = [("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int32), ("n_bv", C.c_int32),
                    ("D", C.POINTER(C.c_float)), ("X", C.POINTER(real)), ("Y", C.POINTER(real)),
                    ("logsigma", C.POINTER(real)), ("mu", C.POINTER(real)),
                    ("bv_start1", C.POINTER(C.c_int64)), ("bv_stop1", C.POINTER(C.c_int64)),
                    ("bv_nb", C.POINTER(C.c_int32)), ("bv_bor", C.POINTER(C.c_int32)),
                    ("bv_off", C.POINTER(C.c_int64)),
                    ("logdelta", C.POINTER(real)), ("theta", C.POINTER(real)),
                    ("has_batch", C.c_int32), ("n_noise", C.c_int32),
                    ("nz_start1", C.POINTER(C.c_int64)), ("nz_stop1", C.POINTER(C.c_int64)),
                    ("nz_kind", C.POINTER(C.c_int32)), ("col_weight", C.POINTER(real)),
                    ("n_xreg", C.c_int32), ("n_yreg", C.c_int32),
                    ("xreg", C.POINTER(RegTerm)), ("yreg", C.POINTER(RegTerm)),
                    ("has_colreg", C.c_int32), ("n_cr", C.c_int32),
                    ("cr_start1", C.POINTER(C.c_int64)), ("cr_stop1", C.POINTER(C.c_int64)),
                    ("cr_w_logsigma", C.POINTER(real)), ("cr_c_logsigma", C.POINTER(real)),
                    ("cr_w_mu", C.POINTER(real)), ("cr_c_mu", C.POINTER(real)),
                    ("has_batchreg", C.c_int32), ("bvb_off", C.POINTER(C.c_int64)),
                    ("br_w_logdelta", C.POINTER(real)), ("br_c_logdelta", C.POINTER(real)),
                    ("br_w_theta", C.POINTER(real)), ("br_c_theta", C.POINTER(real))]

    class Opts(C.Structure):
        _fields_ = [("update_X", C.c_int32), ("update_Y", C.c_int32), ("update_col_layers", C.c_int32),
                    ("frozen_layers", C.c_int32), ("frozen_regs", C.c_int32), ("opt_kind", C.c_int32),
                    ("max_epochs", C.c_int32), ("epoch", C.c_int32), ("tol_max_iters", C.c_int32),
                    ("chunk_rows", C.c_int32),
                    ("lr", real), ("eps", real), ("beta1", real), ("beta2", real),
                    ("abs_tol", real), ("rel_tol", real)]

    class OptState(C.Structure):
        _fields_ = [("acc", C.POINTER(real) * 6), ("mom", C.POINTER(real) * 6),
                    ("bp1", real * 6), ("bp2", real * 6), ("initialized", C.c_int32)]

    return RegTerm, Model, Opts, OptState


class _Lib:
    def __init__(self, precision):
        build()
        self.precision = precision
        self.np_real = np.float64 if precision == 64 else np.float32
        self.c_real = C.c_double if precision == 64 else C.c_float
        self.lib = C.CDLL(str(_BUILD / f"liboracle{precision}.so"))
        assert self.lib.o_sizeof_real() == (8 if precision == 64 else 4)
        self.RegTerm, self.Model, self.Opts, self.OptState = _mk(self.c_real)
        self.lib.o_data_pass.restype = C.c_double
        self.lib.o_loss_and_grads.restype = C.c_double
        self.lib.o_regterm_apply.restype = C.c_double
        self.lib.o_colparamreg.restype = C.c_double
        self.lib.o_batcharrayreg.restype = C.c_double


_LIBS = {}


def get_lib(precision=64):
    if precision not in _LIBS:
        _LIBS[precision] = _Lib(precision)
    return _LIBS[precision]


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64).ravel())


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel())


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else C.POINTER(ctype)()


class OracleModel:
    """Holds one problem instance for the C oracle.  Parameters live in numpy arrays owned here
    (self.X, self.Y, self.logsigma, self.mu, self.logdelta[v], self.theta[v]) and are updated in place by fit()."""

    def __init__(self, D, X, Y, logsigma=None, mu=None, batch_views=None, noise=None, col_weight=None,
                 xreg=None, yreg=None, colreg=None, batchreg=None, precision=64):
        L = self.L = get_lib(precision)
        R = L.np_real
        self.D = np.asfortranarray(np.asarray(D, dtype=np.float32))
        self.M, self.N = self.D.shape
        self.X = np.asfortranarray(np.asarray(X, dtype=R))
        self.Y = np.asfortranarray(np.asarray(Y, dtype=R))
        self.K = self.X.shape[0]
        assert self.X.shape == (self.K, self.M) and self.Y.shape == (self.K, self.N)
        self.logsigma = np.zeros(self.N, R) if logsigma is None else np.array(logsigma, dtype=R)
        self.mu = np.zeros(self.N, R) if mu is None else np.array(mu, dtype=R)
        # batch views: list of dicts(start1, stop1, batch_of_row (M int, 0-based), logdelta (nb x Nv), theta (nb x Nv))
        self.batch_views = batch_views or []
        self.has_batch = batch_views is not None and len(batch_views) > 0
        nbv = len(self.batch_views)
        self._bv_start1 = _i64([b["start1"] for b in self.batch_views])
        self._bv_stop1 = _i64([b["stop1"] for b in self.batch_views])
        nbs, offs, bors = [], [0], []
        for b in self.batch_views:
            ld = np.asarray(b["logdelta"])
            nb, Nv = ld.shape
            assert Nv == b["stop1"] - b["start1"] + 1
            nbs.append(nb)
            offs.append(offs[-1] + nb * Nv)
            bor = np.asarray(b["batch_of_row"], dtype=np.int32)
            assert bor.shape == (self.M,) and bor.max() < nb
            bors.append(bor)
        self._bv_nb = _i32(nbs)
        self._bv_off = _i64(offs)
        self._bvb_off = _i64(np.concatenate([[0], np.cumsum(nbs)]) if nbv else [0])
        self._bv_bor = _i32(np.concatenate(bors)) if nbv else _i32([])
        self.logdelta_flat = np.concatenate(
            [np.asfortranarray(np.asarray(b["logdelta"], dtype=R)).ravel(order="F") for b in self.batch_views]
        ) if nbv else np.zeros(0, R)
        self.theta_flat = np.concatenate(
            [np.asfortranarray(np.asarray(b["theta"], dtype=R)).ravel(order="F") for b in self.batch_views]
        ) if nbv else np.zeros(0, R)
        # noise model: list of (start1, stop1, kind)
        noise = noise or [(1, self.N, "normal")]
        self._nz_start1 = _i64([n[0] for n in noise])
        self._nz_stop1 = _i64([n[1] for n in noise])
        self._nz_kind = _i32([KIND[n[2]] if isinstance(n[2], str) else n[2] for n in noise])
        self.col_weight = np.ones(self.N, R) if col_weight is None else np.array(col_weight, dtype=R)
        self._keep = []
        self._xreg = self._mk_terms(xreg or [], self.M)
        self._yreg = self._mk_terms(yreg or [], self.N)
        # colreg: dict(start1, stop1, w_logsigma, c_logsigma, w_mu, c_mu)  (ColParamReg x2)
        self.colreg = colreg
        self.batchreg = batchreg  # dict(w_logdelta, c_logdelta, w_theta, c_theta): each list (per view) of nb-vectors
        self._build_struct()
        self.state = None

    def _mk_terms(self, terms, n):
        L = self.L
        R = L.np_real
        arr = (L.RegTerm * max(len(terms), 1))()
        for t_i, t in enumerate(terms):
            rt = arr[t_i]
            rt.kind = REG[t["kind"]]
            rt.p = t.get("p", 1.0)
            if t["kind"] in ("group", "ard"):
                s, e = _i64(t["start1"]), _i64(t["stop1"])
                self._keep += [s, e]
                rt.n_ranges = len(s)
                rt.start1, rt.stop1 = _ptr(s, C.c_int64), _ptr(e, C.c_int64)
            if t["kind"] == "l2":
                w = np.ascontiguousarray(np.asarray(t["w"], dtype=R).ravel())
                assert w.size == self.K
                self._keep.append(w)
                rt.w = _ptr(w, L.c_real)
            elif t["kind"] == "group":
                w = np.ascontiguousarray(np.asarray(t["w"], dtype=R))  # (n_ranges, K)
                assert w.shape == (rt.n_ranges, self.K)
                self._keep.append(w)
                rt.w = _ptr(w, L.c_real)
            elif t["kind"] == "ard":
                a = np.ascontiguousarray(np.asarray(t["a"], dtype=R).ravel())
                b = np.ascontiguousarray(np.asarray(t["b"], dtype=R).ravel())
                self._keep += [a, b]
                rt.a, rt.b = _ptr(a, L.c_real), _ptr(b, L.c_real)
            elif t["kind"] == "fsard":
                a = np.ascontiguousarray(np.asarray(t["alpha"], dtype=R).ravel())
                b = np.asfortranarray(np.asarray(t["beta"], dtype=R))
                assert a.size == n and b.shape == (self.K, n)
                self._keep += [a, b]
                rt.a, rt.b = _ptr(a, L.c_real), _ptr(b, L.c_real)
        return arr, len(terms)

    def _build_struct(self):
        L = self.L
        R = L.np_real
        m = self.m = L.Model()
        m.M, m.N, m.K, m.n_bv = self.M, self.N, self.K, len(self.batch_views)
        m.D = _ptr(self.D, C.c_float)
        m.X, m.Y = _ptr(self.X, L.c_real), _ptr(self.Y, L.c_real)
        m.logsigma, m.mu = _ptr(self.logsigma, L.c_real), _ptr(self.mu, L.c_real)
        m.bv_start1, m.bv_stop1 = _ptr(self._bv_start1, C.c_int64), _ptr(self._bv_stop1, C.c_int64)
        m.bv_nb, m.bv_bor = _ptr(self._bv_nb, C.c_int32), _ptr(self._bv_bor, C.c_int32)
        m.bv_off = _ptr(self._bv_off, C.c_int64)
        m.logdelta, m.theta = _ptr(self.logdelta_flat, L.c_real), _ptr(self.theta_flat, L.c_real)
        m.has_batch = int(self.has_batch)
        m.n_noise = len(self._nz_kind)
        m.nz_start1, m.nz_stop1 = _ptr(self._nz_start1, C.c_int64), _ptr(self._nz_stop1, C.c_int64)
        m.nz_kind = _ptr(self._nz_kind, C.c_int32)
        m.col_weight = _ptr(self.col_weight, L.c_real)
        m.xreg, m.n_xreg = self._xreg[0], self._xreg[1]
        m.yreg, m.n_yreg = self._yreg[0], self._yreg[1]
        m.bvb_off = _ptr(self._bvb_off, C.c_int64)
        if self.colreg is not None:
            cr = self.colreg
            self._cr = [_i64(cr["start1"]), _i64(cr["stop1"])] + [
                np.ascontiguousarray(np.asarray(cr[k], dtype=R).ravel())
                for k in ("w_logsigma", "c_logsigma", "w_mu", "c_mu")]
            m.has_colreg, m.n_cr = 1, len(self._cr[0])
            m.cr_start1, m.cr_stop1 = _ptr(self._cr[0], C.c_int64), _ptr(self._cr[1], C.c_int64)
            m.cr_w_logsigma, m.cr_c_logsigma = _ptr(self._cr[2], L.c_real), _ptr(self._cr[3], L.c_real)
            m.cr_w_mu, m.cr_c_mu = _ptr(self._cr[4], L.c_real), _ptr(self._cr[5], L.c_real)
        if self.batchreg is not None and self.has_batch:
            br = self.batchreg
            self._br = [np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=R).ravel() for v in br[k]]))
                        for k in ("w_logdelta", "c_logdelta", "w_theta", "c_theta")]
            for a in self._br:
                assert a.size == self._bvb_off[-1]
            m.has_batchreg = 1
            m.br_w_logdelta, m.br_c_logdelta = _ptr(self._br[0], L.c_real), _ptr(self._br[1], L.c_real)
            m.br_w_theta, m.br_c_theta = _ptr(self._br[2], L.c_real), _ptr(self._br[3], L.c_real)

    # ---- views of the flat BatchArray values as per-view (nb x Nv) matrices
    def _unflat(self, flat):
        out = []
        for v, b in enumerate(self.batch_views):
            nb = int(self._bv_nb[v])
            Nv = b["stop1"] - b["start1"] + 1
            out.append(flat[self._bv_off[v]:self._bv_off[v + 1]].reshape((nb, Nv), order="F"))
        return out

    @property
    def logdelta(self):
        return self._unflat(self.logdelta_flat)

    @property
    def theta(self):
        return self._unflat(self.theta_flat)

    def make_opts(self, update_X=False, update_Y=False, update_col_layers=False, frozen_layers=0,
                  frozen_regs=0, opt="adagrad", lr=1.0, eps=1e-8, beta1=0.9, beta2=0.999, max_epochs=1000,
                  epoch=1, abs_tol=1e-9, rel_tol=1e-6, tol_max_iters=3, chunk_rows=0):
        o = self.L.Opts()
        o.update_X, o.update_Y, o.update_col_layers = int(update_X), int(update_Y), int(update_col_layers)
        o.frozen_layers, o.frozen_regs = int(frozen_layers), int(frozen_regs)
        o.opt_kind = OPT[opt]
        o.max_epochs, o.epoch, o.tol_max_iters, o.chunk_rows = max_epochs, epoch, tol_max_iters, chunk_rows
        o.lr, o.eps, o.beta1, o.beta2, o.abs_tol, o.rel_tol = lr, eps, beta1, beta2, abs_tol, rel_tol
        return o

    def forward(self):
        Z = np.zeros((self.M, self.N), self.L.np_real, order="F")
        self.L.lib.o_forward(C.byref(self.m), _ptr(Z, self.L.c_real))
        return Z

    def loss_and_grads(self, **kw):
        L = self.L
        o = self.make_opts(**kw)
        R = L.np_real
        g = {"X": np.zeros_like(self.X), "Y": np.zeros_like(self.Y), "logsigma": np.zeros(self.N, R),
             "mu": np.zeros(self.N, R), "logdelta": np.zeros_like(self.logdelta_flat),
             "theta": np.zeros_like(self.theta_flat)}
        dl = C.c_double(0)
        loss = L.lib.o_loss_and_grads(C.byref(self.m), C.byref(o), _ptr(g["X"], L.c_real), _ptr(g["Y"], L.c_real),
                                      _ptr(g["logsigma"], L.c_real), _ptr(g["mu"], L.c_real),
                                      _ptr(g["logdelta"], L.c_real), _ptr(g["theta"], L.c_real), C.byref(dl))
        g["logdelta"] = self._unflat(g["logdelta"])
        g["theta"] = self._unflat(g["theta"])
        g["data_loss"] = dl.value
        return loss, g

    def stats(self, use_factors=False):
        L = self.L
        R = L.np_real
        out = {k: np.zeros(self.N, R) for k in ("n", "sum", "sumsq", "sqerr", "ssq_grad")}
        bn, bs = np.zeros(max(self.theta_flat.size, 1), R), np.zeros(max(self.theta_flat.size, 1), R)
        L.lib.o_stats(C.byref(self.m), int(use_factors), _ptr(out["n"], L.c_real), _ptr(out["sum"], L.c_real),
                      _ptr(out["sumsq"], L.c_real), _ptr(out["sqerr"], L.c_real), _ptr(out["ssq_grad"], L.c_real),
                      _ptr(bn, L.c_real), _ptr(bs, L.c_real))
        out["batch_count"] = self._unflat(bn[:self.theta_flat.size])
        out["batch_sqerr"] = self._unflat(bs[:self.theta_flat.size])
        return out

    def reset_optimizer(self):
        self.state = None

    def _ensure_state(self):
        L = self.L
        if self.state is None:
            st = L.OptState()
            bufs = [self.X, self.Y, self.logsigma, self.mu, self.logdelta_flat, self.theta_flat]
            self._st_bufs = []
            for w, b in enumerate(bufs):
                a = np.zeros(b.size, L.np_real)
                mo = np.zeros(b.size, L.np_real)
                self._st_bufs += [a, mo]
                st.acc[w] = _ptr(a, L.c_real)
                st.mom[w] = _ptr(mo, L.c_real)
            st.initialized = 0
            self.state = st
        return self.state

    def fit(self, **kw):
        """Runs o_fit. Optimizer state persists across calls until reset_optimizer() (fit.jl:55 creates
        the optimizer once per mf_fit_adapt_lr! call and re-uses it across LR halvings)."""
        L = self.L
        o = self.make_opts(**kw)
        st = self._ensure_state()
        cap = max(o.max_epochs - o.epoch + 1, 1)
        trace = np.zeros(cap, np.float64)
        n_trace, epochs = C.c_int32(0), C.c_int32(0)
        term = L.lib.o_fit(C.byref(self.m), C.byref(o), C.byref(st), _ptr(trace, C.c_double), cap,
                           C.byref(n_trace), C.byref(epochs))
        return {"term_code": TERM[term], "epochs": epochs.value, "loss": trace[:n_trace.value].copy()}
