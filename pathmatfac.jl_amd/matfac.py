"""MatFacModel and the replacement of `MF.fit!` (the un-vendored MatFac.jl inner loop, called at src/fit.jl:24).

`fit_(matfac, ctx, ...)` marshals the model through the C ABI (include/pmf_hip.h), runs pmf_fit on the GPU (or
the row-sharded loop of parallel.py when torch.distributed is initialised with more than one rank) and copies
the trained parameters back into the model's numpy arrays -- exactly what the Julia shim's `mf_fit!` does."""
import numpy as np

from .layers import BatchScale, BatchShift, ColScale, ColShift, FrozenLayer, Identity
from .optimizers import AdaGrad
from .regularizers import (BatchArrayReg, ColParamReg, FrozenRegularizer, SequenceReg, ZeroReg)
from .util import ids_to_ranges, unique

VALID_LOSSES = ["normal", "bernoulli", "bernoulli_sq_hinge", "poisson", "ordinal3", "ordinal_sq_hinge3"]  # util.jl:128
SUPPORTED_LOSSES = ("normal", "bernoulli", "poisson")


class CompositeNoise:
    """MatFac's CompositeNoise{noises, col_ranges} as used at src/fit.jl:227, plus the per-column weights set by
    MF.set_weight! (src/fit.jl:157, 180)."""

    def __init__(self, feature_distributions):
        dists = list(feature_distributions)
        self.col_ranges = tuple(ids_to_ranges(dists))
        self.noises = tuple(unique(dists))
        for d in self.noises:
            if d not in VALID_LOSSES:
                raise ValueError(f"unknown distribution {d!r}; valid: {VALID_LOSSES}")
            if d not in SUPPORTED_LOSSES:
                raise NotImplementedError(f"noise model {d!r} is not implemented by the HIP path "
                                          f"(supported: {SUPPORTED_LOSSES}); see DESIGN.md 'out of scope'")
        self.weights = np.ones(len(dists), dtype=np.float32)

    def set_weight_(self, w):
        self.weights = np.asarray(w, dtype=np.float32).copy()


class MatFacModel:
    """MatFacModel(M, N, K, feature_distributions; col_transform, X_reg, Y_reg, col_transform_reg) (src/model.jl:69-72).
    X is K x M, Y is K x N (column-major semantics; stored as Fortran-ordered float32)."""

    def __init__(self, M, N, K, feature_distributions, col_transform=None, X_reg=None, Y_reg=None,
                 col_transform_reg=None, rng=None):
        rng = rng or np.random.default_rng()
        # MatFac's own initialisation is not visible (un-vendored); self-specified: N(0,1)/sqrt(K)
        self.X = np.asfortranarray((rng.standard_normal((K, M)) / np.sqrt(K)).astype(np.float32))
        self.Y = np.asfortranarray((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32))
        self.col_transform = col_transform
        self.noise_model = CompositeNoise(feature_distributions)
        self.X_reg = X_reg if X_reg is not None else ZeroReg()
        self.Y_reg = Y_reg if Y_reg is not None else ZeroReg()
        self.col_transform_reg = col_transform_reg

    @property
    def K(self):
        return self.X.shape[0]


# --------------------------------------------------------------------------------------------------------
def _batch_views(col_transform):
    """The BatchArrays of layers 2 and 4 as the C ABI's per-view dicts (they share ranges and row batches by
    construction, layers.jl:247-248).  Returns ([], None, None) when the layers are the identity."""
    l2, l4 = col_transform.unwrapped(2), col_transform.unwrapped(4)
    ba2 = l2.logdelta if isinstance(l2, BatchScale) else None
    ba4 = l4.theta if isinstance(l4, BatchShift) else None
    ref = ba2 if ba2 is not None else ba4
    if ref is None:
        return [], None, None
    views = []
    for v, cr in enumerate(ref.col_ranges):
        nb, Nv = ref.values[v].shape
        ld = ba2.values[v] if ba2 is not None else np.zeros((nb, Nv))
        th = ba4.values[v] if ba4 is not None else np.zeros((nb, Nv))
        views.append(dict(start1=cr.start, stop1=cr.stop, batch_of_row=ref.row_batches[v], logdelta=ld, theta=th))
    return views, ba2, ba4


def marshal(mf, ctx, with_xreg=True, with_yreg=True):
    """Host model -> device (parameters, noise model, regularizers).  The data matrix is handled by the caller.
    A regularizer is only marshalled for a factor that is going to be updated: the loop never evaluates the others
    (and the reference keeps an X_reg of the wrong shape attached while regressing Y only, src/fit.jl:397-428)."""
    ctx.set_factors(mf.X, mf.Y)
    ct = mf.col_transform
    l1, l3 = ct.unwrapped(1), ct.unwrapped(3)
    ctx.set_col_params(l1.logsigma if isinstance(l1, ColScale) else np.zeros(ctx.N, np.float32),
                       l3.mu if isinstance(l3, ColShift) else np.zeros(ctx.N, np.float32))
    views, _, _ = _batch_views(ct)
    ctx.set_batch_views(views)
    nm = mf.noise_model
    ctx.set_noise([(r.start, r.stop) for r in nm.col_ranges], list(nm.noises), nm.weights)
    ctx.clear_xreg()
    if with_xreg:
        mf.X_reg.add_to(ctx, "X")
    ctx.clear_yreg()
    if with_yreg:
        mf.Y_reg.add_to(ctx, "Y")
    sr = mf.col_transform_reg
    kw = {}
    if isinstance(sr, SequenceReg):
        regs = [r.reg if isinstance(r, FrozenRegularizer) else r for r in sr.regs]
        if isinstance(regs[0], ColParamReg) and isinstance(regs[2], ColParamReg):
            kw.update(ranges=[(r.start, r.stop) for r in regs[0].col_ranges],
                      w_logsigma=np.array(regs[0].weights, np.float32), c_logsigma=np.array(regs[0].centers, np.float32),
                      w_mu=np.array(regs[2].weights, np.float32), c_mu=np.array(regs[2].centers, np.float32))
        if views and isinstance(regs[1], BatchArrayReg) and isinstance(regs[3], BatchArrayReg):
            kw.update(w_logdelta=list(regs[1].weights), c_logdelta=list(regs[1].centers),
                      w_theta=list(regs[3].weights), c_theta=list(regs[3].centers))
    ctx.set_layer_regs(**kw)


def unmarshal(mf, ctx, update_X, update_Y, update_col_layers):
    """Device -> host for the parameter groups that were trained."""
    if update_X or update_Y:
        X, Y = ctx.get_factors()
        if update_X:
            mf.X[...] = X
        if update_Y:
            mf.Y[...] = Y
    if update_col_layers:
        ct = mf.col_transform
        ls, mu = ctx.get_col_params()
        l1, l3 = ct.unwrapped(1), ct.unwrapped(3)
        if isinstance(l1, ColScale):
            l1.logsigma[...] = ls
        if isinstance(l3, ColShift):
            l3.mu[...] = mu
        views, ba2, ba4 = _batch_views(ct)
        for v in range(len(views)):
            ld, th = ctx.get_batch_view(v)
            if ba2 is not None:
                ba2.values[v][...] = ld
            if ba4 is not None:
                ba4.values[v][...] = th


def fit_(mf, ctx, opt=None, lr=0.01, update_X=False, update_Y=False, update_col_layers=False, max_epochs=1000,
         epoch=1, abs_tol=1e-9, rel_tol=1e-6, tol_max_iters=3, verbosity=0, print_iter=10, capacity=10 ** 8,
         keep_history=True, dist=None, group=None, **ignored):
    """MF.fit!(matfac, data; ...) -> history dict with "term_code", "epochs", "loss" (src/fit.jl:24-38, 61-69)."""
    opt = opt if opt is not None else AdaGrad(lr)
    marshal(mf, ctx, with_xreg=update_X, with_yreg=update_Y)
    if opt._bound_ctx != id(ctx):          # a new optimizer object: fresh accumulators (src/fit.jl:55)
        ctx.set_optimizer(**opt.params())
        opt._bound_ctx = id(ctx)
    else:                                   # same object, possibly halved eta (src/fit.jl:64): keep the state
        ctx.set_lr(opt.eta)
    ct, sr = mf.col_transform, mf.col_transform_reg
    kw = dict(update_X=update_X, update_Y=update_Y, update_col_layers=update_col_layers,
              frozen_layers=ct.frozen_mask() | sum(1 << i for i, l in enumerate(ct.layers) if isinstance(l, Identity)),
              frozen_regs=sr.frozen_mask() if isinstance(sr, SequenceReg) else 0,
              max_epochs=max_epochs, epoch=epoch, abs_tol=abs_tol, rel_tol=rel_tol, tol_max_iters=tol_max_iters,
              verbosity=verbosity, print_iter=print_iter)
    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        from . import parallel
        h = parallel.fit_distributed(ctx, dist=dist, group=group, **kw)
    else:
        h = ctx.fit(capacity=capacity, **kw)
    unmarshal(mf, ctx, update_X, update_Y, update_col_layers)
    return h
