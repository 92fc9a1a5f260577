"""Fit orchestration above the drop-in boundary (src/fit.jl): `mf_fit!`, `mf_fit_adapt_lr!` and the stage
drivers that only need the gradient-descent loop.  Trailing underscore = the reference's `!`."""
import time

import numpy as np

from . import matfac as MF
from .layers import freeze_layer_, unfreeze_layer_
from .optimizers import AdaGrad

FIT_START_TIME = time.time()


def history_(hist, d=None, **kw):
    """history! (src/util.jl:607-622): the kwargs form stores elapsed seconds, the Dict form absolute time (Q7)."""
    if hist is None:
        return
    if d is None:
        d = {str(k): v for k, v in kw.items()}
        d["time"] = time.time() - FIT_START_TIME
    else:
        d = dict(d)
        d.update({str(k): v for k, v in kw.items()})
        d["time"] = time.time()
    hist.append(d)


def mf_fit_(model, scale_column_losses=False, update_X=False, update_Y=False, update_row_layers=False,
            update_col_layers=False, update_noise_models=True, reg_relative_weighting=False, update_X_reg=False,
            update_Y_reg=False, update_row_layers_reg=False, update_col_layers_reg=False, keep_history=True,
            device=0, **kwargs):
    """mf_fit! (src/fit.jl:9-38): the drop-in boundary.  MF.fit! is replaced by the HIP library."""
    ctx = model.device_context(device)
    return MF.fit_(model.matfac, ctx, update_X=update_X, update_Y=update_Y, update_col_layers=update_col_layers,
                   keep_history=keep_history, **kwargs)


def construct_optimizer(model, lr):
    """src/fit.jl:41-43."""
    return AdaGrad(lr)


def mf_fit_adapt_lr_(model, lr=1.0, min_lr=0.001, max_epochs=1000, history=None, keep_history=True, verbosity=1,
                     print_prefix="", **kwargs):
    """mf_fit_adapt_lr! (src/fit.jl:46-75): halve eta on "loss_increase" until eta < min_lr, resuming the epoch count."""
    opt = construct_optimizer(model, lr)
    epoch = 1
    last = None
    while epoch <= max_epochs:
        h = mf_fit_(model, opt=opt, max_epochs=max_epochs, epoch=epoch, keep_history=True,
                    print_prefix=print_prefix, verbosity=verbosity, **kwargs)
        history_(history, h, name=f"mf_fit_lr={opt.eta}")
        last = h
        if h["term_code"] == "loss_increase":
            opt.eta *= np.float32(0.5)
            opt.eta = float(np.float32(opt.eta))
            if opt.eta < min_lr:
                break
            if verbosity > 0:
                print(f"{print_prefix}Resuming with smaller learning rate ({opt.eta})")
            epoch = h["epochs"]
        else:
            break
    return last


def init_theta_(model, capacity=int(10e8), max_epochs=500, lr_theta=1.0, verbosity=1, print_prefix="",
                history=None, **kwargs):
    """init_theta! (src/fit.jl:106-122): freeze layers 1:3, train BatchShift only."""
    ct = model.matfac.col_transform
    freeze_layer_(ct, [1, 2, 3])
    try:
        mf_fit_adapt_lr_(model, lr=lr_theta, update_col_layers=True, capacity=capacity, max_epochs=max_epochs,
                         verbosity=verbosity, print_prefix=print_prefix, history=history)
    finally:
        unfreeze_layer_(ct, [1, 2, 3])
    history_(history, name="init_theta")


def init_factors_(model, verbosity=1, print_prefix="", history=None, lr=1.0, capacity=10 ** 8, max_epochs=1000,
                  init_factors_method="adagrad", rel_tol=1e-5, abs_tol=1e-5, **kwargs):
    """init_factors! (src/fit.jl:249-288), AdaGrad branch (the L-BFGS branch is out of scope).
    `rel_tol` / `abs_tol` are captured here as in the reference (:256-257) and belong to the L-BFGS branch only: the
    AdaGrad branch (:280-284) does not pass them on, so that stage stops on MF.fit!'s own default tolerances."""
    if init_factors_method != "adagrad":
        raise NotImplementedError("init_factors_method='lbfgs' is out of scope (src/fit_lbfgs.jl)")
    mf_fit_adapt_lr_(model, capacity=capacity, update_X=True, update_Y=True, lr=lr, min_lr=0.05,
                     max_epochs=max_epochs, verbosity=verbosity, print_prefix=print_prefix + "    ",
                     history=history, **kwargs)
    history_(history, name="init_factors")


# =========================================================================================================
# "Next" rows N1 / N2 / N4: the closed-form initialisers and post-processing steps that sit
# between the gradient-descent stages.  Every pass over the data matrix runs on the device (pmf_stats / pmf_fit);
# what remains on the host is O(N) / O(nb x N_v) / O(K^2 N) arithmetic on parameter arrays.
# The MatFac helpers these call (compute_M_estimates, link_col_sqerr, column_nonnan, batched_column_ssq_grads,
# batched_column_nanvar, sqerr_func) are un-vendored: their semantics are self-specified in DESIGN.md section 2.
# =========================================================================================================
from . import regularizers as _R            # noqa: E402
from .layers import BatchScale, BatchShift, ColScale, ColShift, Identity  # noqa: E402
from .regularizers import (ARDRegularizer, FeatureSetARDReg, GroupRegularizer, L2Regularizer, SequenceReg,  # noqa: E402
                           ZeroReg, freeze_reg_, unfreeze_reg_)
from .util import ids_to_ind_mat, ids_to_ranges  # noqa: E402
import copy  # noqa: E402


def _device_stats(model, use_factors, device=0):
    ctx = model.device_context(device)
    MF.marshal(model.matfac, ctx, with_xreg=False, with_yreg=False)
    return ctx.stats(use_factors)


def init_mu_(model, capacity=int(10e8), lr_mu=0.1, max_epochs=500, verbosity=1, print_prefix="", history=None,
             **kwargs):
    """init_mu! (src/fit.jl:82-103): mu <- per-column M-estimates argmin_m sum_i loss(m, D_ij), found by AdaGrad on mu
    alone with X'Y = 0 (MF.compute_M_estimates; rel_tol 1e-5, abs_tol 1e-3)."""
    mf = model.matfac
    ct = mf.col_transform
    X0, Y0 = mf.X.copy(), mf.Y.copy()
    mf.X[...] = 0
    mf.Y[...] = 0
    was_frozen = ct.frozen_mask()
    freeze_layer_(ct, [1, 2, 4])
    sr = mf.col_transform_reg
    reg_frozen = sr.frozen_mask() if isinstance(sr, SequenceReg) else 0
    if isinstance(sr, SequenceReg):
        freeze_reg_(sr, [1, 2, 3, 4])
    try:
        h = mf_fit_(model, opt=construct_optimizer(model, lr_mu), update_col_layers=True, max_epochs=max_epochs,
                    rel_tol=1e-5, abs_tol=1e-3, verbosity=verbosity - 1, print_prefix=print_prefix)
    finally:
        for l in (1, 2, 4):
            if not (was_frozen >> (l - 1)) & 1:
                unfreeze_layer_(ct, l)
        if isinstance(sr, SequenceReg):
            for l in (1, 2, 3, 4):
                if not (reg_frozen >> (l - 1)) & 1:
                    unfreeze_reg_(sr, l)
        mf.X[...] = X0
        mf.Y[...] = Y0
    if history is not None:
        history_(history, h, name="init_mu")


def init_logsigma_(model, capacity=int(10e8), history=None):
    """init_logsigma! (src/fit.jl:125-148): logsigma <- log sqrt( link_col_sqerr / column_nonnan ) with X'Y = 0."""
    st = _device_stats(model, use_factors=False)
    with np.errstate(divide="ignore", invalid="ignore"):
        col_vars = st["sqerr"].astype(np.float64) / st["n"].astype(np.float64)
        model.matfac.col_transform.unwrapped(1).logsigma[...] = np.log(np.sqrt(col_vars))
    history_(history, name="init_logsigma")


def reweight_col_losses_(model, capacity=int(10e8), history=None):
    """reweight_col_losses! (src/fit.jl:151-187): weights <- 1 / (rms column gradient * sigma), non-finite -> 1."""
    M, N = model.data.shape
    nm = model.matfac.noise_model
    nm.set_weight_(np.ones(N, np.float32))                                    # :157
    st = _device_stats(model, use_factors=False)                              # X, Y zeroed (:160-163)
    with np.errstate(divide="ignore", invalid="ignore"):
        rms = np.sqrt(st["ssq_grad"].astype(np.float64) / M)                  # :170 divides by M, not by the count
        rms *= np.exp(model.matfac.col_transform.unwrapped(1).logsigma)       # :175
        w = 1.0 / rms
    w[~np.isfinite(w)] = 1                                                    # :177
    nm.set_weight_(w.astype(np.float32))                                      # :180
    history_(history, name="reweight_col_losses")


def construct_minimal_regularizer(model, capacity=10 ** 8):
    """src/regularizers.jl:750-774: GroupRegularizer over the noise-model column ranges, weight
    K*mean(sigma^2) / (sum(nanvar .* nonnan) / M) per group (nanvar floored at 1/M)."""
    mf = model.matfac
    K, M = mf.X.shape
    st = _device_stats(model, use_factors=False)
    n = st["n"].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = st["sum"] / n
        var = (st["sumsq"].astype(np.float64) - n * mean * mean) / np.maximum(n - 1, 1)   # unbiased, like Julia's var
    var = np.where(np.isfinite(var), var, 0.0)
    var = np.maximum(var, 1.0 / M)                                            # :767
    sigma = np.exp(mf.col_transform.unwrapped(1).logsigma)
    nm = mf.noise_model
    weights = []
    for gidx in nm.col_ranges:
        sl = gidx.slice0()
        gw = K * np.mean(sigma[sl] ** 2) / (np.sum(var[sl] * n[sl]) / M)      # :768
        weights.append(np.full(K, gw, dtype=np.float32))
    return GroupRegularizer(group_idx=[(g.start, g.stop) for g in nm.col_ranges], group_weights=weights,
                            labels=list(nm.noises))


# ---- batch effects (src/fit.jl:297-496) -----------------------------------------------------------------
def theta_mom(theta_values):  # :297-301
    return [v.mean(axis=1, keepdims=True) for v in theta_values], [v.var(axis=1, ddof=1, keepdims=True) for v in theta_values]


def delta2_mom(delta2_values):  # :303-311
    mean = [v.mean(axis=1, keepdims=True) for v in delta2_values]
    var = [v.var(axis=1, ddof=1, keepdims=True) for v in delta2_values]
    alpha = [2.0 + (m * m) / (v + 1e-9) for m, v in zip(mean, var)]
    beta = [m * (a - 1.0) for m, a in zip(mean, alpha)]
    return alpha, beta


def _nans_to_val(arrs, val):  # :313-318
    for a in arrs:
        a[~np.isfinite(a)] = val


def theta_delta_em(model, delta2, sigma2, update_priors=True, batch_em_max_iter=100, batch_em_rtol=1e-8,
                   verbosity=1, print_prefix="", history=None):
    """theta_delta_em (src/fit.jl:326-375).  `model` is the working PathMatFacModel; the per-(batch, column) sums
    (ba_map) are computed on the device."""
    with np.errstate(divide="ignore", invalid="ignore"):
        theta = model.matfac.col_transform.unwrapped(4).theta
        theta_lsq = [v.copy() for v in theta.values]
        st = _device_stats(model, use_factors=True)
        batch_sizes = [c.astype(np.float64) for c in st["batch_count"]]                 # :332
        sig2 = [sigma2[cr.slice0()][None, :] for cr in theta.col_ranges]
        diffs = []
        theta_mean = theta_var = alpha = beta = None
        for it in range(1, batch_em_max_iter + 1):
            if update_priors or it == 1:
                theta_mean, theta_var = theta_mom(theta.values)                          # :344
                alpha, beta = delta2_mom(delta2)                                         # :345
            theta_old = [v.copy() for v in theta.values]
            new_theta = [(e * d2 * s2 + lsq * bs * v) / (s2 * d2 + bs * v)               # :350
                         for e, v, d2, lsq, bs, s2 in zip(theta_mean, theta_var, delta2, theta_lsq, batch_sizes, sig2)]
            _nans_to_val(new_theta, 0.0)                                                 # :352
            for dst, src in zip(theta.values, new_theta):
                dst[...] = src
            st = _device_stats(model, use_factors=True)                                  # :355 ba_map(sqerr_func)
            sq = [s.astype(np.float64) for s in st["batch_sqerr"]]
            _nans_to_val(sq, 0.0)                                                        # :357
            delta2 = [(b + 0.5 * (s / s2)) / (a + 0.5 * bs - 1.0)                        # :359
                      for a, b, s, bs, s2 in zip(alpha, beta, sq, batch_sizes, sig2)]
            _nans_to_val(delta2, 1.0)                                                    # :361
            num = sum(np.sum((t - o) ** 2) for t, o in zip(theta.values, theta_old))
            den = sum(np.sum(t * t) for t in theta.values)
            diff = num / den                                                             # :363
            diffs.append(diff)
            if verbosity > 0:
                print(f"{print_prefix}({it}) ||theta - theta'||^2/||theta||^2 : {diff}")
            if diff < batch_em_rtol:                                                     # :367
                break
    history_(history, name="batch_effect_EM_procedure", diffs=diffs)
    return [v.copy() for v in theta.values], delta2


def init_batch_effects_(model, capacity=10 ** 8, max_epochs=5000, lr_regress=0.25, lr_mu=0.1, lr_theta=1.0,
                        batch_method="EM", batch_em_rtol=1e-8, batch_em_max_iter=100, verbosity=1, print_prefix="",
                        history=None, **kwargs):
    """init_batch_effects! (src/fit.jl:378-496)."""
    n_pref = print_prefix + "    "
    orig = model.matfac
    work = copy.deepcopy(orig)                                                           # :393
    model.matfac = work
    try:
        cond = ids_to_ind_mat(model.sample_conditions)                                   # :397
        M, Kc = cond.shape
        N = work.Y.shape[1]
        work.X = np.asfortranarray(cond.T.astype(np.float32))                            # :400-401
        work.Y = np.zeros((Kc, N), dtype=np.float32, order="F")                          # :402
        init_mu_(model, capacity=capacity, max_epochs=max_epochs, lr_mu=lr_mu, verbosity=verbosity - 1,
                 print_prefix=n_pref, history=history)                                   # :407
        orig.col_transform.unwrapped(3).mu[...] = work.col_transform.unwrapped(3).mu     # :410
        work.X_reg = ZeroReg()                 # (never evaluated: X is not updated; its shape no longer matches)
        work.Y_reg = ZeroReg()                                                           # :415
        mf_fit_adapt_lr_(model, capacity=capacity, max_epochs=max_epochs, lr=lr_regress, min_lr=0.05,
                         verbosity=verbosity - 1, print_prefix=n_pref, update_Y=True, history=history)   # :416-428
        history_(history, name="regress_against_sample_conditions")
        work.col_transform_reg = None                                                    # :432  l -> 0
        init_theta_(model, capacity=capacity, max_epochs=max_epochs, lr_theta=lr_theta, verbosity=verbosity - 1,
                    print_prefix=n_pref, history=history)                                # :435
        theta_ba = work.col_transform.unwrapped(4).theta
        _nans_to_val(theta_ba.values, 0.0)                                               # :438
        with np.errstate(divide="ignore", invalid="ignore"):
            st = _device_stats(model, use_factors=True)
            col_vars = st["sqerr"].astype(np.float64) / st["n"].astype(np.float64)       # :444-447
            col_vars[~np.isfinite(col_vars)] = 1                                         # :449
            ba_vars = [s.astype(np.float64) / c.astype(np.float64) for s, c in zip(st["batch_sqerr"], st["batch_count"])]
            _nans_to_val(ba_vars, 1.0)                                                   # :458
            delta2 = [v / col_vars[cr.slice0()][None, :] for v, cr in zip(ba_vars, theta_ba.col_ranges)]  # :462
        theta_values = [v.copy() for v in theta_ba.values]
        if batch_method in ("EM", "EB"):                                                 # :466-481
            theta_values, delta2 = theta_delta_em(model, delta2, col_vars, update_priors=(batch_method == "EM"),
                                                  batch_em_max_iter=batch_em_max_iter, batch_em_rtol=batch_em_rtol,
                                                  print_prefix=n_pref, verbosity=verbosity - 1, history=history)
    finally:
        model.matfac = orig                                                              # :487
    with np.errstate(divide="ignore", invalid="ignore"):
        orig.col_transform.unwrapped(1).logsigma[...] = np.log(np.sqrt(col_vars))        # :491
        for dst, d2 in zip(orig.col_transform.unwrapped(2).logdelta.values, delta2):
            dst[...] = np.log(np.sqrt(d2))                                               # :493
    for dst, th in zip(orig.col_transform.unwrapped(4).theta.values, theta_values):
        dst[...] = th                                                                    # :494


# ---- post-processing (src/fit.jl:504-555) ---------------------------------------------------------------
def rms(X, axis):
    return np.sqrt(np.mean(X * X, axis=axis, keepdims=True))


def whiten_(model):
    """whiten! (src/fit.jl:504-527)."""
    mf = model.matfac
    X_rms = rms(mf.X.astype(np.float64), 1)
    mf.X[...] = mf.X / X_rms
    mf.Y[...] = mf.Y * X_rms
    ls = mf.col_transform.unwrapped(1).logsigma
    for cr in ids_to_ranges(model.feature_views):
        sl = cr.slice0()
        y_rms_max = float(np.max(rms(mf.Y[:, sl].astype(np.float64), 1)))
        if y_rms_max > 0:
            mf.Y[:, sl] = mf.Y[:, sl] / y_rms_max
            ls[sl] += np.log(y_rms_max)
        else:
            mf.Y[:, sl] = 0
            ls[sl] = np.float32(-1e9)


def rotate_by_svd_(model):
    """rotate_by_svd! (src/fit.jl:530-543): Y <- S*Vt, X' <- X' * U."""
    mf = model.matfac
    U, s, Vt = np.linalg.svd(mf.Y.astype(np.float64), full_matrices=False)
    mf.Y[...] = s[:, None] * Vt
    mf.X[...] = (mf.X.astype(np.float64).T @ U).T


def reorder_by_importance_(model):
    """reorder_by_importance! (src/fit.jl:546-555)."""
    mf = model.matfac
    idx = np.argsort(-np.sum(mf.Y.astype(np.float64) ** 2, axis=1), kind="stable")
    mf.X[...] = mf.X[idx, :]
    mf.Y[...] = mf.Y[idx, :]
    for reg in (mf.Y_reg, mf.X_reg):
        _reorder_reg(reg, idx)


def _reorder_reg(reg, p):  # reorder_reg! (regularizers.jl:5, 53-55, 449-452; featureset_ard.jl:68-81)
    if isinstance(reg, L2Regularizer):
        reg.weights[...] = reg.weights[p]
    elif isinstance(reg, GroupRegularizer):
        reg.group_weights = tuple(w[p] for w in reg.group_weights)
    elif isinstance(reg, FeatureSetARDReg):
        reg.beta[...] = reg.beta[p, :]
        for A in reg.A:
            A[...] = A[:, p]
        reg.lambda_ = tuple(l[p] for l in reg.lambda_)
        if getattr(reg, "ssq_grad", None) is not None:
            reg.ssq_grad = tuple(g[:, p] for g in reg.ssq_grad)
    elif isinstance(reg, _R.CompositeRegularizer):
        for r in reg.regularizers:
            _reorder_reg(r, p)


def reweight_eb_(reg, P, mixture_p=1.0):
    """reweight_eb! for the regularizers in scope (regularizers.jl:39-47, 406-420, 588-609, 634-638): weights
    <- mixture_p / (largest squared singular value) of the (group's) parameter block."""
    if isinstance(reg, L2Regularizer):
        s = np.linalg.svd(P.astype(np.float64), compute_uv=False)
        reg.weights[...] = mixture_p / s[0] ** 2
    elif isinstance(reg, GroupRegularizer):
        K = P.shape[0]
        new = []
        for g in reg.group_idx:
            s = np.linalg.svd(P[:, g.slice0()].astype(np.float64), compute_uv=False)
            new.append(np.full(K, mixture_p / s[0] ** 2, dtype=np.float32))
        reg.group_weights = tuple(new)
    elif isinstance(reg, ARDRegularizer):
        reg.reweight_eb_(P)
    elif isinstance(reg, _R.CompositeRegularizer):
        for r, p in zip(reg.regularizers, reg.mixture_p):
            reweight_eb_(r, P, mixture_p=p * mixture_p)
    elif isinstance(reg, _R.ColParamReg):
        # regularizers.jl:490-497 on the layer's vector (logsigma of ColScale / mu of ColShift, :510-519): per view range,
        # centre = mean, weight = p * (0.1 + 0.5) / (0.1 + 0.5 var), Julia's `var` = sample variance (n - 1)
        v = np.asarray(P.logsigma if isinstance(P, ColScale) else P.mu if isinstance(P, ColShift) else P, dtype=np.float64)
        centers, weights = [], []
        for r in reg.col_ranges:
            x = v[r.slice0()]
            var = float(np.var(x, ddof=1)) if x.size > 1 else float("nan")
            centers.append(float(np.mean(x)))
            weights.append(float(np.float32(mixture_p) * np.float32(0.1 + 0.5) / (np.float32(0.1) + np.float32(0.5) * np.float32(var))))
        reg.centers, reg.weights = tuple(centers), tuple(weights)
    elif isinstance(reg, _R.BatchArrayReg):
        # regularizers.jl:818-838 on the layer's BatchArray (logdelta of BatchScale / theta of BatchShift, :868-875): per
        # view and row batch, centre = mean over the view's columns, weight = p / var; a non-finite weight (one column, or
        # zero variance) becomes 1 + 0.5 * (#columns): the posterior mean of a Gamma(1, 1) precision
        ba = P.logdelta if isinstance(P, BatchScale) else P.theta if isinstance(P, BatchShift) else P
        centers, weights = [], []
        for vals, cr in zip(ba.values, ba.col_ranges):
            x = np.asarray(vals, dtype=np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                var = np.var(x, axis=1, ddof=1) if x.shape[1] > 1 else np.full(x.shape[0], np.nan)
                w = (mixture_p / var).astype(np.float32)
            w[~np.isfinite(w)] = np.float32(1 + 0.5 * len(cr))
            centers.append(np.mean(x, axis=1).astype(np.float32))
            weights.append(w)
        reg.centers, reg.weights = tuple(centers), tuple(weights)
    elif isinstance(reg, SequenceReg):
        # regularizers.jl:928-932: zip over (regs, layers); P is the ViewableComposition
        for i, r in enumerate(reg.regs):
            reweight_eb_(r.reg if isinstance(r, _R.FrozenRegularizer) else r, P.unwrapped(i + 1), mixture_p=mixture_p)
    # pure functions (x -> 0) and FeatureSetARDReg have no adjustable weights (regularizers.jl:941-943)


# ---- stage drivers (src/fit.jl:564-892) -----------------------------------------------------------------
def basic_fit_(model, fit_batch=False, batch_method="EM", fit_mu=False, fit_logsigma=False, reweight_losses=False,
               init_factors=False, init_factors_method="adagrad", fit_factors=False, init_ordinal=False,
               svd_rotate=False, whiten=False, capacity=int(10e8), lr=1.0, max_epochs=1000, verbosity=1,
               print_prefix="", history=None, lr_regress=1.0, lr_mu=0.1, lr_theta=1.0, **kwargs):
    """basic_fit! (src/fit.jl:564-671)."""
    n_prefix = print_prefix + "    "
    ct = model.matfac.col_transform
    if init_ordinal:
        raise NotImplementedError("ordinal noise models are out of scope")
    if fit_batch:
        assert isinstance(ct.unwrapped(2), BatchScale) and isinstance(ct.unwrapped(4), BatchShift), \
            "Model must have batch parameters whenever `fit_batch` is true"
        init_batch_effects_(model, batch_method=batch_method, capacity=capacity, max_epochs=max_epochs,
                            verbosity=verbosity, print_prefix=n_prefix, history=history, lr_regress=lr_regress,
                            lr_theta=lr_theta)
    else:
        if fit_mu:
            init_mu_(model, capacity=capacity, max_epochs=500, verbosity=verbosity, print_prefix=n_prefix,
                     history=history)
        if fit_logsigma:
            init_logsigma_(model, capacity=capacity)
    if reweight_losses:
        reweight_col_losses_(model, capacity=capacity)
    if init_factors:
        init_factors_(model, lr=lr, init_factors_method=init_factors_method, verbosity=verbosity,
                      print_prefix=print_prefix, history=history, capacity=capacity, max_epochs=max_epochs, **kwargs)
    if fit_factors:
        mf_fit_adapt_lr_(model, capacity=capacity, update_X=True, update_Y=True, lr=lr, min_lr=0.05,
                         max_epochs=max_epochs, verbosity=verbosity, print_prefix=n_prefix, history=history, **kwargs)
    if whiten:
        whiten_(model)
    if svd_rotate:
        rotate_by_svd_(model)
    unfreeze_layer_(ct, [1, 2, 3, 4])


def fit_ard_(model, max_epochs=1000, capacity=10 ** 8, verbosity=1, print_prefix="", history=None, lr=1.0,
             lr_regress=1.0, lr_theta=1.0, svd_rotate=True, batch_method="EM", **kwargs):
    """fit_ard! (src/fit.jl:751-808)."""
    n_pref = print_prefix + "    "
    mf = model.matfac
    orig_X_reg, orig_ard = mf.X_reg, mf.Y_reg
    mf.X_reg = ZeroReg()                                                     # :769
    mf.Y_reg = construct_minimal_regularizer(model)                           # :770
    fit_batch = isinstance(mf.col_transform.unwrapped(2), BatchScale)
    basic_fit_(model, fit_batch=fit_batch, fit_mu=True, fit_logsigma=True, init_factors=True, reweight_losses=True,
               svd_rotate=svd_rotate, whiten=True, lr_regress=lr_regress, lr_theta=lr_theta, verbosity=verbosity,
               print_prefix=n_pref, batch_method=batch_method, max_epochs=max_epochs, capacity=capacity,
               history=history, lr=lr, **kwargs)                              # :773-787
    mf.X_reg = orig_X_reg
    reweight_eb_(mf.X_reg, mf.X)                                              # :791-792
    mf.Y_reg = orig_ard
    reweight_eb_(mf.Y_reg, mf.Y)                                              # :793-794
    reweight_col_losses_(model, capacity=capacity)                            # :797
    mf_fit_adapt_lr_(model, capacity=capacity, update_X=True, update_Y=True, lr=lr, min_lr=0.01,
                     max_epochs=max_epochs, verbosity=verbosity, print_prefix=n_pref, history=history, **kwargs)  # :800


def fit_non_ard_(model, fit_reg_weight="EB", **kwargs):
    """fit_non_ard! (src/fit.jl:730-745), without the empirical-Bayes re-fit (basic_fit_reg_weight_eb!)."""
    kwargs.pop("lambda_max", None); kwargs.pop("n_lambda", None); kwargs.pop("lambda_min_frac", None)
    fit_batch = isinstance(model.matfac.col_transform.unwrapped(2), BatchScale)
    if fit_reg_weight == "EB":
        basic_fit_reg_weight_eb_(model, **kwargs)
    else:
        basic_fit_(model, fit_mu=True, fit_logsigma=True, reweight_losses=True, fit_batch=fit_batch,
                   fit_factors=True, **kwargs)


def basic_fit_reg_weight_eb_(model, capacity=int(10e8), lr=1.0, max_epochs=1000, verbosity=1, print_prefix="",
                             history=None, svd_rotate=True, **kwargs):
    """basic_fit_reg_weight_eb! (src/fit.jl:674-727)."""
    n_pref = print_prefix + "    "
    mf = model.matfac
    sr = mf.col_transform_reg
    if isinstance(sr, SequenceReg):
        freeze_reg_(sr, [1, 2, 3, 4])                                          # :681
    orig_X_reg, orig_Y_reg = mf.X_reg, mf.Y_reg
    mf.X_reg = L2Regularizer(mf.X.shape[0], 1.0)                              # X -> 0.5*sum(X.*X)   (:686)
    mf.Y_reg = construct_minimal_regularizer(model)                            # :687
    fit_batch = isinstance(mf.col_transform.unwrapped(2), BatchScale)
    basic_fit_(model, fit_mu=True, fit_logsigma=True, reweight_losses=True, fit_batch=fit_batch, init_factors=True,
               svd_rotate=svd_rotate, whiten=False, verbosity=verbosity, print_prefix=n_pref, capacity=capacity,
               lr=lr, max_epochs=max_epochs, history=history, **kwargs)       # :694-704
    if isinstance(sr, SequenceReg):
        unfreeze_reg_(sr, [1, 2, 3, 4])                                        # :709
    mf.X_reg, mf.Y_reg = orig_X_reg, orig_Y_reg
    if isinstance(sr, SequenceReg):
        reweight_eb_(sr, mf.col_transform)                                     # :712
    reweight_eb_(mf.X_reg, mf.X)                                               # :713
    reweight_eb_(mf.Y_reg, mf.Y)                                               # :714
    history_(history, name="reweight_eb")
    basic_fit_(model, reweight_losses=True, fit_factors=True, verbosity=verbosity, print_prefix=n_pref,
               history=history, capacity=capacity, lr=lr, max_epochs=max_epochs, **kwargs)   # :720-725


def fit_feature_set_ard_(model, lr=1.0, capacity=10 ** 8, max_epochs=1000, fsard_max_iter=10, fsard_max_A_iter=1000,
                         fsard_term_rtol=1e-5, verbosity=1, print_prefix="", svd_rotate=True, history=None, **kwargs):
    """fit_feature_set_ard! (src/fit.jl:814-892)."""
    from .featureset_ard import update_A_
    n_pref = print_prefix + "    "
    mf = model.matfac
    orig_reg = mf.Y_reg
    mf.Y_reg = ARDRegularizer(model.feature_views)                            # :829
    fit_ard_(model, max_epochs=max_epochs, capacity=capacity, lr=lr, verbosity=verbosity, print_prefix=n_pref,
             history=history, svd_rotate=svd_rotate, **kwargs)                 # :832
    mf.Y_reg = orig_reg                                                        # :837
    beta_old = orig_reg.beta.copy()
    for it in range(1, fsard_max_iter + 1):
        update_A_(orig_reg, mf.Y, max_epochs=fsard_max_A_iter, term_iter=50, print_prefix=n_pref, print_iter=100,
                  verbosity=verbosity, ctx=model.device_context())             # :850 (the ISTA loop runs on the device)
        d = beta_old - orig_reg.beta
        beta_diff = float(np.sum(d * d) / np.sum(orig_reg.beta * orig_reg.beta))   # :856-858
        if verbosity > 0:
            print(f"{n_pref}### (dB)^2/(B)^2 = {beta_diff} ###")
        if beta_diff < fsard_term_rtol:                                        # :863
            break
        beta_old[...] = orig_reg.beta
        if it == fsard_max_iter:                                               # :872
            break
        mf_fit_adapt_lr_(model, capacity=capacity, update_X=True, update_Y=True, lr=lr, min_lr=0.01,
                         max_epochs=max_epochs, verbosity=verbosity, print_prefix=n_pref + "    ", history=history)  # :883


def fit_(model, lr=1.0, fit_reg_weight="EB", n_lambda=8, lambda_max=None, lambda_min_frac=1e-3, keep_history=False,
         svd_rotate=True, fit_joint=False, fsard_max_iter=10, fsard_max_A_iter=1000, fsard_term_rtol=1e-5,
         rel_tol=1e-5, abs_tol=1e-5, verbosity=1, print_prefix="", capacity=10 ** 8, **kwargs):
    """fit! (src/fit.jl:923-1018), the master function.  Returns the history list (or None)."""
    global FIT_START_TIME
    FIT_START_TIME = time.time()
    hist = [] if keep_history else None
    history_(hist, name="start")
    mf = model.matfac
    if isinstance(mf.Y_reg, ARDRegularizer):
        fit_ard_(model, history=hist, verbosity=verbosity, print_prefix=print_prefix, rel_tol=rel_tol, abs_tol=abs_tol,
                 capacity=capacity, svd_rotate=svd_rotate, lr=lr, **kwargs)
    elif isinstance(mf.Y_reg, FeatureSetARDReg):
        fit_feature_set_ard_(model, lr=lr, rel_tol=rel_tol, abs_tol=abs_tol, history=hist,
                             fsard_max_iter=fsard_max_iter, fsard_max_A_iter=fsard_max_A_iter,
                             fsard_term_rtol=fsard_term_rtol, svd_rotate=svd_rotate, verbosity=verbosity,
                             print_prefix=print_prefix, capacity=capacity, **kwargs)
    else:
        fit_non_ard_(model, history=hist, rel_tol=rel_tol, abs_tol=abs_tol, fit_reg_weight=fit_reg_weight,
                     lambda_max=lambda_max, n_lambda=n_lambda, lambda_min_frac=lambda_min_frac, svd_rotate=svd_rotate,
                     verbosity=verbosity, print_prefix=print_prefix, capacity=capacity, lr=lr, **kwargs)
    if fit_joint:
        # the reference's fit_joint branch references an undefined `max_epochs` (Q3, src/fit.jl:995): it cannot run there
        raise NotImplementedError("fit_joint=true is broken in the reference (src/fit.jl:995) and not reproduced")
    whiten_(model)                                                             # :1004
    reweight_col_losses_(model, capacity=capacity, history=hist)               # :1008
    reorder_by_importance_(model)                                              # :1011
    history_(hist, name="reorder_factors")
    history_(hist, name="finish")
    return hist
