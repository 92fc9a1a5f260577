"""FeatureSetARD outer loop, host side: update_lambda! (src/featureset_ard.jl:189-209, a K-vector per view from
row means of Y) and the per-view driver of update_A! (:278-294).  The ISTA iterations themselves -- gamma_normal_loss and
its pull-back (:154-186), ISTAOptimiser.update! (src/optimizers.jl:26-62), the bookkeeping of update_A_inner! (:214-276) --
run on the device inside libpmf_hip.so (pmf_fsard_update_A, csrc/pmf_fsard.hip: two small HIP kernels per iteration, no
torch, no BLAS library).  There is no CPU fallback: without the library and a GPU the call raises."""
import numpy as np


def update_lambda_(reg, Y):
    """update_lambda! (featureset_ard.jl:189-209)."""
    new = []
    for cr, S, A in zip(reg.col_ranges, reg.S, reg.A):
        Yv = np.asarray(Y[:, cr.slice0()], dtype=np.float64)
        Y_ms = np.mean(Yv * Yv, axis=1)
        min_ms = min(float(Y_ms.min()), float(reg.v0))
        den = Y_ms - min_ms + 1e-3
        new.append(((A.shape[0] * float(np.mean(S))) / den).astype(np.float32))
    reg.lambda_ = tuple(new)


def update_A_(reg, Y, max_epochs=1000, term_iter=20, atol=1e-5, verbosity=1, print_prefix="", print_iter=100, ctx=None):
    """update_A! (featureset_ard.jl:278-294): per view, A <- 0, ISTA fit, then beta[:, cr] = beta0*(v0 + A'S) -- on the
    device (pmf_fsard_update_A: the whole ISTA loop runs in two small HIP kernels per iteration, no torch).  `ctx`: the
    model's context (its resident Y is refreshed from `Y`, K x N floats); without one a scratch context is made."""
    from ._lib import Context
    K, N = Y.shape
    own = ctx is None
    if own:
        ctx = Context(0)
        ctx.set_data(np.full((1, N), np.nan, dtype=np.float32))
        ctx.set_factors(np.zeros((K, 1), np.float32), Y)
    else:
        ctx.set_Y(Y)
    if getattr(reg, "ssq_grad", None) is None:              # ISTAOptimiser state persists across calls (optimizers.jl:34-37)
        reg.ssq_grad = tuple(np.full(A.shape, 1e-8, dtype=np.float32) for A in reg.A)
    losses = []
    try:
        for v, (cr, A, S) in enumerate(zip(reg.col_ranges, reg.A, reg.S)):
            sl = cr.slice0()
            ssq = np.ascontiguousarray(reg.ssq_grad[v], dtype=np.float32)
            A_new, beta, best, _ = ctx.fsard_update_A(cr.start, cr.stop, S, reg.alpha[sl], reg.lambda_[v], float(reg.alpha0),
                                                      float(reg.v0), float(reg.lr), ssq, max_epochs=max_epochs,
                                                      term_iter=term_iter, atol=atol)
            losses.append(best)
            A[...] = A_new
            reg.ssq_grad[v][...] = ssq
            reg.beta[:, sl] = beta                                                           # :292
            if verbosity > 1:
                print(f"{print_prefix}    View {v + 1}: final loss {best}")
    finally:
        if own:
            ctx.close()
    return losses
