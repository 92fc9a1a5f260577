"""FeatureSetARD outer loop: update_lambda!, update_A! with the projected-AdaGrad ISTA optimiser
(src/featureset_ard.jl:154-294, src/optimizers.jl:26-62).

These are K x N_v / L_v x K dense updates (<= 1000 iterations per view per outer iteration); they run as torch
tensor ops on the GPU (rocBLAS GEMMs for A'S and S*grad', elementwise kernels for the rest) so Y never leaves the
device-side precision path.  There is no CPU fallback: without a GPU torch device the call raises."""
import numpy as np


def _device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("update_A_ needs a GPU (torch.cuda is not available); there is no CPU fallback")
    return torch.device("cuda")


def gamma_normal_loss(A, S, alpha, alpha0, v0, Y):
    """featureset_ard.jl:154-162 (torch tensors)."""
    import torch
    beta0 = alpha0 - 1
    beta = beta0 * (v0 + A.T @ S)
    a5 = alpha + 0.5
    lss = -torch.sum(alpha[None, :] * torch.sum(torch.log(beta), dim=0, keepdim=True)) \
        + torch.sum(a5[None, :] * torch.sum(torch.log(beta + 0.5 * (Y * Y)), dim=0, keepdim=True))
    lss = lss - torch.sum((a5 * torch.log(a5) - alpha * torch.log(alpha))[None, :]
                          + torch.sum(torch.log(torch.abs(Y) + 1e-9), dim=0, keepdim=True))
    return lss


def gamma_normal_grad_A(A, S, alpha, alpha0, v0, Y):
    """The rrule's pull-back (featureset_ard.jl:164-178): grad_A = S * grad_AtS'."""
    beta0 = alpha0 - 1
    beta = beta0 * (v0 + A.T @ S)
    a5 = alpha + 0.5
    grad_AtS = beta0 * ((-alpha[None, :] / beta) + a5[None, :] / (beta + 0.5 * (Y * Y)))
    return S @ grad_AtS.T


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


def update_A_inner_(A, S, Yv, alpha, alpha0, v0, lr, lam, ssq_grad, max_epochs=1000, term_iter=20, atol=1e-5,
                    verbosity=1, print_prefix="", print_iter=100):
    """update_A_inner! (featureset_ard.jl:214-276) with ISTAOptimiser.update! (optimizers.jl:46-62). torch tensors."""
    import torch

    def total(Am):
        return gamma_normal_loss(Am, S, alpha, alpha0, v0, Yv) + torch.sum(lam[None, :] * torch.abs(Am))
    best = float(total(A))
    A_best = A.clone()
    term_count = 0
    for epoch in range(1, max_epochs + 1):
        g = gamma_normal_grad_A(A, S, alpha, alpha0, v0, Yv)
        ssq_grad += g * g                                  # optimizers.jl:50
        eta = lr / torch.sqrt(ssq_grad)                    # :51
        A -= eta * g                                       # :55
        A.clamp_(min=0)                                    # :56
        A.copy_(torch.clamp(torch.abs(A) - lam[None, :] * eta, min=0))   # ist_proj! :40-42, :61
        new = float(total(A))
        if new < best:
            diff = best - new
            best = new
            A_best.copy_(A)
            term_count = 0 if diff > atol else term_count + 1
        else:
            term_count += 1
        if verbosity > 1 and epoch % print_iter == 0:
            print(f"{print_prefix}Iteration {epoch}:\t Loss={new}")
        if term_count >= term_iter:
            break
    A.copy_(A_best)
    return best


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
