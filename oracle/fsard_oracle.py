"""numpy (float64) restatement of the FeatureSetARD outer-loop pieces -- TEST INFRASTRUCTURE ONLY.

Follows src/featureset_ard.jl:154-294 (gamma_normal_loss and its rrule, update_A_inner!, update_A!) and
src/optimizers.jl:26-62 (ISTAOptimiser).  Parity status: the reference's tests for these are smoke-only
(`@test true`, test/runtests.jl:889-935), so this restatement is pinned only by the closed forms it transcribes."""
import numpy as np


def gamma_normal_loss(A, S, alpha, alpha0, v0, Y):          # featureset_ard.jl:154-162
    beta0 = alpha0 - 1
    beta = beta0 * (v0 + A.T @ S)
    a5 = alpha + 0.5
    lss = -np.sum(alpha[None, :] * np.sum(np.log(beta), axis=0)) + np.sum(a5[None, :] * np.sum(np.log(beta + 0.5 * Y * Y), axis=0))
    lss -= np.sum((a5 * np.log(a5) - alpha * np.log(alpha))[None, :] + np.sum(np.log(np.abs(Y) + 1e-9), axis=0, keepdims=True))
    return lss


def grad_A(A, S, alpha, alpha0, v0, Y):                      # featureset_ard.jl:164-178
    beta0 = alpha0 - 1
    beta = beta0 * (v0 + A.T @ S)
    g = beta0 * ((-alpha[None, :] / beta) + (alpha + 0.5)[None, :] / (beta + 0.5 * Y * Y))
    return S @ g.T


def update_A_inner(A, S, Y, alpha, alpha0, v0, lr, lam, ssq_grad, max_epochs=1000, term_iter=20, atol=1e-5):
    """featureset_ard.jl:214-276 + optimizers.jl:46-62.  A, ssq_grad are updated in place; returns best loss."""
    def total(Am):
        return gamma_normal_loss(Am, S, alpha, alpha0, v0, Y) + np.sum(lam[None, :] * np.abs(Am))
    best = total(A)
    A_best = A.copy()
    term_count = 0
    for _ in range(max_epochs):
        g = grad_A(A, S, alpha, alpha0, v0, Y)
        ssq_grad += g * g
        eta = lr / np.sqrt(ssq_grad)
        A -= eta * g
        np.maximum(A, 0, out=A)
        A[...] = np.maximum(np.abs(A) - lam[None, :] * eta, 0)
        new = total(A)
        if new < best:
            diff = best - new
            best = new
            A_best[...] = A
            term_count = 0 if diff > atol else term_count + 1
        else:
            term_count += 1
        if term_count >= term_iter:
            break
    A[...] = A_best
    return best
