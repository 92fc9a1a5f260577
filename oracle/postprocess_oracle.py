"""Independent numpy restatement of the reference's post-processing of a fitted model  --  TEST INFRASTRUCTURE ONLY.

    whiten!                (src/fit.jl:504-527)
    rotate_by_svd!         (src/fit.jl:530-543)
    reorder_by_importance! (src/fit.jl:546-555) with reorder_reg! for L2 / Group / FeatureSetARD
                           (src/regularizers.jl:53-55, 449-452; src/featureset_ard.jl:68-81)

It works on plain arrays in fp64 (X: K x M, Y: K x N, logsigma: N, view column ranges 1-based inclusive) and shares no code
with pathmatfac.jl_amd/fit.py: means are explicit loops over factors, the rotation goes through the eigen-decomposition of
Y Y' instead of an SVD routine, the ordering through Python's sorted().

PARITY STATUS: "parity unpinned" -- the reference's tests hold no known answers for these three functions
(test/runtests.jl calls fit! and asserts only that parameters changed).  What is fixed by the reference's TEXT and
checked here: the formulas, the -1e9 log-sigma of an all-zero view, descending singular values, the stable descending
order of reorder (Julia's sortperm(rev=true) is stable).  What the text does NOT fix is the sign of each singular vector
pair (LAPACK's choice inside `svd`): the product X'Y is invariant under a joint sign flip of row k of X and Y, so
comparisons with the product are made after aligning signs per factor (`align_signs`).
"""
import numpy as np


def rms_rows(A):
    """rms(A; dims=2) (src/util.jl:24-26): sqrt(mean(A .* A)) over each row."""
    out = np.empty(A.shape[0])
    for k in range(A.shape[0]):
        acc = 0.0
        for v in A[k]:
            acc += float(v) * float(v)
        out[k] = (acc / A.shape[1]) ** 0.5
    return out


def whiten(X, Y, logsigma, view_ranges):
    """Returns (X, Y, logsigma) after whiten!.  view_ranges: [(start1, stop1)] = ids_to_ranges(feature_views)."""
    X = np.array(X, dtype=np.float64)
    Y = np.array(Y, dtype=np.float64)
    ls = np.array(logsigma, dtype=np.float64)
    xr = rms_rows(X)                                   # :505
    for k in range(X.shape[0]):
        X[k] /= xr[k]                                  # :506
        Y[k] *= xr[k]                                  # :507
    for (s1, e1) in view_ranges:                       # :510-511
        cols = list(range(s1 - 1, e1))
        yr_max = max(rms_rows(Y[:, cols]))             # :513, :517
        if yr_max > 0:
            Y[:, cols] /= yr_max                       # :519
            ls[cols] += np.log(yr_max)                 # :520
        else:
            Y[:, cols] = 0.0                           # :522
            ls[cols] = float(np.float32(-1e9))         # :523
    return X, Y, ls


def rotate_by_svd(X, Y):
    """Y <- S * Vt, X' <- X' * U  with Y = U S Vt (thin).  Through the symmetric eigenproblem of Y Y' (K x K):
    Y Y' = U S^2 U', so U = eigenvectors (descending eigenvalues) and S Vt = U' Y."""
    X = np.array(X, dtype=np.float64)
    Y = np.array(Y, dtype=np.float64)
    w, U = np.linalg.eigh(Y @ Y.T)
    order = sorted(range(len(w)), key=lambda k: -w[k])
    U = U[:, order]
    Ynew = U.T @ Y                                     # = S * Vt            (:538)
    Xnew = (X.T @ U).T                                 # X' * U, transposed  (:541)
    return Xnew, Ynew


def importance_order(Y):
    """sortperm(vec(sum(Y .* Y, dims=2)), rev=true): 0-based, ties in original order (Julia's sortperm is stable)."""
    ssq = [sum(float(v) * float(v) for v in Y[k]) for k in range(Y.shape[0])]
    return sorted(range(len(ssq)), key=lambda k: (-ssq[k], k))


def reorder_by_importance(X, Y, l2_weights=None, group_weights=None, fsard=None):
    """Returns the reordered copies.  fsard = dict(beta K x N, A [L_v x K], lambda [K]) or None."""
    p = importance_order(np.asarray(Y, dtype=np.float64))
    out = {"order": p, "X": np.asarray(X)[p, :].copy(), "Y": np.asarray(Y)[p, :].copy()}
    if l2_weights is not None:
        out["l2_weights"] = np.asarray(l2_weights)[p].copy()                       # regularizers.jl:53-55
    if group_weights is not None:
        out["group_weights"] = [np.asarray(w)[p].copy() for w in group_weights]    # :449-452
    if fsard is not None:                                                          # featureset_ard.jl:68-81
        out["fsard"] = {"beta": np.asarray(fsard["beta"])[p, :].copy(),
                        "A": [np.asarray(A)[:, p].copy() for A in fsard["A"]],
                        "lambda": [np.asarray(l)[p].copy() for l in fsard["lambda"]]}
    return out


def align_signs(X, Y, X_ref, Y_ref):
    """Flip factor k of (X, Y) where that brings Y closer to Y_ref (the joint flip leaves X'Y unchanged)."""
    X, Y = np.array(X, dtype=np.float64), np.array(Y, dtype=np.float64)
    for k in range(Y.shape[0]):
        if np.dot(Y[k], np.asarray(Y_ref, dtype=np.float64)[k]) < 0:
            X[k] = -X[k]
            Y[k] = -Y[k]
    return X, Y
