"""Independent CPU restatement of the batch-effect EM of PathMatFac -- TEST INFRASTRUCTURE ONLY (never imported by the
product; see oracle/pmf_oracle.py for the rules).

Follows /root/reference/src/fit.jl:
  theta_mom        :297-301   per (view, batch): mean and sample variance of theta over the view's columns
  delta2_mom       :303-311   inverse-gamma moments: alpha = 2 + m^2 / (v + 1e-9), beta = m (alpha - 1)
  theta_delta_em   :326-375   the EM loop (theta update :350, delta^2 update :359, stopping rule :363-369)
  ba_map           src/batch_array.jl:320-334 (per (batch, column) sums of a function of (model, data))
MF.sqerr_func is un-vendored (MatFac.jl): self-specified, as in DESIGN.md section 2, as (invlink(Z) - D)^2 on finite
entries -- Z itself for normal columns, the logistic function for Bernoulli, exp for Poisson.
Dense float64 numpy throughout: no statistics kernel, no device, no shared code with pathmatfac.jl_amd/fit.py.
"""
import numpy as np


def forward(X, Y, logsigma, mu, views):
    """Z = layers(X'Y): ColScale, BatchScale, ColShift, BatchShift (src/layers.jl:20-22, 120-122, 64-66, 199-201)."""
    Z = X.astype(np.float64).T @ Y.astype(np.float64)
    Z *= np.exp(np.asarray(logsigma, np.float64))[None, :]
    for v in views:
        sl = slice(v["start1"] - 1, v["stop1"])
        Z[:, sl] *= np.exp(np.asarray(v["logdelta"], np.float64))[v["batch_of_row"], :]
    Z += np.asarray(mu, np.float64)[None, :]
    for v in views:
        sl = slice(v["start1"] - 1, v["stop1"])
        Z[:, sl] += np.asarray(v["theta"], np.float64)[v["batch_of_row"], :]
    return Z


def ba_map_sums(values_per_entry, views, mask):
    """Per view: (n_batches x N_v) sums of `values_per_entry` (M x N) over the rows of each batch, masked entries only."""
    out = []
    for v in views:
        sl = slice(v["start1"] - 1, v["stop1"])
        nb = np.asarray(v["theta"]).shape[0]
        bor = np.asarray(v["batch_of_row"])
        vals = np.where(mask[:, sl], values_per_entry[:, sl], 0.0)
        S = np.zeros((nb, sl.stop - sl.start))
        for b in range(nb):
            S[b] = vals[bor == b].sum(axis=0)
        out.append(S)
    return out


def sqerr(Z, D, kind_of_col):
    pred = Z.copy()
    for j, k in enumerate(kind_of_col):
        if k == "bernoulli":
            pred[:, j] = 1.0 / (1.0 + np.exp(-Z[:, j]))
        elif k == "poisson":
            pred[:, j] = np.exp(Z[:, j])
    return (pred - D) ** 2


def theta_mom(theta_values):
    return ([np.mean(v, axis=1, keepdims=True) for v in theta_values],
            [np.var(v, axis=1, ddof=1, keepdims=True) if v.shape[1] > 1 else np.full((v.shape[0], 1), np.nan) for v in theta_values])


def delta2_mom(delta2):
    m = [np.mean(v, axis=1, keepdims=True) for v in delta2]
    s = [np.var(v, axis=1, ddof=1, keepdims=True) if v.shape[1] > 1 else np.full((v.shape[0], 1), np.nan) for v in delta2]
    alpha = [2.0 + (mm * mm) / (ss + 1e-9) for mm, ss in zip(m, s)]
    beta = [mm * (a - 1.0) for mm, a in zip(m, alpha)]
    return alpha, beta


def theta_delta_em(D, X, Y, logsigma, mu, views, kind_of_col, delta2, sigma2, update_priors=True, max_iter=100, rtol=1e-8):
    """Returns (theta values per view, delta2 per view, diffs).  `views[v]["theta"]` are updated in place like the
    reference's model.col_transform.layers[4].theta.values."""
    D = np.asarray(D, np.float64)
    finite = np.isfinite(D)
    views = [dict(v, theta=np.array(v["theta"], np.float64), logdelta=np.array(v["logdelta"], np.float64)) for v in views]
    theta_lsq = [v["theta"].copy() for v in views]
    batch_sizes = ba_map_sums(np.ones_like(D), views, finite)
    delta2 = [np.array(d, np.float64) for d in delta2]
    sig2 = [np.asarray(sigma2, np.float64)[v["start1"] - 1:v["stop1"]][None, :] for v in views]
    diffs = []
    theta_mean = theta_var = alpha = beta = None
    with np.errstate(divide="ignore", invalid="ignore"):
        for it in range(1, max_iter + 1):
            if update_priors or it == 1:
                theta_mean, theta_var = theta_mom([v["theta"] for v in views])
                alpha, beta = delta2_mom(delta2)
            theta_old = [v["theta"].copy() for v in views]
            for v, e, var, d2, lsq, bs, s2 in zip(views, theta_mean, theta_var, delta2, theta_lsq, batch_sizes, sig2):
                t = (e * d2 * s2 + lsq * bs * var) / (s2 * d2 + bs * var)
                t[~np.isfinite(t)] = 0.0
                v["theta"] = t
            Z = forward(X, Y, logsigma, mu, views)
            sq = ba_map_sums(sqerr(Z, np.where(finite, D, 0.0), kind_of_col), views, finite)
            for q in sq:
                q[~np.isfinite(q)] = 0.0
            delta2 = [(b + 0.5 * (s / s2)) / (a + 0.5 * bs - 1.0) for a, b, s, bs, s2 in zip(alpha, beta, sq, batch_sizes, sig2)]
            for d in delta2:
                d[~np.isfinite(d)] = 1.0
            num = sum(np.sum((v["theta"] - o) ** 2) for v, o in zip(views, theta_old))
            den = sum(np.sum(v["theta"] ** 2) for v in views)
            diffs.append(num / den)
            if diffs[-1] < rtol:
                break
    return [v["theta"] for v in views], delta2, diffs
