"""Seeded synthetic problem instances shared by the oracle-side and HIP-side tests.

A problem is a plain dict of numpy arrays in the reference's conventions (column-major semantics,
1-based inclusive ranges).  `to_oracle` feeds it to the C oracle, `to_context` to libpmf_hip.so through
the ctypes binding, so every parity test runs the SAME inputs through both.
Distributions follow src/simulate_params.jl.
"""
import numpy as np


def split_ranges(n, parts):
    """n items into `parts` contiguous 1-based inclusive ranges."""
    edges = np.linspace(0, n, parts + 1).astype(int)
    return [(int(edges[i]) + 1, int(edges[i + 1])) for i in range(parts) if edges[i + 1] > edges[i]]


def make_problem(M, N, K, seed=0, bernoulli_frac=0.0, poisson_frac=0.0, n_views=1, batch_views=0, n_batches=4,
                 nan_frac=0.0, xreg=None, yreg=None, weights=False, col_params=False, layer_regs=False,
                 n_groups=3, noise=0.1, scale=1.0, random_init=False, batch_order="mixed"):
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((K, M)) * scale).astype(np.float32)
    Y = (rng.standard_normal((K, N)) * scale).astype(np.float32)
    logsigma = (rng.standard_normal(N) * 0.1).astype(np.float32) if col_params else np.zeros(N, np.float32)
    mu = rng.standard_normal(N).astype(np.float32) if col_params else np.zeros(N, np.float32)
    # noise model: contiguous ranges normal | bernoulli | poisson (columns sorted by distribution: model.jl:50-54)
    n_b = int(round(N * bernoulli_frac))
    n_p = int(round(N * poisson_frac))
    n_n = N - n_b - n_p
    noise_ranges, kinds = [], []
    c = 1
    for cnt, kd in ((n_b, "bernoulli"), (n_n, "normal"), (n_p, "poisson")):  # alphabetical like the reference sort
        if cnt > 0:
            noise_ranges.append((c, c + cnt - 1))
            kinds.append(kd)
            c += cnt
    kind_of_col = np.empty(N, dtype=object)
    for (s, e), kd in zip(noise_ranges, kinds):
        kind_of_col[s - 1:e] = kd
    view_ranges = split_ranges(N, n_views)
    # batch views: the first `batch_views` feature views get row batches
    bviews = []
    for v in range(min(batch_views, len(view_ranges))):
        s, e = view_ranges[v]
        Nv = e - s + 1
        if batch_order == "mixed":
            bor = np.sort(rng.integers(0, n_batches, size=M)).astype(np.int32)  # contiguous batches like real data
            if v % 2 == 1:
                bor = rng.permutation(bor).astype(np.int32)                      # ... and a scrambled one
            bor[:n_batches] = np.arange(n_batches)                                # every batch non-empty
        else:
            # "sorted": every view has contiguous batches (samples grouped by batch, as assemble_model lays them out);
            # "random": every view scrambled.  Every batch non-empty.
            bor = np.sort(np.concatenate([np.arange(n_batches), rng.integers(0, n_batches, size=M - n_batches)]))
            if batch_order == "random":
                bor = rng.permutation(bor)
            bor = bor.astype(np.int32)
        centers_d = rng.standard_normal(n_batches)[:, None] * 0.25
        centers_t = rng.standard_normal(n_batches)[:, None] * 0.25
        bviews.append(dict(start1=s, stop1=e, batch_of_row=bor,
                           logdelta=(centers_d + 0.25 * rng.standard_normal((n_batches, Nv))).astype(np.float32),
                           theta=(centers_t + 0.25 * rng.standard_normal((n_batches, Nv))).astype(np.float32)))
    # data: Z = layers(X'Y) + noise; Bernoulli 1[Z>0]; Poisson counts
    A = X.astype(np.float64).T @ Y.astype(np.float64)
    Z = A * np.exp(logsigma.astype(np.float64))[None, :]
    for b in bviews:
        sl = slice(b["start1"] - 1, b["stop1"])
        Z[:, sl] *= np.exp(b["logdelta"].astype(np.float64))[b["batch_of_row"], :]
    Z += mu.astype(np.float64)[None, :]
    for b in bviews:
        sl = slice(b["start1"] - 1, b["stop1"])
        Z[:, sl] += b["theta"].astype(np.float64)[b["batch_of_row"], :]
    D = Z + noise * rng.standard_normal((M, N))
    for j in range(N):
        if kind_of_col[j] == "bernoulli":
            D[:, j] = (D[:, j] > 0).astype(np.float64)
        elif kind_of_col[j] == "poisson":
            D[:, j] = rng.poisson(np.exp(np.clip(Z[:, j], -5, 3)))
    D = D.astype(np.float32)
    if nan_frac > 0:
        D[rng.random((M, N)) < nan_frac] = np.nan
    col_weight = (0.5 + rng.random(N)).astype(np.float32) if weights else np.ones(N, np.float32)
    if random_init:  # start the fit away from the generating factors
        X = (rng.standard_normal((K, M)) * 0.3).astype(np.float32)
        Y = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
    p = dict(M=M, N=N, K=K, D=np.asfortranarray(D), X=np.asfortranarray(X), Y=np.asfortranarray(Y),
             logsigma=logsigma, mu=mu, batch_views=bviews, noise_ranges=noise_ranges, noise_kinds=kinds,
             col_weight=col_weight, view_ranges=view_ranges, xreg=[], yreg=[], colreg=None, batchreg=None)
    # regularizers
    if xreg == "l2":
        p["xreg"] = [dict(kind="l2", w=(0.5 + rng.random(K)).astype(np.float32), p=1.0)]
    elif xreg == "group":
        gr = split_ranges(M, n_groups)
        p["xreg"] = [dict(kind="group", start1=[g[0] for g in gr], stop1=[g[1] for g in gr],
                          w=(0.5 + rng.random((len(gr), K))).astype(np.float32), p=1.0)]
    elif xreg == "composite":
        gr = split_ranges(M, n_groups)
        p["xreg"] = [dict(kind="l2", w=(0.5 + rng.random(K)).astype(np.float32), p=0.5),
                     dict(kind="group", start1=[g[0] for g in gr], stop1=[g[1] for g in gr],
                          w=(0.5 + rng.random((len(gr), K))).astype(np.float32), p=0.5)]
    if yreg == "group":
        p["yreg"] = [dict(kind="group", start1=[g[0] for g in noise_ranges], stop1=[g[1] for g in noise_ranges],
                          w=(0.5 + rng.random((len(noise_ranges), K))).astype(np.float32), p=1.0)]
    elif yreg == "ard":
        p["yreg"] = [dict(kind="ard", start1=[g[0] for g in view_ranges], stop1=[g[1] for g in view_ranges],
                          a=np.full(len(view_ranges), 1.001, np.float32),
                          b=np.full(len(view_ranges), 0.001, np.float32), p=1.0)]
    elif yreg == "l2":
        p["yreg"] = [dict(kind="l2", w=(0.5 + rng.random(K)).astype(np.float32), p=1.0)]
    elif yreg == "ard_gap":
        # ARDRegularizer with distinct (alpha, beta) per view range and one view left uncovered (regularizers.jl:546-585
        # loops over its col_ranges only; the library encodes "not regularized" as alpha = -0.5)
        keep = [g for i, g in enumerate(view_ranges) if i != 1 or len(view_ranges) == 1]
        p["yreg"] = [dict(kind="ard", start1=[g[0] for g in keep], stop1=[g[1] for g in keep],
                          a=(1.001 + 0.5 * rng.random(len(keep))).astype(np.float32),
                          b=(0.001 + 0.01 * rng.random(len(keep))).astype(np.float32), p=1.0)]
    elif yreg == "fsard":
        alpha = np.full(N, 1.001, np.float32)
        beta = (0.001 * (0.8 + 2.0 * rng.random((K, N)) * (rng.random((K, N)) < 0.2))).astype(np.float32)
        p["yreg"] = [dict(kind="fsard", alpha=alpha, beta=np.asfortranarray(beta), p=1.0)]
    if layer_regs:
        nr = len(view_ranges)
        p["colreg"] = dict(start1=[g[0] for g in view_ranges], stop1=[g[1] for g in view_ranges],
                           w_logsigma=(0.5 + rng.random(nr)).astype(np.float32),
                           c_logsigma=(0.1 * rng.standard_normal(nr)).astype(np.float32),
                           w_mu=(0.5 + rng.random(nr)).astype(np.float32),
                           c_mu=(0.1 * rng.standard_normal(nr)).astype(np.float32))
        if bviews:
            p["batchreg"] = dict(
                w_logdelta=[(0.5 + rng.random(n_batches)).astype(np.float32) for _ in bviews],
                c_logdelta=[(0.1 * rng.standard_normal(n_batches)).astype(np.float32) for _ in bviews],
                w_theta=[(0.5 + rng.random(n_batches)).astype(np.float32) for _ in bviews],
                c_theta=[(0.1 * rng.standard_normal(n_batches)).astype(np.float32) for _ in bviews])
    return p


def to_oracle(p, precision=64):
    from oracle.pmf_oracle import OracleModel
    noise = [(s, e, k) for (s, e), k in zip(p["noise_ranges"], p["noise_kinds"])]
    return OracleModel(p["D"], p["X"], p["Y"], logsigma=p["logsigma"], mu=p["mu"],
                       batch_views=p["batch_views"] if p["batch_views"] else None,
                       noise=noise, col_weight=p["col_weight"], xreg=p["xreg"], yreg=p["yreg"],
                       colreg=p["colreg"], batchreg=p["batchreg"], precision=precision)


def to_context(p, ctx):
    """Marshals the problem through the C ABI (what the Julia shim's mf_fit! does before ccall(:pmf_fit))."""
    ctx.set_data(p["D"])
    ctx.set_factors(p["X"], p["Y"])
    ctx.set_col_params(p["logsigma"], p["mu"])
    ctx.set_batch_views(p["batch_views"])
    ctx.set_noise(p["noise_ranges"], p["noise_kinds"], p["col_weight"])
    ctx.clear_xreg()
    for t in p["xreg"]:
        _add_term(ctx, "X", t)
    ctx.clear_yreg()
    for t in p["yreg"]:
        _add_term(ctx, "Y", t)
    cr, br = p["colreg"], p["batchreg"]
    if cr is not None or br is not None:
        kw = {}
        if cr is not None:
            kw.update(ranges=list(zip(cr["start1"], cr["stop1"])), w_logsigma=cr["w_logsigma"],
                      c_logsigma=cr["c_logsigma"], w_mu=cr["w_mu"], c_mu=cr["c_mu"])
        if br is not None:
            kw.update(w_logdelta=br["w_logdelta"], c_logdelta=br["c_logdelta"], w_theta=br["w_theta"],
                      c_theta=br["c_theta"])
        ctx.set_layer_regs(**kw)
    else:
        ctx.set_layer_regs()
    return ctx


def _add_term(ctx, which, t):
    if t["kind"] == "l2":
        ctx.add_reg_l2(which, t["w"], t.get("p", 1.0))
    elif t["kind"] == "group":
        ctx.add_reg_group(which, list(zip(t["start1"], t["stop1"])), t["w"], t.get("p", 1.0))
    elif t["kind"] == "ard":
        ctx.add_yreg_ard(list(zip(t["start1"], t["stop1"])), t["a"], t["b"], t.get("p", 1.0))
    elif t["kind"] == "fsard":
        ctx.add_yreg_fsard(t["alpha"], t["beta"], t.get("p", 1.0))
    else:
        raise ValueError(t["kind"])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-30)
    return float(np.max(np.abs(a - b)) / denom)


def shard_problem(p, lo, hi):
    """Rows [lo, hi) of the problem as a problem of its own (what one rank of the row-sharded fit holds): D, X and the
    row -> batch maps are sliced, X-regularizer groups (ranges over the GLOBAL rows) are clipped to the shard, everything
    indexed by columns is replicated."""
    import copy
    q = copy.deepcopy(p)
    q["D"] = np.asfortranarray(p["D"][lo:hi])
    q["X"] = np.asfortranarray(p["X"][:, lo:hi])
    q["M"] = hi - lo
    for b in q["batch_views"]:
        b["batch_of_row"] = np.ascontiguousarray(b["batch_of_row"][lo:hi])
    for t in q["xreg"]:
        if t["kind"] == "group":
            s = [max(a, lo + 1) - lo for a in t["start1"]]
            e = [min(b, hi) - lo for b in t["stop1"]]
            keep = [i for i in range(len(s)) if e[i] >= s[i]]
            t["start1"], t["stop1"] = [s[i] for i in keep], [e[i] for i in keep]
            t["w"] = np.asarray(t["w"])[keep]
    return q
