"""CPU baselines for bench.py's `cpu_baseline` leg  --  TEST / MEASUREMENT INFRASTRUCTURE ONLY (never imported by the product).

Two ports of the reference's CPU epoch, timed on the GPU box's host cores on a bounded row sample:

  * `openmp`: the float build of oracle/pmf_oracle.c (one fused loop nest per step, OpenMP over columns), with the team
    sized by the work and by the CPUs the process may really use (cgroup quota / affinity), not by the host's thread count;
  * `blas`:  an epoch with the STRUCTURE of the reference's own CPU path: per row batch of `capacity` = 10^8 entries
    (src/fit.jl:938) three BLAS sgemm calls -- transpose(X)*Y (src/layers.jl:283), and the two products of its Zygote
    adjoint, Y*Gbar' and X_b*Gbar -- with every column layer a materialised m x N broadcast (ColScale `Z .* transpose(exp.(logsigma))`
    src/layers.jl:20-22, pull-back :34-48; ColShift `Z .+ transpose(mu)` :64-66, pull-back :78-90), the masked Gaussian
    loss (self-specified, as everywhere: MatFac is un-vendored), the group regularizer on X and feature-set-ARD on Y
    (src/regularizers.jl:423-446, src/featureset_ard.jl:135-150) and one optimizer step per factor.  numpy's BLAS
    (OpenBLAS) supplies sgemm, as OpenBLAS does under Julia.

Neither is the reference itself (Julia is not in the image): both are `kind: "port"`.
"""
import os
import time

import numpy as np


def usable_cpus():
    """CPUs this process may use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def threads_for(rows, N, K, cap=None):
    """OpenMP team for one epoch of a rows x N problem at K factors: one thread per 2e6 multiply-adds, at most `cap`."""
    cap = cap or usable_cpus()
    return int(min(cap, max(1, (rows * N * K) // 2_000_000)))


def _problem(N, K, rows, seed):
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((K, rows)) * 0.3).astype(np.float32)
    Y = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
    D = (X.T @ Y + 0.1 * rng.standard_normal((rows, N), dtype=np.float32)).astype(np.float32)
    X0 = (rng.standard_normal((K, rows)) * 0.1).astype(np.float32)
    Y0 = (rng.standard_normal((K, N)) * 0.1).astype(np.float32)
    ngr = 4
    edges = np.linspace(0, rows, ngr + 1).astype(int)
    return D, X0, Y0, edges


def openmp_port(N, K, rows, epochs, seed, opt, lr, threads=None):
    """Seconds per epoch of the C oracle's float build on a rows x N sample; returns (s_per_epoch, threads_used)."""
    from oracle import pmf_oracle as po
    lib = po.get_lib(32).lib
    threads = threads or threads_for(rows, N, K)
    lib.o_set_num_threads(int(threads))
    D, X0, Y0, edges = _problem(N, K, rows, seed)
    ngr = len(edges) - 1
    m = po.OracleModel(D, X0, Y0,
                       xreg=[dict(kind="group", start1=list(edges[:-1] + 1), stop1=list(edges[1:]),
                                  w=np.ones((ngr, K), np.float32))],
                       yreg=[dict(kind="fsard", alpha=np.full(N, 1.001, np.float32),
                                  beta=np.full((K, N), 0.001, np.float32))], precision=32)
    m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=1, abs_tol=0, rel_tol=0)   # warm
    t0 = time.perf_counter()
    m.fit(update_X=True, update_Y=True, opt=opt, lr=lr, max_epochs=1 + epochs, epoch=2, abs_tol=0, rel_tol=0)
    return (time.perf_counter() - t0) / epochs, int(lib.o_num_threads())


class _Adam:
    def __init__(self, lr, shape, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.m = np.zeros(shape, np.float32)
        self.v = np.zeros(shape, np.float32)
        self.p1, self.p2 = b1, b2

    def step(self, p, g):
        self.m *= self.b1; self.m += (1 - self.b1) * g
        self.v *= self.b2; self.v += (1 - self.b2) * g * g
        p -= self.lr * (self.m / (1 - self.p1)) / (np.sqrt(self.v / (1 - self.p2)) + self.eps)
        self.p1 *= self.b1; self.p2 *= self.b2


class _AdaGrad:
    def __init__(self, lr, shape, eps=1e-8):
        self.lr, self.eps = lr, eps
        self.acc = np.full(shape, eps, np.float32)   # Flux AdaGrad: accumulator starts at eps (src/optimizers.jl:6-13)

    def step(self, p, g):
        self.acc += g * g
        p -= self.lr * g / (np.sqrt(self.acc) + self.eps)


_POOL = None


def _par_rows(fn, m):
    """Run fn(r0, r1) over row slices on the usable CPUs (numpy ufuncs release the GIL): the m x N broadcasts of the
    reference are memory-bound single passes; one thread per slice keeps them from hiding the sgemm time."""
    global _POOL
    n = usable_cpus()
    if n <= 1 or m < 4 * n:
        return [fn(0, m)]
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(n)
    step = -(-m // n)
    return list(_POOL.map(lambda r: fn(r, min(m, r + step)), range(0, m, step)))


def blas_epoch(D, X, Y, logsigma, mu, w, wq_x, alpha, beta, optX, optY, capacity=10**8):
    """One fit! epoch on the CPU the way the reference runs it: row batches of `capacity` entries, three sgemm per
    batch, every layer and its pull-back an m x N pass over a materialised matrix.  Updates X, Y in place; returns the
    loss."""
    M, N = D.shape
    bs = max(1, min(M, capacity // N))
    sigma = np.exp(logsigma)
    gX = np.empty_like(X)
    gY = np.zeros_like(Y)
    loss = 0.0
    for i0 in range(0, M, bs):
        Xb = X[:, i0:i0 + bs]
        Db = D[i0:i0 + bs]
        Z = Xb.T @ Y                              # sgemm 1: transpose(X)*Y  (m x N, materialised)

        def layers_and_loss(r0, r1):
            z, d = Z[r0:r1], Db[r0:r1]
            np.multiply(z, sigma[None, :], out=z)     # ColScale            src/layers.jl:20-22
            np.add(z, mu[None, :], out=z)             # ColShift            src/layers.jl:64-66
            np.subtract(z, d, out=z)                  # residual; NaN where D is missing
            np.copyto(z, np.float32(0), where=np.isnan(z))   # NaN mask (0 loss, 0 gradient)
            l = 0.5 * float(np.dot((z * z).sum(axis=0, dtype=np.float64), w))   # masked Gaussian loss
            np.multiply(z, w[None, :], out=z)         # G = dloss/dZ = ColShift's pull-back input (Z_bar = copy)
            np.multiply(z, sigma[None, :], out=z)     # ColScale pull-back  src/layers.jl:34-48
            return l
        loss += sum(_par_rows(layers_and_loss, Z.shape[0]))
        gX[:, i0:i0 + bs] = Y @ Z.T               # sgemm 2: grad(X) = Y * Abar'
        gY += Xb @ Z                              # sgemm 3: grad(Y) += X_b * Abar
    # GroupRegularizer on X (dense weights), FeatureSetARD on Y
    loss += 0.5 * float(np.sum(wq_x * X * X, dtype=np.float64))
    gX += wq_x * X
    b = 1.0 + (0.5 / beta) * Y * Y
    loss += float(np.sum((0.5 + alpha)[None, :] * np.log(b), dtype=np.float64))
    gY += (alpha + 0.5)[None, :] * Y / (b * beta)
    optX.step(X, gX)
    optY.step(Y, gY)
    return loss


def _blas_limit():
    """BLAS threads = the CPUs this process may use (OpenBLAS sizes its pool by the host's core count: 64 threads on a
    16-CPU cgroup share is slower than 16)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=usable_cpus(), user_api="blas")
    except Exception:
        import contextlib
        return contextlib.nullcontext()


def blas_port(N, K, rows, epochs, seed, opt, lr):
    """Seconds per epoch of the BLAS-structured port on a rows x N sample; returns (s_per_epoch, blas_threads)."""
    with _blas_limit():
        return _blas_port(N, K, rows, epochs, seed, opt, lr)


def _blas_port(N, K, rows, epochs, seed, opt, lr):
    D, X, Y, edges = _problem(N, K, rows, seed)
    X = np.asfortranarray(X).copy(order="C")
    logsigma = np.zeros(N, np.float32)
    mu = np.zeros(N, np.float32)
    w = np.ones(N, np.float32)
    wq = np.ones_like(X)
    alpha = np.full(N, 1.001, np.float32)
    beta = np.full((K, N), 0.001, np.float32)
    mk = (lambda s: _Adam(lr, s)) if opt == "adam" else (lambda s: _AdaGrad(lr, s))
    oX, oY = mk(X.shape), mk(Y.shape)
    blas_epoch(D, X, Y, logsigma, mu, w, wq, alpha, beta, oX, oY)   # warm
    t0 = time.perf_counter()
    for _ in range(epochs):
        blas_epoch(D, X, Y, logsigma, mu, w, wq, alpha, beta, oX, oY)
    dt = (time.perf_counter() - t0) / epochs
    nthr = 0
    try:
        from threadpoolctl import threadpool_info
        nthr = max([p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"] or [0])
    except Exception:
        pass
    return dt, int(nthr or usable_cpus())


def sgemm_rate(m=2048, n=4096, k=64, reps=3):
    """GFLOP/s numpy's sgemm sustains on an m x k by k x n product here (the yardstick for the `blas` leg)."""
    rng = np.random.default_rng(0)
    a = rng.standard_normal((m, k), dtype=np.float32)
    b = rng.standard_normal((k, n), dtype=np.float32)
    with _blas_limit():
        a @ b
        t0 = time.perf_counter()
        for _ in range(reps):
            a @ b
        dt = time.perf_counter() - t0
    return 2.0 * m * n * k * reps / dt / 1e9
