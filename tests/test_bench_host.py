"""Host-side pieces of bench.py that nobody can rehearse on eight GPUs: the rank -> device map, and the CPU baselines
(oracle/cpu_baseline.py) -- the BLAS-structured leg must compute the same epoch as the C oracle, and the OpenMP team is
sized by the work."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_device_for_rank():
    f = _bench().device_for_rank
    assert [f(r, 8) for r in range(8)] == list(range(8))      # the whole node visible: LOCAL_RANK is the device
    assert f(3, 1) == 0 and f(0, 1) == 0 and f(7, 1) == 0     # a launcher that pins one device per rank: device 0
    with pytest.raises(RuntimeError, match="only 4 visible"):
        f(5, 4)
    with pytest.raises(RuntimeError, match="no HIP device"):
        f(0, 0)


def test_openmp_team_is_sized_by_the_work():
    from oracle import cpu_baseline as cb
    assert cb.threads_for(500, 200, 4, cap=256) == 1            # configs[0]: 4e5 multiply-adds, one thread
    assert cb.threads_for(2000, 10000, 32, cap=16) == 16
    assert cb.threads_for(2000, 10000, 32, cap=256) == 256
    assert 1 <= cb.usable_cpus() <= 4096


def test_blas_leg_computes_the_oracles_epoch():
    """One epoch of the BLAS-structured port == one epoch of the C oracle (fp64): loss at the initial parameters and the
    parameters after the AdaGrad step."""
    from oracle import cpu_baseline as cb
    from oracle import pmf_oracle as po
    N, K, rows = 300, 8, 200
    D, X0, Y0, edges = cb._problem(N, K, rows, 3)
    D[5, 7] = np.nan
    D[100, 0] = np.nan
    ngr = len(edges) - 1
    alpha = np.full(N, 1.001, np.float32)
    beta = np.full((K, N), 0.001, np.float32)
    m = po.OracleModel(D, X0, Y0,
                       xreg=[dict(kind="group", start1=list(edges[:-1] + 1), stop1=list(edges[1:]), w=np.ones((ngr, K)))],
                       yreg=[dict(kind="fsard", alpha=alpha, beta=beta)], precision=64)
    r = m.fit(update_X=True, update_Y=True, opt="adagrad", lr=0.05, max_epochs=1, abs_tol=0, rel_tol=0)
    X, Y = X0.copy(), Y0.copy()
    oX, oY = cb._AdaGrad(0.05, X.shape), cb._AdaGrad(0.05, Y.shape)
    loss = cb.blas_epoch(D, X, Y, np.zeros(N, np.float32), np.zeros(N, np.float32), np.ones(N, np.float32), np.ones_like(X),
                         alpha, beta, oX, oY, capacity=64 * N)      # (several row batches)
    assert abs(loss - r["loss"][0]) <= 2e-5 * abs(r["loss"][0])
    assert np.abs(X - m.X).max() <= 2e-4 * np.abs(m.X).max()
    assert np.abs(Y - m.Y).max() <= 2e-4 * np.abs(m.Y).max()


def test_cpu_baselines_are_fast_on_the_reference_sized_case():
    """configs[0] (500 x 200, K = 4): an epoch takes milliseconds on one thread -- >= 100 iterations/s (round 2 spun 256
    OpenMP threads on it and measured 1.2)."""
    from oracle import cpu_baseline as cb
    t_omp, thr = cb.openmp_port(200, 4, 500, 20, 1, "adam", 0.01)
    t_blas, _ = cb.blas_port(200, 4, 500, 20, 1, "adam", 0.01)
    assert thr == 1
    assert 1.0 / t_omp >= 100 and 1.0 / t_blas >= 100, (t_omp, t_blas)
