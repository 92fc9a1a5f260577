"""Ablation timing of the fused data-pass kernel (development aid): PMF_DEBUG_FLAGS bits and gradient flags."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import  # noqa: E402

pkg = pmf_import.load()
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (40000, 20000, 64)))
ctx = pkg.Context(0)
rng = np.random.default_rng(1)
ctx.set_data_device(None, M, N)
ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32), (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32))
ctx.set_batch_views([])
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
ctx.synth_data(seed=5, noise=0.1)
flops = 6.0 * M * N * K


def t(label, dbg, ux, uy, reps=5):
    os.environ["PMF_DEBUG_FLAGS"] = str(dbg)
    o = ctx.make_opts(update_X=ux, update_Y=uy)
    for _ in range(2):
        ctx.epoch_begin(o)
    ctx.epoch_loss()
    ctx.kernel_time(reset=True)
    for _ in range(reps):
        ctx.epoch_begin(o)
    ctx.epoch_loss()
    ms, n = ctx.kernel_time()
    nm = (32 + (32 if ux else 0) + (32 if uy else 0)) * (K // 64 if K >= 64 else 1)
    print(f"{label:40s} dbg={dbg:2d} gx={int(ux)} gy={int(uy)}: {ms:8.3f} ms  ({flops/ms/1e9:6.1f} TF/s-equivalent of full 6MNK)", flush=True)


t("full", 0, True, True)
t("no LDS adds", 1, True, True)
t("no D loads", 2, True, True)
t("no epilogue", 4, True, True)
t("no flush", 8, True, True)
t("no adds+flush", 9, True, True)
t("no adds/loads/epi/flush", 15, True, True)
t("no B2 barrier (wrong results)", 16, True, True)
t("no B2, no reduce/flush", 25, True, True)
t("no slab writes", 32, True, True)
t("skeleton w/o B2, slab writes", 15 + 16 + 32, True, True)
t("forward+GEMM2 only", 0, True, False)
t("forward+GEMM3 only", 0, False, True)
t("forward only", 0, False, False)
t("forward only, no loads/epi", 6, False, False)
