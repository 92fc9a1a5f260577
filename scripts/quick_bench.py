"""Quick device timing of the fused data pass (development aid; bench.py is the contract benchmark)."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import  # noqa: E402

pkg = pmf_import.load()


def run(M, N, K, epochs=5, seed=1):
    ctx = pkg.Context(0)
    rng = np.random.default_rng(seed)
    ctx.set_data_device(None, M, N)
    ctx.set_factors((rng.standard_normal((K, M)) * 0.3).astype(np.float32),
                    (rng.standard_normal((K, N)) * 0.3).astype(np.float32))
    ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32))
    ctx.set_batch_views([])
    ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32))
    t0 = time.time()
    ctx.synth_data(seed=123, noise=0.1)
    t_synth = time.time() - t0
    # restart from fresh factors
    ctx.set_factors((rng.standard_normal((K, M)) * 0.1).astype(np.float32),
                    (rng.standard_normal((K, N)) * 0.1).astype(np.float32))
    ctx.add_reg_l2("X", np.ones(K, np.float32))
    ctx.set_optimizer("adagrad", lr=0.05)
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=2, abs_tol=0, rel_tol=0)  # warmup
    ctx.kernel_time(reset=True)
    t0 = time.time()
    r = ctx.fit(update_X=True, update_Y=True, max_epochs=2 + epochs, epoch=3, abs_tol=0, rel_tol=0)
    wall = time.time() - t0
    ms, n = ctx.kernel_time()
    flops = 6.0 * M * N * K
    print(f"M={M} N={N} K={K}: synth {t_synth:.2f}s; {epochs} epochs wall {wall*1e3/epochs:.2f} ms/epoch; "
          f"fused kernel {ms:.3f} ms x{n} -> {flops/ms/1e9:.1f} TF/s ({flops/ms/1e9/157.3*100:.1f}% of f32 MFMA peak), "
          f"D stream {4.0*M*N/ms/1e6:.0f} GB/s; loss {r['loss'][0]:.6g} -> {r['loss'][-1]:.6g} term={r['term_code']}",
          flush=True)
    ctx.close()


if __name__ == "__main__":
    run(500, 200, 4, epochs=20)
    run(20000, 10000, 32, epochs=10)
    run(20000, 10000, 64, epochs=10)
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        run(200000, 50000, 64, epochs=3)
