"""Socket power / clocks while the fused kernel runs back to back (development aid; rocm-smi is sampled from a thread while
the main thread queues epochs):   PMF_PRECISION=bf16x3 python scripts/power_trace.py M N K [store] [zero]"""
import os
import subprocess
import sys
import threading
import time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in sys.argv[1:4])
store = sys.argv[4] if len(sys.argv) > 4 else "f32"
zero = len(sys.argv) > 5 and sys.argv[5] == "zero"
rng = np.random.default_rng(1)
sc = 0.0 if zero else 1.0
X0 = (rng.standard_normal((K, M)) * 0.1 * sc).astype(np.float32); Y0 = (rng.standard_normal((K, N)) * 0.1 * sc).astype(np.float32)
Xt = (rng.standard_normal((K, M)) * 0.3 * sc).astype(np.float32); Yt = (rng.standard_normal((K, N)) * 0.3 * sc).astype(np.float32)
ctx = pkg.Context(0)
ctx.set_data_device(None, M, N, store=store)
ctx.set_factors(Xt, Yt)
ctx.set_col_params(np.zeros(N, np.float32), np.zeros(N, np.float32)); ctx.set_batch_views([])
ctx.set_noise([(1, N)], ["normal"], np.ones(N, np.float32)); ctx.synth_data(seed=5, noise=0.0 if zero else 0.1)
ctx.set_factors(X0, Y0)
ctx.set_optimizer("adagrad", lr=0.0)
ctx.fit(update_X=True, update_Y=True, max_epochs=2, abs_tol=0, rel_tol=0)
samples, stop = [], False


def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
        samples.append((time.time(), r.stdout))
        time.sleep(0.05)


th = threading.Thread(target=sampler); th.start()
ctx.kernel_time(reset=True)
t0 = time.time()
ctx.fit(update_X=True, update_Y=True, max_epochs=int(os.environ.get("PMF_PT_EPOCHS", "150")), epoch=2, abs_tol=0, rel_tol=0)
t1 = time.time()
ms, n = ctx.kernel_time()
stop = True; th.join()
import json
pw, sclk = [], []
for t, out in samples:
    if not (t0 + 0.3 <= t <= t1 - 0.1):
        continue
    try:
        d = json.loads(out)
    except Exception:
        continue
    c = d.get("card0", {})
    for k, v in c.items():
        kl = k.lower()
        if "power" in kl and "(w)" in kl:
            try: pw.append(float(v))
            except ValueError: pass
        if kl.startswith("sclk clock speed"):
            try: sclk.append(float(str(v).strip("()Mhz ")))
            except ValueError: pass
print(f"{M}x{N} K={K} store={store} zero={zero} precision={os.environ.get('PMF_PRECISION', 'f32')}: kernel {ms:.2f} ms avg of {n}; "
      f"power {np.mean(pw) if pw else float('nan'):.0f} W (max {max(pw) if pw else float('nan'):.0f}, {len(pw)} samples); sclk {np.mean(sclk) if sclk else float('nan'):.0f} MHz", flush=True)
if os.environ.get("PMF_PT_RAW") and samples:
    print(samples[len(samples) // 2][1])
