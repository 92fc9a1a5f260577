"""End-to-end fit!(model) (src/fit.jl:923) on a mid-size synthetic model, with a per-stage time line (development aid).
python scripts/fit_e2e.py [M N K]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import pmf_import
pkg = pmf_import.load()
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (20000, 10000, 32)
rng = np.random.default_rng(0)
nrb, nsets = 4, 20
Z = (rng.standard_normal((K, M)).astype(np.float32).T @ rng.standard_normal((K, N)).astype(np.float32)).astype(np.float32)
Z[rng.random((M, N)) < 0.02] = np.nan
conds = [f"condition_{1 + (i * 2) // M}" for i in range(M)]
fids = [f"x_{i}" for i in range(1, N + 1)]
views = [1] * (N // 2) + [2] * (N - N // 2)
batch_dict = {v: [f"rowbatch{1 + (i * nrb) // M}" for i in range(M)] for v in (1, 2)}
fsets = {}
for v, (lo, hi) in enumerate(((0, N // 2), (N // 2, N)), start=1):
    edges = np.linspace(lo, hi, nsets + 1).astype(int)
    fsets[v] = [[fids[j] for j in range(edges[s], edges[s + 1])] for s in range(nsets)]
t0 = time.time()
model = pkg.make_model(Z, K=K, sample_conditions=conds, feature_views=views, feature_ids=fids, batch_dict=batch_dict,
                       feature_sets_dict=fsets, Y_fsard=True, fsard_v0=0.5, rng=rng)
print(f"make_model {time.time()-t0:.2f} s")
t0 = time.time()
hist = pkg.fit_(model, verbosity=0, lr=0.05, max_epochs=200, rel_tol=1e-5, abs_tol=1e-5, fsard_term_rtol=1e-3,
                fsard_max_iter=2, fsard_max_A_iter=200, keep_history=True)
tot = time.time() - t0
print(f"fit_ total {tot:.2f} s, {len(hist)} history entries")
prev = None
for d in hist:
    t = d.get("time", None)
    nm = d.get("name")
    ep = d.get("epochs", "")
    el = d.get("elapsed", "")
    print(f"  {nm!s:38s} epochs={ep!s:6s} term={d.get('term_code','')!s:14s} t={t if t is None else round(t,2)} {el}")
