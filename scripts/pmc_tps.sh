#!/bin/bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and time of the K = 128 split kernel against the column-segment length
# (development aid):  bash scripts/pmc_tps.sh   (through gpurun, from the repo root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tps
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PMF_BENCH_PRECISION=bf16x3 PMF_BENCH_STORE=bf16
for s in 1 0.5 0.25; do
  for c in FETCH_SIZE WRITE_SIZE; do
    PMF_TPS_SCALE=$s rocprofv3 --pmc $c --output-format csv -d "$OUT/s${s}_$c" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --M 125000 --N 100000 --K 128 > "$OUT/s${s}_$c.log" 2>&1
    f=$(find "$OUT/s${s}_$c" -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$s" "$c" <<'PY'
import csv, sys
f, s, c = sys.argv[1:4]
tot, n = 0.0, 0
for r in csv.DictReader(open(f)):
    if "pmf_fused_sb4" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
        tot += float(r["Counter_Value"]); n += 1
# one row per (dispatch, counter) or per dimension: sum over rows of a dispatch, mean over dispatches
import collections
d = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "pmf_fused_sb4" in r.get("Kernel_Name", "") and r.get("Counter_Name") == c:
        d[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = sum(d.values()) / max(1, len(d))
gb = v * (2 * 1024 if c == "FETCH_SIZE" else 1024) / 1e9 if c == "FETCH_SIZE" else v * 1024 / 1e9
print(f"tps_scale={s} {c}: {v:.4g} per launch -> {gb:.1f} GB ({len(d)} launches)")
PY
  done
done
