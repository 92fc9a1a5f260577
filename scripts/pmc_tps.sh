#!/bin/bash
# HBM-side traffic of the configs[4] shard pass as a function of the column-segment length (PMF_TPS_SCALE); run through gpurun.
# usage: bash scripts/pmc_tps.sh "1 0.6 0.4" [extra bench args]
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tps
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SCALES=${1:-"1 0.5"}
shift || true
for sc in $SCALES; do
  export PMF_TPS_SCALE=$sc
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $pmc --output-format csv -d "$OUT/${pmc}_$sc" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 "$@" > "$OUT/${pmc}_$sc.log" 2>&1
  done
  python3 - "$OUT" "$sc" <<'PY'
import csv, glob, sys, collections
out, sc = sys.argv[1], sys.argv[2]
tot = {}
for pmc in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/{pmc}_{sc}/*/*_counter_collection.csv")
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "pmf_fused" in r["Kernel_Name"] and r["Counter_Name"] == pmc]
    tot[pmc] = sum(v) / len(v)
rd, wr = tot["FETCH_SIZE"] * 2 * 1024 / 1e9, tot["WRITE_SIZE"] * 1024 / 1e9
print(f"tps_scale {sc}: read {rd:.1f} GB + written {wr:.1f} GB = {rd + wr:.1f} GB per launch")
PY
done
