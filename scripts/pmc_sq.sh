#!/bin/bash
# SQ stall-category counters of the fused kernel (two separate --pmc passes; run through gpurun from the repo root)
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_${1:-x}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/a" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/a.log" 2>&1
echo "a rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM \
  --output-format csv -d "$OUT/b" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/b.log" 2>&1
echo "b rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob(f"{sys.argv[1]}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pmf_fused_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                n += 1
    nl = max(1, n // max(1, len(acc)))
    for k, v in sorted(acc.items()):
        print(f"{sub} {k:32s} {v / nl:.4g} per launch")
PY
