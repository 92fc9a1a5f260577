#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root).
# usage: scripts/profile.sh <tag>      -> gpurun_out/prof_<tag>/{stats,pmc_sq,pmc_fetch,pmc_write}
set -uo pipefail
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
echo "stats rc=$?" >> "$OUT/stats.log"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?" >> "$OUT/pmc_sq.log"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?" >> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE TCC_EA0_ATOMIC_sum --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?" >> "$OUT/pmc_write.log"
ls -R "$OUT" | head -50
