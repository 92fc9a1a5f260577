#!/bin/bash
# Round-3 rocprofv3 evidence (run through gpurun from the repo root):  bash scripts/profile_r3.sh   (PMF_PROFILE_SET=big | small | config4 | headline | full_model to run a part)
# For every workload: one --kernel-trace --stats run and three separate PMC runs (SQ, FETCH_SIZE, WRITE_SIZE); raw output under
# gpurun_out/prof_r3/<name>/, condensed into profiles/r3_<name>_* by scripts/summarize_r3.py.
set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, then the command's arguments (a python script with its arguments)
  local name=$1; shift
  local d=$OUT/$name
  mkdir -p "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -- python3 "$@" > "$d/stats.log" 2>&1; echo "stats rc=$?" >> "$d/stats.log"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$d/pmc_sq" -- python3 "$@" > "$d/pmc_sq.log" 2>&1; echo "rc=$?" >> "$d/pmc_sq.log"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$d/pmc_fetch" -- python3 "$@" > "$d/pmc_fetch.log" 2>&1; echo "rc=$?" >> "$d/pmc_fetch.log"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$d/pmc_write" -- python3 "$@" > "$d/pmc_write.log" 2>&1; echo "rc=$?" >> "$d/pmc_write.log"
  echo "[$name] done: $(tail -1 $d/stats.log)"
}
B="--steps 4 --warmup 1 --no-cpu-baseline"
if [ "${PMF_PROFILE_SET:-all}" = "config4" ]; then   # both flavours of the configs[4] shard (pmf_fused_sb8_kernel)
  prof config4_shard "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16
  prof config4_shard_full "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 --full-model
  ls "$OUT"; exit 0
fi
if [ "${PMF_PROFILE_SET:-all}" = "headline" ]; then
  prof headline "$ROOT/bench.py" $B
  ls "$OUT"; exit 0
fi
if [ "${PMF_PROFILE_SET:-all}" = "full_model" ]; then   # the full-model flavours only (added late in round 2)
  prof config4_shard_full "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 --full-model
  prof config2_full "$ROOT/bench.py" $B --M 20000 --N 10000 --K 32 --full-model
  ls "$OUT"; exit 0
fi
if [ "${PMF_PROFILE_SET:-all}" = "big" ]; then   # the three large workloads (the rest: PMF_PROFILE_SET=small)
  prof headline "$ROOT/bench.py" $B
  prof config4_shard "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16
  prof config4_shard_full "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 --full-model
  ls "$OUT"; exit 0
fi
if [ "${PMF_PROFILE_SET:-all}" != "small" ]; then
  prof headline "$ROOT/bench.py" $B
  prof config4_shard "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16
  prof config4_shard_full "$ROOT/bench.py" $B --M 125000 --N 100000 --K 128 --precision bf16x3 --store bf16 --full-model
fi
prof config2_full "$ROOT/bench.py" $B --M 20000 --N 10000 --K 32 --full-model
prof config1 "$ROOT/bench.py" $B --M 20000 --N 10000 --K 32
prof general "$ROOT/scripts/kbench_mixed.py" 100000 50000 64 all
prof layers "$ROOT/scripts/kbench_layers.py" 100000 50000 64
prof xonly_yonly "$ROOT/scripts/kbench_xonly.py" 100000 50000 64
ls "$OUT"
