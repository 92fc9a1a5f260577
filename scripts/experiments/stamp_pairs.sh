#!/bin/bash
# Builds one diagnostic library per (from, to) stamp pair of pmf_fused_sb2p_kernel (two-stamp variant of PMF_STAMPS: see
# pmf_common.h) into gpurun_scratch/stamp_<from>_<to>.so.  CPU only.  Then on the GPU: scripts/stamp_pairs_run.py
cd "$(dirname "$0")/.."
mkdir -p gpurun_scratch
for pair in "0 1" "1 2" "2 3" "3 4" "4 10" "10 9" "9 5" "5 6" "6 0"; do
  set -- $pair
  PMF_LIB=$PWD/gpurun_scratch/stamp_$1_$2.so PMF_BUILD_DIR=.build_st_$1_$2 pathmatfac.jl_amd/csrc/build.sh -DPMF_STAMPS -DPMF_STAMP_FROM=$1 -DPMF_STAMP_TO=$2 2>&1 | grep -E "error" | head -3
done
ls -la gpurun_scratch/*.so
