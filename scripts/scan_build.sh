#!/bin/bash
# Compiles every translation unit of libpmf_hip.so with -save-temps (each in a directory of its own) and scans the device
# assembly for the VGPR->AGPR pair-copy miscompile (scripts/scan_agpr_pair_copy.py).  CPU only; ~2 min with 8 jobs.
set -u
CS="${PMF_CSRC:-$(cd "$(dirname "$0")/../pathmatfac.jl_amd/csrc" && pwd)}"
OUT=${1:-/tmp/pmf_scan}
rm -rf "$OUT"; mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$CS"
one() { local name=$1 src=$2; shift 2; mkdir -p "$OUT/$name"; (cd "$OUT/$name" && /opt/rocm/bin/hipcc $FLAGS "$@" -save-temps -c "$CS/$src" -o x.o >/dev/null 2>&1); }
jobs_=()
for kr in 11 12 21 31 41; do for db in 0 1; do one fused_${kr}${db} pmf_k_fused.hip -DPMF_KB=${kr:0:1} -DPMF_RBW=${kr:1:1} -DPMF_DB=$db & done; done
wait
for kb in 1 2; do for db in 0 1; do one sb_${kb}${db} pmf_k_sb.hip -DPMF_KB=$kb -DPMF_DB=$db & done; done
for db in 0 1; do one sb2_$db pmf_k_sb2.hip -DPMF_DB=$db & one sb4_4$db pmf_k_sb4.hip -DPMF_KB=4 -DPMF_DB=$db & done
one layers pmf_k_layers.hip & one main pmf_hip.hip &
wait
python3 "$(dirname "$0")/scan_agpr_pair_copy.py" "$OUT"/*/*gfx950*.s | grep -v ": 0 suspicious"
