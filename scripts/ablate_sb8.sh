#!/bin/bash
# Measurement-only builds of pmf_fused_sb8_kernel (WRONG results), one cost removed per build (bits: csrc/pmf_fused_sb8.hip.inc):
#   bash scripts/ablate_sb8.sh build   (CPU, ~1 min per variant)  ->  pathmatfac.jl_amd/libpmf_abl8_<bits>.so
#   bash scripts/ablate_sb8.sh run M N K [store]   (GPU)          ->  one process, all variants interleaved (scripts/ab_many.py)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
BITS="${PMF_ABL_BITS:-0 1 2 4 8 16 32 64 128}"
if [ "${1:-}" = "build" ]; then
  for b in $BITS; do
    PMF_LIB="$ROOT/pathmatfac.jl_amd/libpmf_abl8_$b.so" PMF_BUILD_DIR=".build_ab8_$b" "$ROOT/pathmatfac.jl_amd/csrc/build.sh" -DPMF_ABLATE=$b 2>&1 | grep -E "error" || true
    echo "built $b"
  done
else
  shift || true
  libs=""
  for b in $BITS; do libs="$libs $ROOT/pathmatfac.jl_amd/libpmf_abl8_$b.so"; done
  PMF_AB_STORE="${4:-f32}" PMF_PRECISION=bf16x3 python3 "$ROOT/scripts/ab_many.py" "$1" "$2" "$3" $libs
fi
