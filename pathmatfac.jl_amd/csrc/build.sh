#!/bin/bash
# Builds libpmf_hip.so for gfx950 (hipcc cross-compiles without a GPU).
# PMF_KEEP_TEMPS=dir keeps the device assembly (.s) there for inspection.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
EXTRA=()
if [[ -n "${PMF_KEEP_TEMPS:-}" ]]; then mkdir -p "$PMF_KEEP_TEMPS"; cd "$PMF_KEEP_TEMPS"; EXTRA+=(-save-temps); else cd "$HERE"; fi
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics \
  -Wall -Wno-unused-function "${EXTRA[@]}" -o "$HERE/../libpmf_hip.so" "$HERE/pmf_hip.hip" "$@"
