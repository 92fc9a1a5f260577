#!/bin/bash
# Builds libpmf_hip.so for gfx950 (hipcc cross-compiles without a GPU): csrc/Makefile, translation units in parallel.
# PMF_KEEP_TEMPS=dir keeps the device assembly (.s) of every unit there for inspection.  Extra arguments go to hipcc.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
EXTRA="$*"
if [[ -n "${PMF_KEEP_TEMPS:-}" ]]; then mkdir -p "$PMF_KEEP_TEMPS"; EXTRA="$EXTRA -save-temps=obj"; fi
make -s -j"${PMF_BUILD_JOBS:-8}" -C "$HERE" EXTRA="$EXTRA" ${PMF_LIB:+LIB="$PMF_LIB"} ${PMF_BUILD_DIR:+B="$PMF_BUILD_DIR"}
if [[ -n "${PMF_KEEP_TEMPS:-}" ]]; then cp "$HERE"/.build/*.s "$PMF_KEEP_TEMPS"/ 2>/dev/null || true; fi
