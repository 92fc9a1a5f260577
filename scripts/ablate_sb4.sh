#!/bin/bash
# builds measurement-only variants of the library (PMF_ABLATE bits, see pmf_fused_sb4.hip.inc) next to the real one
# usage (CPU container): scripts/ablate_sb4.sh 1 2 4 ...   -> pathmatfac.jl_amd/libpmf_ab<bits>.so
set -e
cd "$(dirname "$0")/../pathmatfac.jl_amd/csrc"
for b in "$@"; do
  PMF_LIB=../libpmf_ab$b.so PMF_BUILD_DIR=.build_ab$b ./build.sh -DPMF_ABLATE=$b
done
