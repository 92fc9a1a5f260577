#!/bin/bash
# fused-kernel time vs column-segment length (PMF_TPS_SCALE scales the model's choice), split-bf16 and exact
for prec in bf16x3 f32; do
  for s in 1 0.5 0.3 0.2 0.1; do
    echo -n "prec=$prec tps_scale=$s: "
    PMF_PRECISION=$prec PMF_TPS_SCALE=$s python scripts/kbench.py ${1:-200000} ${2:-50000} ${3:-64} 2>&1 | grep -o "fused kernel [0-9.]* ms"
  done
done
