#!/bin/bash
# repeat the 2-rank worker pair; stop at the first run whose debug output shows a non-finite term
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25; do
  python scripts/dbg_2rank.py ${PREC:-f32} > gpurun_out/d2_$i.log 2>&1
  if grep -q "inf\|nan\|nonfinite [1-9]" gpurun_out/d2_$i.log; then echo "run $i: BAD"; grep "pmf rank" gpurun_out/d2_$i.log | head -30; break; else echo "run $i ok"; fi
done
