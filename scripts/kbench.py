"""Fused-kernel timing at other factor ranks (development aid): python scripts/kbench.py M N K [K ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
import quick_bench
M, N = int(sys.argv[1]), int(sys.argv[2])
for K in sys.argv[3:]:
    quick_bench.run(M, N, int(K), epochs=3)
