"""hd16 two-V packed attention at (QG, PF) = (4, 1): denominator forms 0 / 1 / 2, A/B/A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_attn import run
run(16, 16384, 4, 16, 2, amp=0.5, lsv=2)
for rnd in range(2):
    for lsv in (0, 1, 2):
        run(16, 16384, 4, 16, 2, amp=0.5, lsv=lsv)
