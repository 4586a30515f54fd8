"""hd16 two-V packed attention, (QG, PF) = (4, 1) [default] against (4, 2) [uwir_attn_variant bit 7], A/B/A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_attn import run
for cfg in ((16, 16384, 4, 16, 2), (4, 65536, 4, 16, 2)):
    for rnd in range(2):
        for lsv in (8, 136):
            run(*cfg, amp=0.5, lsv=lsv)
