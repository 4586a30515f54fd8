"""hd32 one-V packed attention: (QG, PF) = (4, 2) [default] against (2, 1) [uwir_attn_variant bit 8], A/B/A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_attn import run
run(16, 4096, 12, 32, 1, amp=0.5, lsv=8)
for cfg in ((16, 4096, 12, 32, 1), (4, 16384, 12, 32, 1)):
    for rnd in range(2):
        for lsv in (8, 8 + 256):
            run(*cfg, amp=0.5, lsv=lsv)
