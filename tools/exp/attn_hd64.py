"""hd = 64 attention (level 2 of the student) at the sizes of a 256^2 and a 512^2 forward, fast path on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_attn import run
for cfg in ((16, 1024, 12, 64, 1), (4, 4096, 12, 64, 1), (4, 4096, 4, 64, 2), (1, 16384, 12, 64, 1)):
    for fast in (True, False):
        run(*cfg, amp=0.5, fast=fast)
