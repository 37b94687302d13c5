#!/usr/bin/env python3
"""Measurement aid: plain wgrad (gemm_tn4w) time against the minimum rows per M range (option 9) at small M."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops
bf = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M in (25216, 50432, 201728):
    for (Nn, Kk) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        g = torch.randn(M, Nn, device="cuda").to(bf); x = torch.randn(M, Kk, device="cuda").to(bf)
        dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
        out = []
        for mr in (512, 1024, 2048, 4096):
            ops.set_option(9, mr)
            out.append(f"{mr}: {timeit(lambda: ops.gemm_tn(g, x, dw, db=db))*1e3:.0f} us")
        ops.set_option(9, 2048)
        print(f"M {M} {Nn}x{Kk}: " + " | ".join(out), flush=True)
