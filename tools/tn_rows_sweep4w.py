#!/usr/bin/env python3
"""Measurement aid: rows per M range (option 10; 0 = the automatic choice) of the grouped wgrad on gemm_tn4w, balanced and skewed groups, two batch sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bf = torch.bfloat16
Nn, Kk, E = 768, 768, 8
for Bimg in (1024, 128):
    M = 2 * Bimg * 196
    g = torch.randn(M, Nn, device="cuda").to(bf); x = torch.randn(Bimg * 197, Kk, device="cuda").to(bf)
    gath = ((torch.arange(M, device="cuda") // 196 % Bimg) * 197 + 1 + torch.arange(M, device="cuda") % 196).int()
    dw = torch.zeros(E, Nn, Kk, device="cuda"); db = torch.zeros(E, Nn, device="cuda")
    bal = torch.tensor([i * (M // E) for i in range(E + 1)], device="cuda", dtype=torch.int32)
    skew = torch.tensor([0, M // 2, M // 2, M // 2, M, M, M, M, M], device="cuda", dtype=torch.int32)
    for name, off in (("balanced", bal), ("2 of 8 groups", skew)):
        out = []
        for rows in (0, 1024, 2048, 4096, 8192, 16384):
            ops.set_option(10, rows)
            out.append(f"{rows}: {timeit(lambda: ops.gemm_tn(g, x, dw, db=db, x_rowmap=gath, row_off=off, n_groups=E, stride_w=Nn * Kk, stride_db=Nn, M=M))*1e3:.0f} us")
        ops.set_option(10, 0)
        print(f"B {Bimg} {name:14s}: " + " | ".join(out), flush=True)
