#!/usr/bin/env python3
"""Measurement aid: the hand-written NT GEMM next to the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the step's
own shapes.  Not used by the product path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


bf = torch.bfloat16
shapes = [(201728, 2304, 768), (201728, 768, 768), (201728, 3072, 768), (201728, 768, 3072),
          (78848, 768, 768), (50432, 3072, 768), (25216, 3072, 768), (16384, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf)
    ms = timeit(lambda: ops.gemm_nt(a, b, c))
    ms2 = timeit(lambda: torch.matmul(a, b.t(), out=c))
    f = 2 * M * N * K / 1e9
    print(f"nt {M}x{N}x{K}: ours {ms:.3f} ms {f/ms:.0f} TF/s | library {ms2:.3f} ms {f/ms2:.0f} TF/s", flush=True)
    del a, b, c
