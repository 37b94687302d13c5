#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bf = torch.bfloat16
for (M, N, K) in [(8192, 3072, 768), (16384, 3072, 768), (50432, 3072, 768), (8192, 768, 3072), (8192, 3072, 3072), (16384, 4096, 4096), (201728, 3072, 768), (201728, 768, 768)]:
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf)
    ms = timeit(lambda: ops.gemm_nt(a, b, c))
    print(f"nt {M}x{N}x{K}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
    del a, b, c
