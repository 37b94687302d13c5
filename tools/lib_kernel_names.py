#!/usr/bin/env python3
"""Run the vendor-library GEMM on two of the step's shapes (for a rocprofv3 kernel trace: which library kernel is chosen)."""
import torch
bf = torch.bfloat16
for (M, N, K) in [(201728, 2304, 768), (201728, 768, 3072), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf)
    for _ in range(3):
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
