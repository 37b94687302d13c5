#!/usr/bin/env python3
"""Micro-benchmarks of the GEMM kernels at the ViT-B shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    M = 50432
    bf = torch.bfloat16
    print("gemm_nt  (M, N, K): ms, TFLOP/s")
    for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
        c = torch.empty(M, N, device="cuda", dtype=bf)
        ms = timeit(lambda: ops.gemm_nt(a, b, c))
        print(f"  nt {M}x{N}x{K}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s")
    print("gemm_tn  (M, Nn, Kk, nsplit)")
    for (Nn, Kk) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        g = torch.randn(M, Nn, device="cuda").to(bf); x = torch.randn(M, Kk, device="cuda").to(bf)
        dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
        for ns in (1, 2, 4, 8, 16):
            ms = timeit(lambda: ops.gemm_tn(g, x, dw, db=db, nsplit=ns))
            print(f"  tn {M}x{Nn}x{Kk} nsplit={ns}: {ms:.3f} ms  {2*M*Nn*Kk/ms/1e9:.0f} TF/s")
        ms = timeit(lambda: ops.gemm_tn(g, x, dw, nsplit=8))
        print(f"  tn {M}x{Nn}x{Kk} nsplit=8 no-db: {ms:.3f} ms  {2*M*Nn*Kk/ms/1e9:.0f} TF/s")


if __name__ == "__main__":
    main()
