#!/usr/bin/env python3
"""Measurement aid: gemm_nt4w (4 waves, 128x128 wave tiles, hand-interleaved) against gemm_nt512 and the vendor library."""
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


lib = ops.load_library()
bf = torch.bfloat16
shapes = [(1024, 512, 128), (2048, 768, 256), (2000, 1000, 192), (201728, 2304, 768), (201728, 768, 768), (201728, 3072, 768), (201728, 768, 3072),
          (50432, 3072, 768), (16384, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf); c2 = torch.empty(M, N, device="cuda", dtype=bf)
    lib.medmoe_set_option(7, 0)
    ops.gemm_nt(a, b, c)
    ms0 = timeit(lambda: ops.gemm_nt(a, b, c))
    lib.medmoe_set_option(7, 1)
    ops.gemm_nt(a, b, c2)
    torch.cuda.synchronize()
    same = torch.equal(c, c2)
    err = 0.0
    if M * N <= 8192 * 8192:
        ref = a.float() @ b.float().t()
        err = ((c2.float() - ref).norm() / ref.norm()).item()
    ms1 = timeit(lambda: ops.gemm_nt(a, b, c2))
    lib.medmoe_set_option(7, 0)
    ms2 = timeit(lambda: torch.matmul(a, b.t(), out=c))
    f = 2 * M * N * K / 1e9
    print(f"nt {M}x{N}x{K}: nt512 {f/ms0:.0f} | nt4w {f/ms1:.0f} | library {f/ms2:.0f} TF/s   bit-equal {same} rel-err {err:.2e}", flush=True)
    del a, b, c, c2
