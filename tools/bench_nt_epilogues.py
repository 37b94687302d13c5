#!/usr/bin/env python3
"""Measurement aid: the fused-epilogue variants on gemm_nt4w (option 7 = 1) and gemm_nt512 (0), step shapes."""
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


bf = torch.bfloat16
for M in (201728, 25216):
    for (N, K, name) in [(3072, 768, "gelu_daux"), (3072, 768, "mul_aux"), (768, 3072, "mul_aux"), (768, 768, "bias_res"), (768, 3072, "bias_res"), (2304, 768, "bias"), (3072, 768, "none")]:
        a = torch.randn(M, K, device="cuda").to(bf); b = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
        c = torch.empty(M, N, device="cuda", dtype=bf); aux = torch.randn(M, N, device="cuda").to(bf); res = torch.randn(M, N, device="cuda").to(bf)
        bias = torch.randn(N, device="cuda")
        def run():
            if name == "gelu_daux": ops.gemm_nt(a, b, c, bias=bias, aux=aux, epi=ops.EPI_GELU_DAUX)
            elif name == "mul_aux": ops.gemm_nt(a, b, c, aux=aux, epi=ops.EPI_MUL_AUX)
            elif name == "bias_res": ops.gemm_nt(a, b, c, bias=bias, residual=res)
            elif name == "bias": ops.gemm_nt(a, b, c, bias=bias)
            else: ops.gemm_nt(a, b, c)
        ts = []
        timeit(run)                                    # the first measurement of a shape runs 5-8 % low (clocks / caches): discarded
        for opt in (1, 0):
            ops.set_option(7, opt)
            ts.append(timeit(run))
        ops.set_option(7, 1)
        f = 2.0 * M * N * K / 1e9
        print(f"M {M} N {N} K {K} {name:10s}: nt4w {f/ts[0]:.0f} | nt512 {f/ts[1]:.0f} TF/s", flush=True)
        del a, b, c, aux, res
