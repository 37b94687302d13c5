#!/usr/bin/env python3
"""Runs a few launches of one GEMM shape (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops, load_library
import ctypes
kind = sys.argv[1] if len(sys.argv) > 1 else "nt256"
M, N, K = 50432, 3072, 768
bf = torch.bfloat16
a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
c = torch.empty(M, N, device="cuda", dtype=bf)
if kind == "nt128":
    load_library().medmoe_set_option(ctypes.c_int(1), ctypes.c_int(0))
if kind in ("nt256", "nt128"):
    for _ in range(5):
        ops.gemm_nt(a, b, c)
else:
    g = torch.randn(M, N, device="cuda").to(bf); dw = torch.zeros(N, K, device="cuda")
    for _ in range(5):
        ops.gemm_tn(g, a, dw, nsplit=8)
torch.cuda.synchronize()
