#!/usr/bin/env python3
"""Resident attention at 197 tokens (13 key tiles): 512-thread workgroups (eight waves, 13 query blocks in two uneven rounds) against 448 (seven waves x two
blocks) - medmoe_set_option(15, threads).  B = 1024 and 128, 12 heads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops

for B in (1024, 128):
    N, H = 197, 12
    D = H * 64
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B * H * N, device="cuda")
    dout = torch.randn(B * N, D, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H * N, device="cuda")
    ref = None
    for rt in (512, 448, 512, 448):
        ops.set_option(15, rt)
        line = f"B {B} threads {rt}:"
        for name, fn in (("fwd", lambda: ops.attn_fwd(qkv, out, lse, None, B, N, H)), ("bwd", lambda: ops.attn_bwd(qkv, out, dout, lse, None, dqkv, delta, B, N, H))):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            line += f"  {name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us"
        cur = (out.clone(), dqkv.clone())
        if ref is None:
            ref = cur
        else:
            line += f"  same output {torch.equal(ref[0], cur[0])} same dqkv {torch.equal(ref[1], cur[1])}"
        print(line, flush=True)
ops.set_option(15, 0)
