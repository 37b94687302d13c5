#!/usr/bin/env python3
"""Attention kernel timing (HIP events): resident (round 1) vs streaming kernels on the step's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops

def run(B, N, H, resident, masked=False, reps=10):
    D = H * 64
    qkv = (torch.randn(B * N, 3 * D, device="cuda") * 0.5).to(torch.bfloat16)
    out = torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B * H * N, device="cuda")
    dout = torch.randn(B * N, D, device="cuda").to(torch.bfloat16); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H * N, device="cuda")
    mask = None
    if masked:
        lens = torch.randint(8, N + 1, (B,), device="cuda")
        mask = (torch.arange(N, device="cuda")[None] < lens[:, None]).to(torch.uint8).contiguous()
    ops.set_option(11, resident)
    res = {}
    for name, fn in (("fwd", lambda: ops.attn_fwd(qkv, out, lse, mask, B, N, H)), ("bwd", lambda: ops.attn_bwd(qkv, out, dout, lse, mask, dqkv, delta, B, N, H))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / reps
    ops.set_option(11, 1)
    return res

for B, N, H, masked in ((1024, 197, 12, False), (1024, 77, 12, True), (256, 257, 16, False), (64, 577, 16, False)):
    for resident in ((1, 0) if N <= 592 else (0,)):
        r = run(B, N, H, resident, masked)
        fl = 4.0 * B * H * N * N * 64
        print(f"B {B} N {N} H {H} {'resident ' if resident else 'streaming'}: fwd {r['fwd']:.3f} ms ({fl / r['fwd'] / 1e9:.0f} TFLOP/s)  bwd (dq + dkv) {r['bwd']:.3f} ms ({2.5 * fl / r['bwd'] / 1e9:.0f} TFLOP/s)", flush=True)
