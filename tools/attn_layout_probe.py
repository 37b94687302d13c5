#!/usr/bin/env python3
"""Measurement aid: is attention bound by the strided head slices of the [M, 3*H*64] qkv layout?  The same number of
(batch, head) problems run as B x H heads of one tensor and as B*H single-head batches (each head's q|k|v rows contiguous)."""
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
N = 197
for (B, H) in [(1024, 12), (1024 * 12, 1)]:
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, device="cuda").to(bf)
    out = torch.empty(B * N, D, device="cuda", dtype=bf); lse = torch.empty(B * H * N, device="cuda")
    dout = torch.randn(B * N, D, device="cuda").to(bf); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H * N, device="cuda")
    tf = timeit(lambda: ops.attn_fwd(qkv, out, lse, None, B, N, H))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, None, dqkv, delta, B, N, H))
    print(f"B {B} H {H}: fwd {tf*1e3:.0f} us, bwd (dq + dkv) {tb*1e3:.0f} us", flush=True)
