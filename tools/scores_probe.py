#!/usr/bin/env python3
"""The transposed score GEMM (scores512_kernel<NTT, true>) alone at cfg2 / batch 1024 class sizes (lengths uniform in 8..77), image-major
output, with the measurement switches of medmoe_set_option(12): 0 full kernel, 1 no epilogue, 2 epilogue arithmetic without its stores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops

BF = torch.bfloat16
B = int(os.environ.get("PROBE_B", "1024"))
P, HWq, T, D = 196, 208, 77, 768
dev = "cuda"
torch.manual_seed(0)
lens = torch.randint(8, 78, (B,))
ctx = (torch.randn(B * P, D, device=dev) * 0.3).to(BF)
words = (torch.randn(B, T, D, device=dev) * 0.3).to(BF)
capd = lens.int().to(dev)
lse = torch.empty(B * 208, B, device=dev)
MODES = [int(v) for v in os.environ.get("PROBE_MODES", "0,2,1").split(",")]
tot = {m: 0.0 for m in MODES}
for ntt in range(1, 6):
    members = torch.nonzero((lens + 15) // 16 == ntt).flatten().int().to(dev)
    n_c = int(members.numel())
    if not n_c:
        continue
    Kp = n_c * 16 * ntt
    X = torch.empty(B, Kp, HWq, device=dev, dtype=BF)
    flop = 2.0 * B * P * Kp * D
    line = f"ntt {ntt} ({n_c:4d} captions):"
    for mode in MODES:
        ops.set_option(12, mode)
        f = lambda: ops.call("local_scores_t", ctx, words, capd, X, lse, B, B, P, T, D, members, n_c, ntt, 0, HWq, Kp * HWq)
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            f()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3
        tot[mode] += ms
        line += f"  mode {mode}: {ms:6.2f} ms {flop / ms / 1e9:6.0f} TF"
    print(line, flush=True)
ops.set_option(12, 0)
print("totals:", {m: round(v, 2) for m, v in tot.items()})
