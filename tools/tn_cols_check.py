#!/usr/bin/env python3
"""Step-by-step check of medmoe_gemm_tn_cols against fp32 matmuls (prints before every launch, so a fault names its call)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmoe_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
g = torch.Generator(device=dev); g.manual_seed(0)
ri = lambda *s: torch.randint(-3, 4, s, device=dev, generator=g).to(BF)


def step(name, fn):
    print("launch", name, flush=True)
    fn(); torch.cuda.synchronize()
    print("   done", name, flush=True)


# (1) the plain wgrad path (gemm_tn4w_kernel<false>) through medmoe_gemm_tn
M, Nn, Kk = 8192, 512, 256
G, X = ri(M, Nn), ri(M, Kk)
dW = torch.zeros(Nn, Kk, device=dev)
step("plain gemm_tn", lambda: ops.gemm_tn(G, X, dW))
print("   exact:", torch.equal(dW, G.float().t() @ X.float()))
# (2) one group, partial last tile in N, M small (12 sub-stages)
M, Nn, Kk = 384, 1872, 768
G, X = ri(M, Nn), ri(M, Kk)
dW = torch.zeros(Nn, Kk, device=dev)
step("cols one group", lambda: ops.call("gemm_tn_cols", G, Nn, X, Kk, dW, Kk, M, Nn, Kk, 1, 0, 0, 0, 0, 0))
print("   exact:", torch.equal(dW, G.float().t() @ X.float()))
# (3) nine column groups of 208 x 208
B, HWp = 9, 208
U, A = ri(M, B * HWp), ri(M, B * HWp)
out = torch.zeros(B, HWp, HWp, device=dev)
step("cols nine groups", lambda: ops.call("gemm_tn_cols", U, B * HWp, A, B * HWp, out, HWp, M, HWp, HWp, B, HWp, HWp, HWp * HWp, 0, 0))
ref = torch.bmm(U.float().view(M, B, HWp).permute(1, 2, 0), A.float().view(M, B, HWp).permute(1, 0, 2))
print("   exact:", torch.equal(out, ref))
# (4) two sub-stages only
M = 64
U, A = ri(M, B * HWp), ri(M, B * HWp)
out = torch.zeros(B, HWp, HWp, device=dev)
step("cols M=64", lambda: ops.call("gemm_tn_cols", U, B * HWp, A, B * HWp, out, HWp, M, HWp, HWp, B, HWp, HWp, HWp * HWp, 0, 0))
ref = torch.bmm(U.float().view(M, B, HWp).permute(1, 2, 0), A.float().view(M, B, HWp).permute(1, 0, 2))
print("   exact:", torch.equal(out, ref))
# (5) rows more than 4 GB from the base (the transposed pair matrices of cfg2 at batch 1024 are 19 GB)
M, ld, Nn, Kk, NG = 5184, 425984, 208, 104, 2
print("alloc", flush=True)
big = torch.empty(M, ld, device=dev, dtype=BF); torch.cuda.synchronize()
cols = [1000, 1000 + 212992]
blocks = [ri(M, 256) for _ in cols]
print("fill", flush=True)
for c, blk in zip(cols, blocks):
    big[:, c:c + 256].copy_(blk)
torch.cuda.synchronize()
print("   readback:", all(torch.equal(big[:, c:c + 256], blk) for c, blk in zip(cols, blocks)), flush=True)
X = ri(M, 256)
out = torch.zeros(NG, Nn, Kk, device=dev); torch.cuda.synchronize()
step("cols wide rows", lambda: ops.call("gemm_tn_cols", big[:, 1000:], ld, X, 256, out, Kk, M, Nn, Kk, NG, 212992, 128, Nn * Kk, 0, 0))
print("   exact:", all(torch.equal(out[q], blocks[q][:, :Nn].float().t() @ X[:, q * 128: q * 128 + Kk].float()) for q in range(NG)))
# (6) chunked columns: nine images of 208 columns, 19968 elements apart, as ONE operand of 1872 columns (tiles span images)
M, B, HWp, D = 384, 9, 208, 768
Gm_ = ri(B, M, HWp)                      # image-major [B][M][HWp]
X = ri(M, D)
dW = torch.zeros(B * HWp, D, device=dev)
step("cols chunked", lambda: ops.call("gemm_tn_cols", Gm_, HWp, X, D, dW, D, M, B * HWp, D, 1, 0, 0, 0, HWp, M * HWp))
print("   exact:", torch.equal(dW.view(B, HWp, D), torch.einsum("bmh,md->bhd", Gm_.float(), X.float())))
