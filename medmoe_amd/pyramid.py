"""Pyramid-geometry expert (reference swin.py:11-80 with unequal token counts per scale, the reference's own Swin stages:
3136 / 784 / 196 / 49 tokens of 96 / 192 / 384 / 768 channels): per scale Conv1d(k=1) + ReLU on the MFMA GEMM, linear interpolation
to the longest scale (`medmoe_lerp_tokens_*`), then the same fused scale-attention as the equal-length geometry.
First slice of SURVEY.md 8(f) rank 4: ONE expert, forward and backward, checked against the reference fixture
tests/golden/expert_pyramid_mfma.npz.  (The ViT towers of BASELINE.json have equal token counts per stage: the engine elides the
interpolation there.)"""
from typing import Dict, List

import torch

from . import ops

BF = torch.bfloat16


class PyramidExpert:
    def __init__(self, weights: Dict[str, torch.Tensor], device="cuda:0"):
        """weights: reference names (proj_convs.{s}.0.weight [Do, D_s, 1], .bias, attn_proj.0.weight [Dh, Do], .bias,
        attn_proj.2.weight [1, Dh], .bias [1]), fp32."""
        dev = torch.device(device)
        self.wp = [weights[f"proj_convs.{s}.0.weight"].reshape(weights[f"proj_convs.{s}.0.weight"].shape[0], -1).to(dev) for s in range(4)]
        self.bp = [weights[f"proj_convs.{s}.0.bias"].float().to(dev).contiguous() for s in range(4)]
        self.w0 = weights["attn_proj.0.weight"].to(dev); self.b0 = weights["attn_proj.0.bias"].float().to(dev).contiguous()
        self.w2 = weights["attn_proj.2.weight"].reshape(1, -1).float().to(dev).contiguous()
        self.b2 = weights["attn_proj.2.bias"].reshape(1).float().to(dev).contiguous()
        self.Do, self.Dh = self.w0.shape[1], self.w0.shape[0]
        self.wp16 = [w.to(BF).contiguous() for w in self.wp]; self.wp16t = [w.t().to(BF).contiguous() for w in self.wp]
        self.w016 = self.w0.to(BF).contiguous(); self.w016t = self.w0.t().to(BF).contiguous()
        self.dev = dev

    def forward(self, feats: List[torch.Tensor]) -> torch.Tensor:
        """feats: 4 x bf16 [n, P_s, D_s] -> [n, P, Do] with P = max P_s."""
        n = feats[0].shape[0]
        P = max(f.shape[1] for f in feats)
        R, Do, Dh, dev = n * P, self.Do, self.Dh, self.dev
        self.feats, self.n, self.P = feats, n, P
        self.G = torch.empty(4, R, Do, device=dev, dtype=BF); self.H1 = torch.empty(4, R, Dh, device=dev, dtype=BF)
        self.small = []
        for s, f in enumerate(feats):
            Ps = f.shape[1]
            g = self.G[s] if Ps == P else torch.empty(n * Ps, Do, device=dev, dtype=BF)
            ops.gemm_nt(f.reshape(n * Ps, -1), self.wp16[s], g, bias=self.bp[s], epi=ops.EPI_RELU)             # swin.py:41
            if Ps != P:
                ops.call("lerp_tokens_fwd", g, self.G[s], n, Ps, P, Do)                                        # swin.py:42
            self.small.append(g)
            ops.gemm_nt(self.G[s], self.w016, self.H1[s], bias=self.b0, epi=ops.EPI_RELU)                      # swin.py:25-27,62
        self.eout = torch.empty(R, Do, device=dev, dtype=BF); self.wts = torch.empty(R, 4, device=dev)
        self.slot_e = torch.zeros(n, device=dev, dtype=torch.int32)
        ops.call("scale_attn_fwd", self.G, self.H1, self.w2, self.b2, self.slot_e, P, self.eout, self.wts, R, Do, Dh)   # swin.py:62-80
        return self.eout.view(n, P, Do)

    def backward(self, dy: torch.Tensor):
        """dy bf16 [n, P, Do] -> (input gradients 4 x [n, P_s, D_s] bf16, parameter gradients dict fp32, reference names)."""
        n, P, Do, Dh, dev = self.n, self.P, self.Do, self.Dh, self.dev
        R = n * P
        dG = torch.empty(4, R, Do, device=dev, dtype=BF); dH1 = torch.empty(4, R, Dh, device=dev, dtype=BF)
        g = {"attn_proj.2.weight": torch.zeros(1, Dh, device=dev), "attn_proj.2.bias": torch.zeros(1, device=dev),
             "attn_proj.0.weight": torch.zeros(Dh, Do, device=dev), "attn_proj.0.bias": torch.zeros(Dh, device=dev)}
        item = torch.arange(n, device=dev, dtype=torch.int32); gates = torch.ones(n, device=dev)
        ops.call("scale_attn_bwd", dy.reshape(n, P, Do).contiguous(), None, self.G, self.H1, self.wts, self.w2, self.eout, self.slot_e, item, gates,
                 1, P, dG, dH1, g["attn_proj.2.weight"], g["attn_proj.2.bias"], None, R, Do, Dh)
        dfeats = []
        for s, f in enumerate(self.feats):
            Ps, Ds = f.shape[1], f.shape[2]
            ops.gemm_tn(dH1[s], self.G[s], g["attn_proj.0.weight"], db=g["attn_proj.0.bias"])
            ops.gemm_nt(dH1[s], self.w016t, dG[s], residual=dG[s])                           # gradient w.r.t. the interpolated projection
            if Ps == P:
                dsm = dG[s]
                dsm.mul_((self.small[s] > 0).to(BF))                                          # equal length: plain ReLU' (engine path fuses it)
            else:
                dsm = torch.empty(n * Ps, Do, device=dev, dtype=BF)
                ops.call("lerp_tokens_bwd", dG[s], self.small[s], dsm, n, Ps, P, Do)          # interpolate^T, then ReLU' of the projection
            gw = torch.zeros(Do, Ds, device=dev); gb = torch.zeros(Do, device=dev)
            ops.gemm_tn(dsm, f.reshape(n * Ps, Ds), gw, db=gb)
            g[f"proj_convs.{s}.0.weight"], g[f"proj_convs.{s}.0.bias"] = gw.view(Do, Ds, 1), gb
            df = torch.empty(n * Ps, Ds, device=dev, dtype=BF)
            ops.gemm_nt(dsm, self.wp16t[s], df)
            dfeats.append(df.view(n, Ps, Ds))
        return dfeats, g
