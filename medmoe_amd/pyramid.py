"""Pyramid-geometry expert (reference swin.py:11-80 with unequal token counts per scale, the reference's own Swin stages:
3136 / 784 / 196 / 49 tokens of 96 / 192 / 384 / 768 channels): per scale Conv1d(k=1) + ReLU on the MFMA GEMM, linear interpolation
to the longest scale (`medmoe_lerp_tokens_*`), then the same fused scale-attention as the equal-length geometry.
First slice of SURVEY.md 8(f) rank 4: ONE expert, forward and backward, checked against the reference fixture
tests/golden/expert_pyramid_mfma.npz.  (The ViT towers of BASELINE.json have equal token counts per stage: the engine elides the
interpolation there.)"""
from typing import Dict, List, Optional

import torch

from . import ops
from .flat import FlatStore
from .swin import fork_wgrad, join_side_stream

BF = torch.bfloat16


class PyramidExpert:
    def __init__(self, weights: Dict[str, torch.Tensor], device="cuda:0", store: Optional[FlatStore] = None, prefix: str = ""):
        """weights: reference names (proj_convs.{s}.0.weight [Do, D_s, 1], .bias, attn_proj.0.weight [Dh, Do], .bias,
        attn_proj.2.weight [1, Dh], .bias [1]), fp32.  store / prefix: the parameters already live in a FlatStore under prefix + name (the
        encoder's arena, `gemm` listing the five GEMM weights); weights is ignored then."""
        dev = torch.device(device)
        self.own = store is None
        if store is None:
            store = FlatStore({k: v for k, v in weights.items()}, dev, gemm=[f"proj_convs.{s}.0.weight" for s in range(4)] + ["attn_proj.0.weight"])
        self.store, self.pre = store, prefix
        n = lambda k: prefix + k
        self.bp = [store.f32(n(f"proj_convs.{s}.0.bias")) for s in range(4)]
        self.b0 = store.f32(n("attn_proj.0.bias"))
        self.w2 = store.f32(n("attn_proj.2.weight")).view(1, -1)
        self.b2 = store.f32(n("attn_proj.2.bias")).view(1)
        self.wp16 = [store.w16(n(f"proj_convs.{s}.0.weight")) for s in range(4)]
        self.wp16t = [store.w16t(n(f"proj_convs.{s}.0.weight")) for s in range(4)]
        self.w016, self.w016t = store.w16(n("attn_proj.0.weight")), store.w16t(n("attn_proj.0.weight"))
        self.Do, self.Dh = self.w016.shape[1], self.w016.shape[0]
        self.dev = dev
        self.wgrad_stream = None                                    # as SwinTower.wgrad_stream; the owner joins it (standalone: backward does)

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
        st, pre = self.store, self.pre
        if self.own:
            st.zero_grad()                                          # a shared arena is zeroed by its owner
        g = {k: st.grad(pre + k) for k in ("attn_proj.2.weight", "attn_proj.2.bias", "attn_proj.0.weight", "attn_proj.0.bias")}
        item = torch.arange(n, device=dev, dtype=torch.int32); gates = torch.ones(n, device=dev)
        ops.call("scale_attn_bwd", dy.reshape(n, P, Do).contiguous(), None, self.G, self.H1, self.wts, self.w2, self.eout, self.slot_e, item, gates,
                 1, P, dG, dH1, g["attn_proj.2.weight"], g["attn_proj.2.bias"], None, R, Do, Dh)
        dfeats = []
        for s, f in enumerate(self.feats):
            Ps, Ds = f.shape[1], f.shape[2]
            ops.gemm_tn(dH1[s], self.G[s], g["attn_proj.0.weight"], db=g["attn_proj.0.bias"], stream=fork_wgrad(self.wgrad_stream, dH1, self.G))
            ops.gemm_nt(dH1[s], self.w016t, dG[s], residual=dG[s])                           # gradient w.r.t. the interpolated projection
            if Ps == P:
                dsm = dG[s]
                dsm.mul_((self.small[s] > 0).to(BF))                                          # equal length: plain ReLU' (engine path fuses it)
            else:
                dsm = torch.empty(n * Ps, Do, device=dev, dtype=BF)
                ops.call("lerp_tokens_bwd", dG[s], self.small[s], dsm, n, Ps, P, Do)          # interpolate^T, then ReLU' of the projection
            gw, gb = st.grad2d(pre + f"proj_convs.{s}.0.weight"), st.grad(pre + f"proj_convs.{s}.0.bias")
            ops.gemm_tn(dsm, f.reshape(n * Ps, Ds), gw, db=gb, stream=fork_wgrad(self.wgrad_stream, dsm, f))
            g[f"proj_convs.{s}.0.weight"], g[f"proj_convs.{s}.0.bias"] = st.grad(pre + f"proj_convs.{s}.0.weight"), gb
            df = torch.empty(n * Ps, Ds, device=dev, dtype=BF)
            ops.gemm_nt(dsm, self.wp16t[s], df)
            dfeats.append(df.view(n, Ps, Ds))
        if self.own:
            join_side_stream(self.wgrad_stream)
        return dfeats, g
