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


class GroupedPyramidExperts:
    """All E pyramid experts of a batch in ONE launch sequence (reference swin.py:83-108 computes every expert on every sample and gathers the
    selected one; `PyramidExpert` above computes one expert on its samples): the samples are sorted by their selected expert, every per-expert
    GEMM becomes a grouped GEMM over row ranges (tile tables (expert, first row, end row) / row offsets per scale, built on the host from the
    router's top-1 indices ON THE DEVICE by medmoe_dispatch - no host read), the scale attention takes the expert of each sample slot.  Parameters and
    gradients are the stacked views of a `FlatStore` that lays the experts' tensors out back to back (`GroupedPyramidExperts.groups`)."""

    SCALES = 4

    @staticmethod
    def groups(n_expert: int, prefix: str = "moe.experts."):
        """FlatStore `groups` (alias, members) that make the stacked [E, ...] views contiguous."""
        g = []
        for s in range(GroupedPyramidExperts.SCALES):
            g.append((f"moe.stack.proj{s}.weight", [f"{prefix}{e}.proj_convs.{s}.0.weight" for e in range(n_expert)]))
            g.append((f"moe.stack.proj{s}.bias", [f"{prefix}{e}.proj_convs.{s}.0.bias" for e in range(n_expert)]))
        for nm in ("attn_proj.0.weight", "attn_proj.0.bias", "attn_proj.2.weight", "attn_proj.2.bias"):
            g.append((f"moe.stack.{nm}", [f"{prefix}{e}.{nm}" for e in range(n_expert)]))
        return g

    def __init__(self, store: FlatStore, n_expert: int, device):
        self.store, self.E, self.dev = store, n_expert, torch.device(device)
        E = n_expert
        sh = store.shapes
        self.Do = sh["moe.experts.0.attn_proj.0.weight"][1]
        self.Dh = sh["moe.experts.0.attn_proj.0.weight"][0]
        self.Ds = [sh[f"moe.experts.0.proj_convs.{s}.0.weight"][1] for s in range(4)]
        Do, Dh = self.Do, self.Dh
        v = store._view
        self.wp = [v(store.p16, f"moe.stack.proj{s}.weight", (E, Do, self.Ds[s])) for s in range(4)]
        self.wpt = [v(store.p16t, f"moe.stack.proj{s}.weight", (E, self.Ds[s], Do)) for s in range(4)]
        self.bp = [v(store.p32, f"moe.stack.proj{s}.bias", (E, Do)) for s in range(4)]
        self.w0, self.w0t = v(store.p16, "moe.stack.attn_proj.0.weight", (E, Dh, Do)), v(store.p16t, "moe.stack.attn_proj.0.weight", (E, Do, Dh))
        self.b0 = v(store.p32, "moe.stack.attn_proj.0.bias", (E, Dh))
        self.w2, self.b2 = v(store.p32, "moe.stack.attn_proj.2.weight", (E, Dh)), v(store.p32, "moe.stack.attn_proj.2.bias", (E,))
        self.wgrad_stream = None
        self._ones = None
        self._tab_key, self._tab_bufs = None, []

    def _grads(self):
        st, E, Do, Dh, v = self.store, self.E, self.Do, self.Dh, self.store._view
        return dict(wp=[v(st.g32, f"moe.stack.proj{s}.weight", (E, Do, self.Ds[s])) for s in range(4)],
                    bp=[v(st.g32, f"moe.stack.proj{s}.bias", (E, Do)) for s in range(4)],
                    w0=v(st.g32, "moe.stack.attn_proj.0.weight", (E, Dh, Do)), b0=v(st.g32, "moe.stack.attn_proj.0.bias", (E, Dh)),
                    w2=v(st.g32, "moe.stack.attn_proj.2.weight", (E, Dh)), b2=v(st.g32, "moe.stack.attn_proj.2.bias", (E,)))

    def _tables(self, idx32: torch.Tensor, B: int, Ps: List[int]):
        """Per scale, ON THE DEVICE (medmoe_dispatch, the ViT engine's router dispatch, run once per token count): the samples' order by
        expert, the expert of every sorted sample, 128-row and 256-row tile tables (expert, first row, end row of the expert's range, 0) and
        the row offsets [E + 1] of the grouped wgrad.  No host read: the step issues without waiting for the tower."""
        E, dev = self.E, self.dev
        key = (B, tuple(Ps))
        if self._tab_key != key:
            self._tab_bufs = []
            for P in Ps:
                mt = (B * P + 127) // 128 + E
                z = lambda *shape: torch.zeros(*shape, device=dev, dtype=torch.int32)
                self._tab_bufs.append(dict(P=P, mt=mt, slot_of=z(B), item=z(B), eos=z(B), row_off=z(E + 1), tiles=z(2 * mt, 4), count=z(2), rowmap=z(B * P)))
            self._tab_key = key
        out = []
        for t in self._tab_bufs:
            P, mt = t["P"], t["mt"]
            ops.call("dispatch", idx32, B, 1, E, P, P + 1, t["slot_of"], t["item"], t["eos"], t["row_off"], t["tiles"], t["count"], mt, t["rowmap"])
            out.append({128: dict(tiles=t["tiles"], tile_count=t["count"], max_tiles=mt),
                        256: dict(tiles=t["tiles"][mt:], tile_count=t["count"][1:], max_tiles=(B * P + 255) // 256 + E),
                        "row_off": t["row_off"], "item": t["item"], "eos": t["eos"]})
        return out

    def _grp(self, tab, K, M):
        """Tile-table arguments of a grouped NT GEMM with reduction length K: the 256-row kernel needs K >= 128 and K % 64 == 0."""
        if K >= 128 and K % 64 == 0:
            return dict(M=M, tile_rows=256, **tab[256])
        return dict(M=M, **tab[128])

    def forward(self, hs: List[torch.Tensor], top: torch.Tensor) -> torch.Tensor:
        """hs: 4 x bf16 [B, P_s, D_s]; top: int32 [B] (or [B, 1]) selected expert per sample, on the device.  Returns bf16 [B, P, Do] in the
        caller's sample order."""
        E, Do, Dh, dev = self.E, self.Do, self.Dh, self.dev
        B = hs[0].shape[0]
        Ps = [h.shape[1] for h in hs]
        P = max(Ps)
        R = B * P
        self.tab = self._tables(top.to(torch.int32).reshape(B).contiguous(), B, Ps)
        self.tabP = self.tab[Ps.index(P)]
        self.order32 = self.tabP["item"]                             # sorted slot -> sample (stable: ascending sample index within an expert)
        order = self.order = self.order32.long()
        self.B, self.P, self.Ps = B, P, Ps
        self.slot_e = self.tabP["eos"]
        self.fs = [h.index_select(0, order) for h in hs]             # the stage features, samples sorted by expert
        self.G = torch.empty(4, R, Do, device=dev, dtype=BF); self.H1 = torch.empty(4, R, Dh, device=dev, dtype=BF)
        self.small = []
        for s in range(4):
            Psn, Ds = Ps[s], self.Ds[s]
            g = self.G[s] if Psn == P else torch.empty(B * Psn, Do, device=dev, dtype=BF)
            ops.gemm_nt(self.fs[s].view(B * Psn, Ds), self.wp[s], g, bias=self.bp[s], stride_b=Do * Ds, stride_bias=Do, epi=ops.EPI_RELU,
                        **self._grp(self.tab[s], Ds, B * Psn))                                                  # swin.py:41
            if Psn != P:
                ops.call("lerp_tokens_fwd", g, self.G[s], B, Psn, P, Do)                                       # swin.py:42
            self.small.append(g)
            ops.gemm_nt(self.G[s], self.w0, self.H1[s], bias=self.b0, stride_b=Dh * Do, stride_bias=Dh, epi=ops.EPI_RELU,
                        **self._grp(self.tabP, Do, R))                                                         # swin.py:25-27,62
        self.eout = torch.empty(R, Do, device=dev, dtype=BF); self.wts = torch.empty(R, 4, device=dev)
        ops.call("scale_attn_fwd", self.G, self.H1, self.w2, self.b2, self.slot_e, P, self.eout, self.wts, R, Do, Dh)   # swin.py:62-80
        out = torch.empty(B, P, Do, device=dev, dtype=BF)
        out.index_copy_(0, order, self.eout.view(B, P, Do))
        return out

    def backward(self, d_local: torch.Tensor, d_global: Optional[torch.Tensor]) -> List[torch.Tensor]:
        """d_local bf16 [B, P, Do] (caller's sample order), d_global fp32 [B, Do] or None (the gradient of the token mean: added as
        d_global / P to every token inside the kernel) -> gradients w.r.t. hs (4 x bf16 [B, P_s, D_s], caller's order); the parameter
        gradients are ADDED to the store's gradient arena (the stacked views)."""
        E, Do, Dh, dev, B, P, Ps = self.E, self.Do, self.Dh, self.dev, self.B, self.P, self.Ps
        R = B * P
        g = self._grads()
        dG = torch.empty(4, R, Do, device=dev, dtype=BF); dH1 = torch.empty(4, R, Dh, device=dev, dtype=BF)
        if self._ones is None or self._ones.numel() != B:
            self._ones = torch.ones(B, device=dev)
        ops.call("scale_attn_bwd", d_local.contiguous(), d_global, self.G, self.H1, self.wts, self.w2, self.eout, self.slot_e, self.order32,
                 self._ones, 1, P, dG, dH1, g["w2"], g["b2"], None, R, Do, Dh)
        out = []
        dT = torch.empty(R, Do, device=dev, dtype=BF)
        for s in range(4):
            Psn, Ds = Ps[s], self.Ds[s]
            ops.gemm_tn(dH1[s], self.G[s], g["w0"], db=g["b0"], row_off=self.tabP["row_off"], n_groups=E, stride_w=Dh * Do, stride_db=Dh,
                        nsplit=4, M=R, stream=fork_wgrad(self.wgrad_stream, dH1, self.G))
            # the hidden layer's dgrad into its own buffer (a read-modify-write of dG cost a 154 MB residual read per scale: 211 us against
            # ~100); the consumer sums the two parts while it interpolates^T and applies ReLU' (identity interpolation at the finest scale)
            ops.gemm_nt(dH1[s], self.w0t, dT, stride_b=Dh * Do, **self._grp(self.tabP, Dh, R))
            dsm = torch.empty(B * Psn, Do, device=dev, dtype=BF)
            ops.call("lerp_tokens_bwd2", dG[s], dT, self.small[s], dsm, B, Psn, P, Do)
            fsv = self.fs[s].view(B * Psn, Ds)
            ops.gemm_tn(dsm, fsv, g["wp"][s], db=g["bp"][s], row_off=self.tab[s]["row_off"], n_groups=E, stride_w=Do * Ds, stride_db=Do,
                        nsplit=4, M=B * Psn, stream=fork_wgrad(self.wgrad_stream, dsm, fsv))
            df = torch.empty(B * Psn, Ds, device=dev, dtype=BF)
            ops.gemm_nt(dsm, self.wpt[s], df, stride_b=Do * Ds, **self._grp(self.tab[s], Do, B * Psn))
            d = torch.empty(B, Psn, Ds, device=dev, dtype=BF)
            d.index_copy_(0, self.order, df.view(B, Psn, Ds))
            out.append(d)
        return out
