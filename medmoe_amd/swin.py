"""Swin-T tower on the HIP kernels (second slice: the reference's image encoder is HF
`SwinModel('microsoft/swin-tiny-patch4-window7-224')` with `output_hidden_states=True`, reference swin.py:119-149; the algorithm is
transformers' modeling_swin.py: SwinEmbeddings / SwinLayer / SwinSelfAttention / SwinPatchMerging).  Forward AND backward, parameters under
the SwinModel state_dict names, so a reference checkpoint loads as it is.

    tower = SwinTower(SwinModel(cfg).state_dict())            # or a checkpoint's state dict
    out = tower.forward(images_bf16)                           # hidden_states[0..3] (3136x96, 784x192, 196x384, 49x768), last_hidden_state, pooled
    grads = tower.backward(d_hidden_states, d_last)            # fp32 gradients under the same names

Every Linear runs on `medmoe_gemm_nt` / `medmoe_gemm_tn` (stage 1's 96 / 288-wide contractions on the K % 64 == 32 tail of the 128x128 kernel),
LayerNorm on `medmoe_layernorm_*`, GELU and its derivative in the GEMM epilogues, the window attention on `medmoe_win_attn_*` (cyclic shift,
window partition / reverse and the shift mask are row arithmetic inside the kernel), patch merging on `medmoe_patch_merge`.  torch only allocates,
gathers the 169-entry bias tables into the kernel's [heads][64][64] form and scatters their gradient back.
Stochastic depth (`drop_path_rate` 0.1 in SwinConfig, active in train mode as the reference trains it): `forward(images, drop_path=masks)`
with the per-block keep masks of `sample_drop_path`; None = eval mode.
Not in this slice: the engine / train_step integration (the BASELINE configs name ViT towers)."""
import contextlib
from typing import Dict, List, Optional

import torch

from . import ops
from .flat import FlatStore

BF, F32 = torch.bfloat16, torch.float32


def fork_wgrad(stream, *inputs):
    """Order `stream` (a torch.cuda.Stream, or None) after everything enqueued so far on the current stream and return its raw handle for
    `ops.gemm_tn(..., stream=)`: weight-gradient GEMMs - nothing in the backward chain reads their result, and at batch 32 most of them fill
    a fraction of the chip - run underneath the dgrad chain.  The inputs are torch-allocated scratch: record_stream keeps the caching
    allocator from handing their memory to a later allocation before the side stream has read it.  None: everything stays on the current
    stream.  The caller joins with `join_side_stream` before anything reads the outputs."""
    if stream is None:
        return None
    h = stream.cuda_stream
    ops.stream_fork(ops.current_stream_handle(), h)
    for t in inputs:
        t.record_stream(stream)
    return h


def join_side_stream(stream):
    if stream is not None:
        ops.stream_fork(stream.cuda_stream, ops.current_stream_handle())


def relative_position_index(ws: int = 7) -> torch.Tensor:
    """modeling_swin.py SwinSelfAttention.__init__: index of the (2 ws - 1)^2 table for every (query, key) pair of a window."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


class SwinTower:
    def __init__(self, weights: Dict[str, torch.Tensor], device="cuda:0", depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), embed_dim=96,
                 image_size=224, patch=4, eps=1e-5):
        self.dev = torch.device(device)
        self.depths, self.heads, self.E, self.img, self.patch, self.eps = tuple(depths), tuple(heads), embed_dim, image_size, patch, eps
        self.res0 = image_size // patch
        if self.res0 % (7 * 2 ** (len(depths) - 1)) or any(embed_dim * 2 ** s != heads[s] * 32 for s in range(len(depths))):
            raise ValueError("SwinTower: the window-attention kernel takes 7x7 windows and heads of 32 channels (Swin-T/S/B geometry)")
        fw = {k: v for k, v in weights.items() if v.dtype.is_floating_point}
        groups, gemm = [], []
        for s, depth in enumerate(self.depths):
            for i in range(depth):
                pre = f"encoder.layers.{s}.blocks.{i}."
                groups.append((pre + "qkv", [pre + f"attention.{n}_proj.weight" for n in "qkv"]))      # one [3C, C] GEMM weight, one [3C] bias
                groups.append((pre + "qkv_b", [pre + f"attention.{n}_proj.bias" for n in "qkv"]))
                gemm += [pre + "qkv"] + [pre + nm + ".weight" for nm in ("attention.o_proj", "mlp.fc1", "mlp.fc2")]
            if s + 1 < len(self.depths):
                gemm.append(f"encoder.layers.{s}.downsample.reduction.weight")
        self.store = st = FlatStore(fw, self.dev, groups, gemm)
        self.w = {k: st.f32(k) for k in fw}                         # fp32 masters: views of the flat buffer (change in place, then refresh())
        self.index = relative_position_index().to(self.dev)
        self.wgrad_stream = None                                    # a torch.cuda.Stream: weight-gradient GEMMs run on it (joined at the end of backward)
        c = {}
        for s, depth in enumerate(self.depths):
            for i in range(depth):
                pre = f"encoder.layers.{s}.blocks.{i}."
                c[pre + "qkv"], c[pre + "qkv_t"], c[pre + "qkv_b"] = st.w16(pre + "qkv"), st.w16t(pre + "qkv"), st.f32(pre + "qkv_b")
                for nm in ("attention.o_proj", "mlp.fc1", "mlp.fc2"):
                    c[pre + nm], c[pre + nm + "_t"] = st.w16(pre + nm + ".weight"), st.w16t(pre + nm + ".weight")
                b = torch.zeros(self.heads[s], 64, 64, device=self.dev)
                b[:, :, 49:] = -30000.0                             # padding keys of the 64-token tile
                c[pre + "bias"] = b
            if s + 1 < len(self.depths):
                r = f"encoder.layers.{s}.downsample.reduction.weight"
                c[f"red{s}"], c[f"red{s}_t"] = st.w16(r), st.w16t(r)
        pw = self.w["embeddings.patch_embeddings.projection.weight"].reshape(self.E, -1)
        self.kp = (pw.shape[1] + 63) // 64 * 64
        c["pe"] = torch.zeros(self.E, self.kp, device=self.dev, dtype=BF)
        self.c = c
        self.refresh()

    # ------------------------------------------------------------------------------------------------------------
    def refresh(self, cast: bool = True):
        """bf16 working copies of the GEMM weights ([out, in] for forward, [in, out] for dgrad) after self.w changed: one cast and one batched
        transpose over the flat buffer, the zero-padded patch-embedding matrix, and the [heads][64][64] forms of the 169-entry bias tables.
        cast=False: the flat copies are current already (the fused Adam kernel wrote them), only the derived forms are rebuilt."""
        if cast:
            self.store.refresh()
        pw = self.w["embeddings.patch_embeddings.projection.weight"].reshape(self.E, -1)
        self.c["pe"][:, :pw.shape[1]].copy_(pw)
        for s, depth in enumerate(self.depths):
            for i in range(depth):
                pre = f"encoder.layers.{s}.blocks.{i}."
                t = self.w[pre + "attention.relative_position_bias.relative_position_bias_table"]
                self.c[pre + "bias"][:, :49, :49].copy_(t[self.index.view(-1)].view(49, 49, self.heads[s]).permute(2, 0, 1))

    def _bias(self, pre: str, heads: int) -> torch.Tensor:
        return self.c[pre + "bias"]

    def _ln(self, x, name, save):
        y = torch.empty_like(x)
        mean = torch.empty(x.shape[0], device=self.dev); rstd = torch.empty_like(mean)
        ops.layernorm_fwd(x, self.w[name + ".weight"], self.w[name + ".bias"], y, mean, rstd, self.eps)
        save.update(x=x, mean=mean, rstd=rstd, name=name)
        return y

    # ------------------------------------------------------------------------------------------------------------
    def drop_path_rates(self, rate: float = 0.1) -> List[float]:
        """modeling_swin.py SwinEncoder: torch.linspace(0, drop_path_rate, sum(depths)), one probability per block."""
        n = sum(self.depths)
        return [rate * i / max(1, n - 1) for i in range(n)]

    def sample_drop_path(self, B: int, rate: float = 0.1, generator: Optional[torch.Generator] = None) -> List[Optional[torch.Tensor]]:
        """Train-mode masks: per block a [B] tensor of 0 / 1 (1 = keep), None where the probability is 0 (SwinDropPath.forward)."""
        out = []
        for pr in self.drop_path_rates(rate):
            out.append(None if pr == 0.0 else torch.floor(torch.rand(B, generator=generator) + (1.0 - pr)))
        return out

    def forward(self, images: torch.Tensor, drop_path: Optional[List[Optional[torch.Tensor]]] = None, drop_path_rate: float = 0.1) -> Dict[str, object]:
        """images: bf16 (or fp32) [B, 3, S, S], already normalised.  Returns hidden_states (list of bf16 [B, L_s, C_s]: the embedding output and
        the outputs of stages 1-3 after their patch merging, what the reference's MoE consumes), last_hidden_state [B, L_4, C_4], pooled.
        drop_path: None = eval mode; else the per-block keep masks of `sample_drop_path` (train mode, stochastic depth on the attention branch
        as modeling_swin.py SwinLayer.forward applies it: shortcut + mask / keep_prob * attention_output)."""
        dev, c, w = self.dev, self.c, self.w
        B = images.shape[0]
        R = self.res0
        self.B = B
        tape: List[dict] = []
        # ---- SwinEmbeddings: Conv2d(3, E, 4, 4) as im2col + GEMM, LayerNorm ----
        kp = c["pe"].shape[1]
        patches = self._patches if getattr(self, "_patches", None) is not None and self._patches.shape[0] == B * R * R else None
        if patches is None:                                         # the pad columns stay zero: allocated (and zeroed) once per batch size
            patches = self._patches = torch.zeros(B * R * R, kp, device=dev, dtype=BF)
        ops.call("patchify_ld", images.contiguous(), patches, B, 3, self.img, self.img, self.patch, 1 if images.dtype == F32 else 0, kp)
        proj = torch.empty(B * R * R, self.E, device=dev, dtype=BF)
        ops.gemm_nt(patches, c["pe"], proj, bias=w["embeddings.patch_embeddings.projection.bias"])
        emb_ln = {}
        x = self._ln(proj, "embeddings.norm", emb_ln)
        self.emb = dict(patches=patches, ln=emb_ln)
        hs = [x]
        C, res = self.E, R
        rates, blk = self.drop_path_rates(drop_path_rate), 0
        for s, depth in enumerate(self.depths):
            heads = self.heads[s]
            M = B * res * res
            for i in range(depth):
                pre = f"encoder.layers.{s}.blocks.{i}."
                shift = 0 if (i % 2 == 0 or res <= 7) else 3
                t = dict(pre=pre, C=C, res=res, heads=heads, shift=shift, ln1={}, ln2={})
                ln1 = self._ln(x, pre + "layernorm_before", t["ln1"])
                qkv = torch.empty(M, 3 * C, device=dev, dtype=BF)
                ops.gemm_nt(ln1, c[pre + "qkv"], qkv, bias=c[pre + "qkv_b"])
                bias = self._bias(pre, heads)
                att = torch.empty(M, C, device=dev, dtype=BF)
                lse = torch.empty(B * (res // 7) ** 2 * heads, 64, device=dev)
                ops.call("win_attn_fwd", qkv, bias, att, lse, B, res, res, C, heads, shift)
                x1 = torch.empty(M, C, device=dev, dtype=BF)
                mask = drop_path[blk] if drop_path is not None else None
                if mask is None:
                    ops.gemm_nt(att, c[pre + "attention.o_proj"], x1, bias=w[pre + "attention.o_proj.bias"], residual=x)
                else:                                            # x1 = x + mask_b / keep * (o_proj(att) + bias)
                    t["dp"] = (mask.to(dev, F32) / (1.0 - rates[blk])).contiguous()
                    branch = torch.empty(M, C, device=dev, dtype=BF)
                    ops.gemm_nt(att, c[pre + "attention.o_proj"], branch, bias=w[pre + "attention.o_proj.bias"])
                    ops.call("drop_path", branch, x, t["dp"], x1, B, res * res * C)
                blk += 1
                ln2 = self._ln(x1, pre + "layernorm_after", t["ln2"])
                h = torch.empty(M, 4 * C, device=dev, dtype=BF); dg = torch.empty_like(h)
                ops.gemm_nt(ln2, c[pre + "mlp.fc1"], h, bias=w[pre + "mlp.fc1.bias"], aux=dg, epi=ops.EPI_GELU_DAUX)      # h = GELU(z), dg = GELU'(z)
                x2 = torch.empty(M, C, device=dev, dtype=BF)
                ops.gemm_nt(h, c[pre + "mlp.fc2"], x2, bias=w[pre + "mlp.fc2.bias"], residual=x1)
                t.update(ln1_out=ln1, qkv=qkv, bias=bias, att=att, lse=lse, ln2_out=ln2, h=h, dg=dg)
                tape.append(t)
                x = x2
            if s + 1 < len(self.depths):
                merged = torch.empty(M // 4, 4 * C, device=dev, dtype=BF)
                ops.call("patch_merge", x, merged, B, res, res, C, 0)
                t = dict(down=s, C=C, res=res, ln={})
                lnm = self._ln(merged, f"encoder.layers.{s}.downsample.norm", t["ln"])
                x = torch.empty(M // 4, 2 * C, device=dev, dtype=BF)
                ops.gemm_nt(lnm, c[f"red{s}"], x)
                t["ln_out"] = lnm
                tape.append(t)
                C, res = 2 * C, res // 2
                hs.append(x)
        self.final_ln = {}
        last = self._ln(x, "layernorm", self.final_ln)
        self.tape, self.C_last, self.res_last = tape, C, res
        L = res * res
        return {"hidden_states": [h_.view(B, -1, h_.shape[-1]) for h_ in hs], "last_hidden_state": last.view(B, L, C),
                "pooled": last.view(B, L, C).float().mean(1)}

    # ------------------------------------------------------------------------------------------------------------
    def _ln_bwd(self, dy, saved, add=None):
        dx = torch.empty_like(saved["x"])
        name = saved["name"]
        gw, gb = self.store.grad(name + ".weight"), self.store.grad(name + ".bias")
        ops.layernorm_bwd(dy, saved["x"], saved["mean"], saved["rstd"], self.w[name + ".weight"], dx, gw, gb, add=add)
        return dx

    def _wgrad(self, g, x, wname, bname=None):
        """dW += g^T x, db += column sums of g, straight into the (zeroed) gradient arena; wname None: a scratch dW is returned instead."""
        Nn, Kk = g.shape[-1], x.shape[-1]
        dw = self.store.grad2d(wname) if wname is not None else torch.zeros(Nn, Kk, device=self.dev)
        db = self.store.grad(bname) if bname else None
        tiles = ((Nn + 127) // 128) * ((Kk + 127) // 128)            # small outputs: split the token rows over enough workgroups to fill the chip
        ops.gemm_tn(g, x, dw, db=db, nsplit=max(1, min(512 // tiles, g.shape[0] // 512)), stream=fork_wgrad(self.wgrad_stream, g, x))
        return dw, db

    def backward(self, d_hidden: Optional[List[Optional[torch.Tensor]]] = None, d_last: Optional[torch.Tensor] = None,
                 zero_grad: bool = True) -> Dict[str, torch.Tensor]:
        """d_hidden[s]: gradient w.r.t. hidden_states[s] (bf16 [B, L_s, C_s] or None), d_last: w.r.t. last_hidden_state.  Returns fp32 parameter
        gradients under the SwinModel state_dict names: views of the store's gradient arena (zeroed here; `self.store.new_grad_arena()` first
        when views of the previous step's gradients are still in use)."""
        dev, c, w, B = self.dev, self.c, self.w, self.B
        if zero_grad:                                               # False: accumulate onto the arena's contents (gradient accumulation)
            self.store.zero_grad()
        d_hidden = list(d_hidden or []) + [None] * 4
        if d_last is not None:
            dx = self._ln_bwd(d_last.reshape(-1, self.C_last).to(BF).contiguous(), self.final_ln)
        else:
            dx = torch.zeros(B * self.res_last ** 2, self.C_last, device=dev, dtype=BF)
        n_hs = 1 + sum(1 for t in self.tape if "down" in t)
        hs_i = n_hs - 1                                             # hidden_states index of the input of the stage being walked
        for t in reversed(self.tape):
            if "down" in t:
                s, C, res = t["down"], t["C"], t["res"]
                if d_hidden[hs_i] is not None:                      # the merged output is hidden_states[hs_i]
                    dx = dx + d_hidden[hs_i].reshape(dx.shape).to(BF)
                hs_i -= 1
                self._wgrad(dx, t["ln_out"], f"encoder.layers.{s}.downsample.reduction.weight")
                dlnm = torch.empty(dx.shape[0], 4 * C, device=dev, dtype=BF)
                ops.gemm_nt(dx, c[f"red{s}_t"], dlnm)
                dmerged = self._ln_bwd(dlnm, t["ln"])
                dx = torch.empty(dx.shape[0] * 4, C, device=dev, dtype=BF)
                ops.call("patch_merge", dmerged, dx, B, res, res, C, 1)
                continue
            pre, C, res, heads, shift = t["pre"], t["C"], t["res"], t["heads"], t["shift"]
            M = dx.shape[0]
            # x2 = x1 + fc2(GELU(fc1(LN2(x1))))
            self._wgrad(dx, t["h"], pre + "mlp.fc2.weight", pre + "mlp.fc2.bias")
            dz = torch.empty(M, 4 * C, device=dev, dtype=BF)
            ops.gemm_nt(dx, c[pre + "mlp.fc2_t"], dz, aux=t["dg"], epi=ops.EPI_MUL_AUX)
            self._wgrad(dz, t["ln2_out"], pre + "mlp.fc1.weight", pre + "mlp.fc1.bias")
            dln2 = torch.empty(M, C, device=dev, dtype=BF)
            ops.gemm_nt(dz, c[pre + "mlp.fc1_t"], dln2)
            dx1 = self._ln_bwd(dln2, t["ln2"], add=dx)
            # x1 = x + [mask / keep *] o_proj(window_attention(qkv(LN1(x))))
            dbr = dx1
            if "dp" in t:
                dbr = torch.empty_like(dx1)
                ops.call("drop_path", dx1, None, t["dp"], dbr, B, res * res * C)
            self._wgrad(dbr, t["att"], pre + "attention.o_proj.weight", pre + "attention.o_proj.bias")
            datt = torch.empty(M, C, device=dev, dtype=BF)
            ops.gemm_nt(dbr, c[pre + "attention.o_proj_t"], datt)
            dqkv = torch.empty(M, 3 * C, device=dev, dtype=BF)
            slabs = torch.empty(B * (res // 7) ** 2, heads, 64, 64, device=dev)         # dS of every (image, window, head)
            ops.call("win_attn_bwd", t["qkv"], t["bias"], datt, t["lse"], dqkv, slabs, B, res, res, C, heads, shift)
            tname = pre + "attention.relative_position_bias.relative_position_bias_table"
            # the bias table's gradient (sum of dS over the windows, scattered through the relative-position index) is parameter-gradient work:
            # on the second stream with the weight gradients, off the dgrad chain (three ATen launches per block, 100 MB of dS at stage 1)
            side = self.wgrad_stream
            fork_wgrad(side, slabs)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                dbias = slabs.sum(0)
                self.store.grad(tname).index_add_(0, self.index.view(-1), dbias[:, :49, :49].permute(1, 2, 0).reshape(-1, heads))
            self._wgrad(dqkv, t["ln1_out"], pre + "qkv", pre + "qkv_b")          # the q / k / v gradients are the row blocks of this one
            dln1 = torch.empty(M, C, device=dev, dtype=BF)
            ops.gemm_nt(dqkv, c[pre + "qkv_t"], dln1)
            dx = self._ln_bwd(dln1, t["ln1"], add=dx1)
        if d_hidden[0] is not None:
            dx = dx + d_hidden[0].reshape(dx.shape).to(BF)
        dproj = self._ln_bwd(dx, self.emb["ln"])
        dw, db = self._wgrad(dproj, self.emb["patches"], None, "embeddings.patch_embeddings.projection.bias")
        pw = w["embeddings.patch_embeddings.projection.weight"]
        join_side_stream(self.wgrad_stream)
        self.store.grad("embeddings.patch_embeddings.projection.weight").add_(dw[:, :pw[0].numel()].reshape(pw.shape))
        return self.store.grads()
