"""Mirror of the hot-path part of reference src/losses.py: same class / function names,
argument order, defaults and return containers, computed by medmoe_amd's HIP kernels.

  GLORIAGlobalContrastiveLoss        losses.py:757-794
  GLORIALocalContrastiveLoss         losses.py:954-1026 (+ attention_fn :698-736, cosine_similarity :690-695)
  contrastive_loss_with_temperature  losses.py:527-592  (+ _gather_embeddings_and_labels :503-524)
  SoftGLORIAGlobalContrastiveLoss    losses.py:814-883  } the two GLoRIA losses with the soft-label head: positives / negatives chosen per
  SoftGLORIALocalContrastiveLoss     losses.py:1111-1214 } row by a caption-to-caption score matrix and two thresholds (softXEnt :796-803)
  HardNegativeContrastiveLoss        losses.py:885-927   (margin loss on the nmax hardest in-batch negatives)
  ZEROGlobalContrastiveLoss / ZEROLocalContrastiveLoss   losses.py:740-755, 929-952 (ablation switches: the loss is 0)

Inputs must be CUDA tensors; gradients flow to the image-side inputs and to the text side (the local loss differentiates the word
embeddings for the 196- / 64-region geometries; the reference experiment freezes the text tower, med-moe.yaml:35).  Variants the reference config never selects
(FLAVA pretraining losses) are out of scope.
"""
import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, OrderedDict

import torch
from torch import Tensor, nn

from medmoe_amd import ops
from src.utils.distributed import BackpropType

DEFAULT_LOGIT_SCALE = math.log(1 / 0.07)


@dataclass
class ContrastiveLossOutput(OrderedDict):
    loss: Tensor
    logits_a: Tensor
    logits_b: Tensor
    loss_a: Tensor
    loss_b: Tensor


@dataclass
class GLORIALocalContrastiveLossOutput(OrderedDict):
    loss0: Tensor
    loss1: Tensor
    att_maps: List[Tensor]


_TL_CACHE: Dict[Any, Any] = {}      # (B, HW, T, D, device) -> TransposedLocalLoss: the local loss' buffers between calls


def _f32c(t: Tensor) -> Tensor:
    return t.detach().float().contiguous()


def _soft_args(idx, probs, B: int, dev):
    """(soft scores fp32 [B, B] on the device, threshold1, threshold2) of the reference's `idx` / `probs` arguments, or None."""
    if idx is None:
        return None
    if probs is None or len(probs) != 2:
        raise ValueError("Soft-GLoRIA: probs must be (threshold1, threshold2) (medmoe_module.py:292-293)")
    soft = torch.as_tensor(idx).detach().to(dev, torch.float32).contiguous()
    if soft.shape != (B, B):
        raise ValueError(f"Soft-GLoRIA: idx must be the [B, B] caption score matrix, got {tuple(soft.shape)}")
    return soft, float(probs[0]), float(probs[1])


def _head(S: Tensor, dS: Tensor, B: int, rs: int, cs: int, scale: float, accumulate: int, loss: Tensor, soft):
    """Cross-entropy against the diagonal (soft is None), the hard-negative margin head (soft = ("hardneg", margin)) or the Soft-GLoRIA
    head over the rows (rs, cs = B, 1) / columns (1, B) of S."""
    if soft is None:
        ops.call("ce_strided", S, dS, B, B, rs, cs, 0, scale, 1.0 / B, accumulate, loss)
    elif soft[0] == "hardneg":
        ops.call("hardneg_strided", S, dS, B, B, rs, cs, soft[1], 1.0, accumulate, loss)
    else:
        ops.call("soft_xent_strided", S, dS, soft[0], B, B, rs, cs, scale, soft[1], soft[2], 1.0 / B, accumulate, loss)


class _GloriaGlobalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img: Tensor, txt: Tensor, temp3: float, eps: float, soft=None):
        a, b = _f32c(img), _f32c(txt)
        B, D = a.shape
        dev = a.device
        na = torch.empty(B, device=dev); nb = torch.empty(B, device=dev)
        S = torch.empty(B, B, device=dev); dS = torch.empty(B, B, device=dev)
        loss = torch.zeros(1, device=dev)
        ops.call("rownorm", a, na, B, D); ops.call("rownorm", b, nb, B, D)
        ops.call("sgemm", a, b, S, B, B, D, D, 1, 1, D, B, 1.0, 0.0)
        ops.call("cos_scale", S, na, nb, B, B, eps)
        if soft is not None and soft[0] == "hardneg" and len(soft) > 2 and soft[2] > 1:
            # nmax > 1 (losses.py:909-911): the nmax hardest negatives per image and per caption.  The scores come from the kernels above; the
            # selection is torch.topk on the [B, B] matrix (B^2 floats), its gradient flows back into dS for the kernels below.
            with torch.enable_grad():
                Sg = S.detach().requires_grad_(True)
                diag = Sg.diag()
                Sm = Sg - 2.0 * torch.diag(diag)
                k = min(int(soft[2]), B)
                max_c = torch.topk(Sm, k, dim=0).values               # [k, B]: per caption column the hardest images
                max_i = torch.topk(Sm, k, dim=1).values               # [B, k]: per image row the hardest captions
                val = torch.clamp(max_c + (soft[1] - diag).view(1, -1), min=0).sum() + torch.clamp(max_i + (soft[1] - diag).view(-1, 1), min=0).sum()
                dS = torch.autograd.grad(val, Sg)[0].contiguous()
            loss = val.detach().reshape(1)
        else:
            _head(S, dS, B, B, 1, temp3, 0, loss, soft)
            _head(S, dS, B, 1, B, temp3, 1, loss, soft)
        ca = torch.empty(B, device=dev); cb = torch.zeros(B, device=dev)
        ops.call("cos_scale_bwd", dS, S, na, nb, ca, cb, B, B, eps)
        da = torch.empty(B, D, device=dev); db = torch.empty(B, D, device=dev)
        ops.call("sgemm", dS, b, da, B, D, B, B, 1, D, 1, D, 1.0, 0.0)
        ops.call("add_rowscaled", da, a, ca, B, D)
        ops.call("sgemm", dS, a, db, B, D, B, 1, B, D, 1, D, 1.0, 0.0)
        ops.call("add_rowscaled", db, b, cb, B, D)
        ctx.save_for_backward(da, db)
        ctx.dtypes = (img.dtype, txt.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        da, db = ctx.saved_tensors
        return (g * da).to(ctx.dtypes[0]), (g * db).to(ctx.dtypes[1]), None, None, None


class GLORIAGlobalContrastiveLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.eps = 1e-8
        self.temp3 = 10.0

    def forward(self, cnn_code: Tensor, rnn_code: Tensor, temp3: float = 10.0, idx: int = None,
                probs: Tensor = None) -> Tensor:
        if cnn_code.dim() != 2 or cnn_code.shape != rnn_code.shape:
            raise ValueError("GLORIAGlobalContrastiveLoss expects two [B, D] embeddings")
        return _GloriaGlobalFn.apply(cnn_code, rnn_code, float(temp3), self.eps, self._soft(idx, probs, cnn_code))

    def _soft(self, idx, probs, cnn_code):
        return None                                              # losses.py:766-794 ignores idx / probs


class SoftGLORIAGlobalContrastiveLoss(GLORIAGlobalContrastiveLoss):
    """losses.py:814-883: `idx` = caption-to-caption scores [B, B] of the frozen text model, `probs` = (threshold1, threshold2)."""

    def _soft(self, idx, probs, cnn_code):
        if idx is None:
            raise ValueError("SoftGLORIAGlobalContrastiveLoss needs idx (soft scores) and probs (thresholds)")
        return _soft_args(idx, probs, cnn_code.shape[0], cnn_code.device)


class HardNegativeContrastiveLoss(nn.Module):
    """losses.py:885-927 (the margin loss on the hardest in-batch negative, listed as an alternative global loss in
    med-moe_pretraining.yaml:33): cosine scores, per image the hardest caption and per caption the hardest image (the diagonal entry
    competes as its own negative: scores - 2 diag(diag)), relu(hardest + margin - positive), summed.  nmax = 1 (the reference default) runs
    the margin head kernel; nmax > 1 selects the nmax hardest per row / column with torch.topk over the kernel-computed score matrix."""

    def __init__(self, nmax: int = 1, margin: float = 0.2):
        super().__init__()
        if int(nmax) < 1:
            raise ValueError("HardNegativeContrastiveLoss: nmax >= 1")
        self.margin, self.nmax = margin, int(nmax)

    def forward(self, imgs: Tensor, caps: Tensor, temp3: float = 10.0, idx: int = None, probs: Tensor = None) -> Tensor:
        if imgs.dim() != 2 or imgs.shape != caps.shape:
            raise ValueError("HardNegativeContrastiveLoss expects two [B, D] embeddings")
        return _GloriaGlobalFn.apply(imgs, caps, 1.0, 1e-12, ("hardneg", float(self.margin), self.nmax))


class ZEROGlobalContrastiveLoss(nn.Module):
    """losses.py:740-755: the ablation switch `global loss = 0`."""

    def forward(self, cnn_code: Tensor, rnn_code: Tensor, temp3: float = 10.0, idx: int = None, probs: Tensor = None) -> Tensor:
        return torch.zeros((), device=cnn_code.device)


class _GloriaLocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img_features: Tensor, words_emb: Tensor, cap_lens, temp1, temp2, temp3, soft=None, agg_mean=False):
        B, D, H, W = img_features.shape
        HW, T = H * W, words_emb.shape[2]
        dev = img_features.device
        bf = torch.bfloat16
        ctx16 = img_features.detach().reshape(B, D, HW).transpose(1, 2).to(bf).contiguous().view(B * HW, D)
        w16 = words_emb.detach().transpose(1, 2).to(bf).contiguous()
        cap = torch.as_tensor(list(cap_lens) if not torch.is_tensor(cap_lens) else cap_lens, dtype=torch.int32).to(dev)
        want_w = bool(words_emb.requires_grad)
        if want_w and not (ops.local_pair3_supported(HW, T) and D % 64 == 0 and D >= 128 and words_emb.shape[0] == B):
            raise NotImplementedError("gradient w.r.t. the word embeddings: built for the 196- / 64-region geometries (csrc/pair3.hip) with a "
                                      f"width that is a multiple of 64; got {HW} regions, {T} words, width {D}")
        if not ops.local_fast_path(HW, T):
            # region counts without an LDS-tiled pair kernel (the Swin tower's 56 x 56 = 3136): the reference's own formulation as grouped
            # GEMMs (medmoe_amd/local_generic.py)
            from medmoe_amd.local_generic import GenericLocalLoss
            key = ("generic", B, HW, T, D, str(dev))
            gen = _TL_CACHE.get(key)                              # ten buffers of B*HWp x B*Tp: built once per geometry, not per step
            if gen is None:
                _TL_CACHE.clear()
                gen = _TL_CACHE[key] = GenericLocalLoss(B, HW, T, D, dev)
            sim = gen.forward(ctx16, w16, cap, temp1, temp2)
            if agg_mean:          # losses.py:1006-1009: log of the MEAN over a caption's words = log-sum minus log(#words), a per-caption constant
                sim = (sim - torch.log(cap.clamp(min=1, max=T).float())[None, :]).contiguous()
            g0 = torch.empty(B, B, device=dev); g1 = torch.empty(B, B, device=dev)
            l0 = torch.zeros(1, device=dev); l1 = torch.zeros(1, device=dev)
            _head(sim, g0, B, B, 1, temp3, 0, l0, soft)
            _head(sim, g1, B, 1, B, temp3, 0, l1, soft)
            att = gen.matching_attention_maps()                                                     # [B, T, HW]
            ctx.gen, ctx.generic, ctx.transposed, ctx.gen_generation = gen, True, False, gen.generation
            ctx.save_for_backward(g0, g1)
            ctx.geom = (B, D, H, W, img_features.dtype)
            return l0[0], l1[0], att
        ctx.generic = False
        if ops.local_pair3_supported(HW, T) and D % 32 == 0 and D >= 128 and words_emb.shape[0] == B:
            # 196 / 64 regions: the engine's own fast path - ragged transposed pair matrices, one wave per (image, caption, word tile)
            # (medmoe_amd/local_transposed.py); the instance and its buffers are kept for the next call of the same geometry
            from medmoe_amd.local_transposed import TransposedLocalLoss
            key = (B, HW, T, D, str(dev), want_w)
            tl = _TL_CACHE.get(key)
            if tl is None:
                _TL_CACHE.clear()                                 # one geometry at a time: the pair matrices are large
                tl = _TL_CACHE[key] = TransposedLocalLoss.standalone(B, HW, T, D, dev, word_grad=want_w)
            cap_host = cap.cpu().numpy() if torch.is_tensor(cap_lens) else [int(v) for v in cap_lens]
            att = torch.zeros(B, T, HW, device=dev)
            sim = tl.forward(ctx16, w16, cap, cap_host, temp1, temp2, att=att)
            if agg_mean:          # losses.py:1006-1009: log of the MEAN over a caption's words = log-sum minus log(#words), a per-caption constant
                sim = (sim - torch.log(cap.clamp(min=1, max=T).float())[None, :]).contiguous()
            g0 = torch.empty(B, B, device=dev); g1 = torch.empty(B, B, device=dev)
            l0 = torch.zeros(1, device=dev); l1 = torch.zeros(1, device=dev)
            _head(sim, g0, B, B, 1, temp3, 0, l0, soft)
            _head(sim, g1, B, 1, B, temp3, 0, l1, soft)
            ctx.tl, ctx.transposed, ctx.tl_gen, ctx.words_dtype = tl, True, tl.generation, words_emb.dtype
            ctx.save_for_backward(g0, g1)
            ctx.geom = (B, D, H, W, img_features.dtype)
            return l0[0], l1[0], att
        ctx.transposed = False
        HWp, Tp, GW = ops.local_geometry(HW, T)
        if (B * Tp) % 64 or D % 64:
            raise ValueError("GLORIALocalContrastiveLoss (HIP): B*ceil16(T) and D must be multiples of 64")
        i32 = torch.int32
        wn = torch.empty(B, T, device=dev); wT = torch.empty(D, B * Tp, device=dev, dtype=bf)
        ops.call("words_prep", w16, wn, wT, B, T, Tp, D)
        ar = torch.arange(B * HW, device=dev)
        tl = torch.tensor([[b, m, (b + 1) * HW, 0] for b in range(B) for m in range(b * HW, (b + 1) * HW, 128)], device=dev, dtype=i32)
        cnt = torch.tensor([tl.shape[0]], device=dev, dtype=i32)
        gmp = torch.zeros(B * HWp, GW, device=dev, dtype=bf)
        ops.gemm_nt(ctx16, ctx16, gmp, c_rowmap=(ar // HW * HWp + ar % HW).to(i32), tiles=tl, tile_count=cnt,
                    max_tiles=tl.shape[0], stride_b=HW * D, M=B * HW, N=HW, col_perm=True)
        sim = torch.empty(B, B, device=dev); att = torch.zeros(B, T, HW, device=dev)
        a1 = torch.empty(B * HWp, B * Tp, device=dev, dtype=bf); lse = torch.empty(B * HWp, B, device=dev)
        ops.call("local_scores", ctx16, w16, cap, a1, lse, B, B, HW, T, D)
        ops.call("local_pair", None, None, gmp, wn, cap, None, sim, None, None, None, att, a1, lse, B, B, HW, T, D, temp1, temp2, 1e-8, 0)
        if agg_mean:          # losses.py:1006-1009: log of the MEAN over a caption's words = log-sum minus log(#words), a per-caption constant
            sim = (sim - torch.log(cap.clamp(min=1, max=T).float())[None, :]).contiguous()
        g0 = torch.empty(B, B, device=dev); g1 = torch.empty(B, B, device=dev)
        l0 = torch.zeros(1, device=dev); l1 = torch.zeros(1, device=dev)
        _head(sim, g0, B, B, 1, temp3, 0, l0, soft)
        _head(sim, g1, B, 1, B, temp3, 0, l1, soft)
        ctx.save_for_backward(ctx16, w16, gmp, wn, cap, wT, g0, g1, a1, lse)
        ctx.geom = (B, D, H, W, T, HWp, Tp, temp1, temp2, img_features.dtype)
        return l0[0], l1[0], att

    @staticmethod
    def backward(ctx, gl0, gl1, _gatt):
        if ctx.generic:
            g0, g1 = ctx.saved_tensors
            B, D, H, W, dt = ctx.geom
            dctx = ctx.gen.backward((gl0 * g0 + gl1 * g1).contiguous(), ctx.gen_generation)
            return dctx.view(B, H * W, D).transpose(1, 2).reshape(B, D, H, W).to(dt), None, None, None, None, None, None, None
        if ctx.transposed:
            g0, g1 = ctx.saved_tensors
            B, D, H, W, dt = ctx.geom
            d_l = torch.empty(B, H * W, D, device=g0.device, dtype=torch.bfloat16)
            d_w = ctx.tl.backward((gl0 * g0 + gl1 * g1).contiguous(), d_l, ctx.tl_gen)
            if d_w is not None:                                   # [B, T, D] -> the reference's [B, D, T]
                d_w = d_w.transpose(1, 2).to(ctx.words_dtype)
            return d_l.transpose(1, 2).reshape(B, D, H, W).to(dt), d_w, None, None, None, None, None, None
        ctx16, w16, gmp, wn, cap, wT, g0, g1, a1, lse = ctx.saved_tensors
        B, D, H, W, T, HWp, Tp, temp1, temp2, dt = ctx.geom
        HW = H * W
        dev = ctx16.device
        bf, i32 = torch.bfloat16, torch.int32
        gsim = (gl0 * g0 + gl1 * g1).contiguous()
        dS = torch.empty(B * HWp, B * Tp, device=dev, dtype=bf); U = torch.empty_like(dS)
        A = a1                                                        # A tiles overwrite the A1 tiles in place
        ops.call("local_pair", None, None, gmp, wn, cap, gsim, None, dS, A, U, None, a1, lse, B, B, HW, T, D, temp1, temp2, 1e-8, 1)
        dC = torch.empty(B * HWp, D, device=dev)
        ops.gemm_nt(dS, wT, dC)
        tlp = torch.tensor([[b, m, (b + 1) * HWp, 0] for b in range(B) for m in range(b * HWp, (b + 1) * HWp, 128)], device=dev, dtype=i32)
        cntp = torch.tensor([tlp.shape[0]], device=dev, dtype=i32)
        dGm = torch.empty(B * HWp, HWp, device=dev, dtype=bf)
        ops.gemm_nt(U, A, dGm, tiles=tlp, tile_count=cntp, max_tiles=tlp.shape[0], stride_b=HWp * B * Tp, M=B * HWp, N=HWp)
        arp = torch.arange(B * HWp, device=dev)
        ops.gemm_tn(dGm, ctx16, dC.view(B, HWp, D), x_rowmap=(arp // HWp * HW + torch.clamp(arp % HWp, max=HW - 1)).to(i32),
                    row_off=(torch.arange(B + 1, device=dev) * HWp).to(i32), n_groups=B, stride_w=HWp * D, nsplit=1, M=B * HWp)
        d_img = dC.view(B, HWp, D)[:, :HW].transpose(1, 2).reshape(B, D, H, W).to(dt)
        return d_img, None, None, None, None, None, None, None


class GLORIALocalContrastiveLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, img_features: Tensor, words_emb: Tensor, cap_lens: List[float], temp1: float = 4.0,
                temp2: float = 5.0, temp3: float = 10.0, agg: str = "sum", idx: int = None,
                probs: Tensor = None) -> GLORIALocalContrastiveLossOutput:
        if agg not in ("sum", "mean"):
            raise ValueError("agg must be 'sum' (the reference default, losses.py:969) or 'mean'")
        loss0, loss1, att = _GloriaLocalFn.apply(img_features, words_emb, cap_lens, float(temp1), float(temp2), float(temp3),
                                                 self._soft(idx, probs, img_features), agg == "mean")
        B, D, H, W = img_features.shape
        maps = [att[i, : int(cap_lens[i])].reshape(1, int(cap_lens[i]), H, W) for i in range(B)]
        return GLORIALocalContrastiveLossOutput(loss0=loss0, loss1=loss1, att_maps=maps)

    def _soft(self, idx, probs, img_features):
        return None                                              # losses.py:961-1026 ignores idx / probs


class SoftGLORIALocalContrastiveLoss(GLORIALocalContrastiveLoss):
    """losses.py:1111-1214: the local similarities under the soft-label head (idx, probs as in SoftGLORIAGlobalContrastiveLoss)."""

    def _soft(self, idx, probs, img_features):
        if idx is None:
            raise ValueError("SoftGLORIALocalContrastiveLoss needs idx (soft scores) and probs (thresholds)")
        return _soft_args(idx, probs, img_features.shape[0], img_features.device)


class ZEROLocalContrastiveLoss(nn.Module):
    """losses.py:929-952: the ablation switch `local loss = 0`."""

    def forward(self, img_features: Tensor, words_emb: Tensor, cap_lens: List[float], temp1: float = 4.0, temp2: float = 5.0,
                temp3: float = 10.0, agg: str = "sum", idx: int = None, probs: Tensor = None) -> GLORIALocalContrastiveLossOutput:
        z = torch.zeros((), device=img_features.device)
        return GLORIALocalContrastiveLossOutput(loss0=z, loss1=z.clone(), att_maps=[])


class _ClipFn(torch.autograd.Function):
    """logits_a = a_loc b_all^T e^s ; logits_b = b_loc a_all^T e^s ; CE both ; mean (losses.py:558-584).
    Every returned tensor is differentiable: backward combines the incoming gradients of `loss`, `loss_a`,
    `loss_b`, `logits_a` and `logits_b` into one gradient per logits matrix and runs the four GEMMs once.
    Gathered-key gradients follow gather_tensor (distributed.py:28-58): GLOBAL = summed over ranks
    (reduce-scatter), LOCAL = this rank's slice only, NONE = none; without a process group the reference
    ignores backprop_type (losses.py:508-510: the "gathered" tensors ARE the inputs)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, logit_scale: Tensor, backprop_type):
        from medmoe_amd import dist as D_
        al, bl = _f32c(a), _f32c(b)
        B, D = al.shape
        dev = al.device
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        if distributed:
            a_all, b_all = D_.gather_embeddings(al, bl)
            off = D_.label_offset(B)
        else:
            a_all, b_all, off = al, bl, 0
        Bg = a_all.shape[0]
        t = torch.exp(logit_scale.detach().float()).reshape(())
        la = torch.empty(B, Bg, device=dev); lb = torch.empty(B, Bg, device=dev)
        tf = float(t)
        ops.call("sgemm", al, b_all, la, B, Bg, D, D, 1, 1, D, Bg, tf, 0.0)
        ops.call("sgemm", bl, a_all, lb, B, Bg, D, D, 1, 1, D, Bg, tf, 0.0)
        dla = torch.empty_like(la); dlb = torch.empty_like(lb)              # d loss_a / d logits_a, d loss_b / d logits_b
        loss_a = torch.zeros(1, device=dev); loss_b = torch.zeros(1, device=dev)
        ops.call("ce_strided", la, dla, B, Bg, Bg, 1, off, 1.0, 1.0 / B, 0, loss_a)
        ops.call("ce_strided", lb, dlb, B, Bg, Bg, 1, off, 1.0, 1.0 / B, 0, loss_b)
        ctx.save_for_backward(al, bl, a_all, b_all, la, lb, dla, dlb)
        ctx.meta = (tf, off, distributed, backprop_type, a.dtype, b.dtype, logit_scale.dtype, logit_scale.shape)
        return 0.5 * (loss_a[0] + loss_b[0]), la, lb, loss_a[0], loss_b[0]

    @staticmethod
    def backward(ctx, g, g_la, g_lb, ga, gb):
        from medmoe_amd import dist as D_
        al, bl, a_all, b_all, la, lb, dla, dlb = ctx.saved_tensors
        tf, off, distributed, backprop_type, dt_a, dt_b, dt_s, shp_s = ctx.meta
        B, D = al.shape
        Bg = a_all.shape[0]
        dev = al.device
        z = lambda x: 0.0 if x is None else x.float()
        Ga = ((0.5 * z(g) + z(ga)) * dla + z(g_la)).contiguous()            # total gradient reaching logits_a
        Gb = ((0.5 * z(g) + z(gb)) * dlb + z(g_lb)).contiguous()
        da = torch.empty(B, D, device=dev); db = torch.empty(B, D, device=dev)
        ops.call("sgemm", Ga, b_all, da, B, D, Bg, Bg, 1, D, 1, D, tf, 0.0)
        ops.call("sgemm", Gb, a_all, db, B, D, Bg, Bg, 1, D, 1, D, tf, 0.0)
        dscale = (Ga * la).sum() + (Gb * lb).sum()                          # logits = e^s * (...): d logits / ds = logits
        if (not distributed) or backprop_type != BackpropType.NONE:
            d_b_all = torch.empty(Bg, D, device=dev); d_a_all = torch.empty(Bg, D, device=dev)
            ops.call("sgemm", Ga, al, d_b_all, Bg, D, B, 1, Bg, D, 1, D, tf, 0.0)
            ops.call("sgemm", Gb, bl, d_a_all, Bg, D, B, 1, Bg, D, 1, D, tf, 0.0)
            if distributed and backprop_type == BackpropType.GLOBAL:
                da += D_.scatter_key_grads(d_a_all); db += D_.scatter_key_grads(d_b_all)
            else:
                da += d_a_all[off:off + B]; db += d_b_all[off:off + B]
        return da.to(dt_a), db.to(dt_b), dscale.to(dt_s).reshape(shp_s), None


def contrastive_loss_with_temperature(embeddings_a: Tensor, embeddings_b: Tensor, logit_scale: nn.Parameter,
                                      mask: Optional[Tensor] = None,
                                      backprop_type: BackpropType = BackpropType.GLOBAL,
                                      cross_entropy_kwargs: Optional[Dict[str, Any]] = None) -> ContrastiveLossOutput:
    """losses.py:527-592.  The gathered logits and their cross-entropy come from the HIP kernels (_ClipFn).  With `mask` (bool [B]: rows that
    count, :572-575) or `cross_entropy_kwargs` (e.g. label_smoothing, :577-581) - options nothing on the MedMoE path sets - the two
    cross-entropies are taken by torch on the kernels' [B, B_global] logits (gradients flow back through _ClipFn's logits outputs)."""
    loss, la, lb, loss_a, loss_b = _ClipFn.apply(embeddings_a, embeddings_b, logit_scale, backprop_type)
    if mask is not None or cross_entropy_kwargs:
        import torch.nn.functional as F
        from src.utils.distributed import get_rank
        B = la.shape[0]
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        labels = torch.arange(B, device=la.device) + (B * get_rank() if distributed else 0)          # :516-518
        if mask is not None:
            m = mask.to(la.device).bool()
            la, lb, labels = la[m], lb[m], labels[m]
        kw = dict(cross_entropy_kwargs or {})
        loss_a, loss_b = F.cross_entropy(la, labels, **kw), F.cross_entropy(lb, labels, **kw)
        loss = (loss_a + loss_b) / 2
    return ContrastiveLossOutput(loss=loss, logits_a=la, logits_b=lb, loss_a=loss_a, loss_b=loss_b)
