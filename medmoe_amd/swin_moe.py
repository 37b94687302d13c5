"""The reference's image encoder as it is written - `SWIN.forward` (reference swin.py:119-149): HF Swin-T tower with all hidden states ->
router on the mean of the last hidden state (swin.py:94-100) -> the selected modality expert over the four pyramid stages (swin.py:11-80,
105-108) -> global_feat [B, 768], local_feat [B, 3136, 768] (= [B, 768, 56, 56] token-major), router probabilities.  Forward and backward,
stand-alone: `SwinTower` + `PyramidExpert` + the engine's router kernels.  Only the SELECTED expert of a sample is
computed (the reference computes all six and gathers: same values, swin.py:105-108).

Parameter names: tower = SwinModel state_dict names prefixed `model.`, MoE = the reference's `moe.router.{0,2}.*`,
`moe.experts.{e}.proj_convs.{s}.0.*`, `moe.experts.{e}.attn_proj.{0,2}.*` (swin.py:83-92)."""
import os
from typing import Dict, Optional

import torch

from . import ops
from .flat import FlatStore
from .pyramid import GroupedPyramidExperts, PyramidExpert
from .swin import SwinTower, join_side_stream

BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class SwinMoEEncoder:
    def __init__(self, weights: Dict[str, torch.Tensor], n_expert: int = 6, device="cuda:0"):
        self.dev = torch.device(device)
        self.E = n_expert
        self.tower = SwinTower({k[len("model."):]: v for k, v in weights.items() if k.startswith("model.")}, device)
        mw = {k: v for k, v in weights.items() if k.startswith("moe.")}
        gemm = [f"moe.experts.{e}.proj_convs.{s}.0.weight" for e in range(n_expert) for s in range(4)] + \
               [f"moe.experts.{e}.attn_proj.0.weight" for e in range(n_expert)]
        # router + experts: one arena (fp32 master, gradients, bf16 copies), the experts' tensors back to back so that [E, ...] stacks are views
        self.store = FlatStore(mw, self.dev, groups=GroupedPyramidExperts.groups(n_expert), gemm=gemm)
        self.w = {k: self.store.f32(k) for k in mw}
        self.experts = [PyramidExpert({}, device, store=self.store, prefix=f"moe.experts.{e}.") for e in range(n_expert)]
        # MEDMOE_SWIN_GROUPED=0: one launch sequence per selected expert (the first form; A/B runs) instead of the grouped one
        self.grouped = GroupedPyramidExperts(self.store, n_expert, self.dev) if os.environ.get("MEDMOE_SWIN_GROUPED", "1") == "1" else None
        self.hidden = self.w["moe.router.0.weight"].shape[0]
        # weight-gradient GEMMs of the tower and the experts on a second stream underneath the dgrad chain (MEDMOE_OVERLAP_WGRAD=0: one stream)
        self.side = torch.cuda.Stream(self.dev) if (self.dev.type == "cuda" and os.environ.get("MEDMOE_OVERLAP_WGRAD", "1") == "1") else None
        self.tower.wgrad_stream = self.side
        for ex in self.experts:
            ex.wgrad_stream = self.side
        if self.grouped is not None:
            self.grouped.wgrad_stream = self.side

    def refresh(self):
        """bf16 working copies after the fp32 parameters (self.w / self.tower.w: views of the two arenas) changed: an optimizer step."""
        self.tower.refresh()
        self.store.refresh()

    def parameter_views(self) -> Dict[str, torch.Tensor]:
        """name -> fp32 view of the arenas (constructor names): what a module's nn.Parameters alias so that an optimizer step needs no copy."""
        out = {"model." + k: v for k, v in self.tower.w.items()}
        out.update(self.w)
        return out

    def new_grad_arenas(self):
        """Fresh gradient buffers for the next backward (the old ones stay alive through the .grad views still pointing into them)."""
        self.tower.store.new_grad_arena()
        self.store.new_grad_arena()

    def forward(self, images: torch.Tensor, drop_path=None, drop_path_rate: float = 0.1) -> Dict[str, torch.Tensor]:
        """drop_path: None (eval) or the per-block keep masks of `self.tower.sample_drop_path(B)` (train mode, SwinConfig.drop_path_rate)."""
        dev, w, E = self.dev, self.w, self.E
        t = self.tower.forward(images, drop_path=drop_path, drop_path_rate=drop_path_rate)
        hs, last = t["hidden_states"], t["last_hidden_state"]
        B, Dv = last.shape[0], last.shape[2]
        self.B, self.hs = B, hs
        router_in = torch.empty(B, Dv, device=dev)
        ops.call("mean_tokens", last.contiguous(), router_in, B, last.shape[1], Dv, 0, last.shape[1])             # swin.py:137
        self.router_in = router_in
        self.router_h = torch.empty(B, self.hidden, device=dev); self.probs = torch.empty(B, E, device=dev)
        self.idx = torch.empty(B, 1, device=dev, dtype=I32); gates = torch.empty(B, 1, device=dev)
        ops.call("router_fwd", router_in, w["moe.router.0.weight"], w["moe.router.0.bias"], w["moe.router.2.weight"], w["moe.router.2.bias"],
                 self.router_h, self.probs, self.idx, gates, B, Dv, self.hidden, E, 1)                          # swin.py:98-100
        P, Do = hs[0].shape[1], self.experts[0].Do
        top = self.idx[:, 0].long()
        if self.grouped is not None:
            out = self.out = self.grouped.forward([h.contiguous() for h in hs], self.idx)                      # swin.py:105-108, every expert at once
            return {"global_feat": out.float().mean(1), "local_feat": out, "router_probs": self.probs, "top_expert": top}
        out = torch.empty(B, P, Do, device=dev, dtype=BF)
        self.sel = []
        for e in range(E):
            sel = (top == e).nonzero(as_tuple=True)[0]
            self.sel.append(sel)
            if sel.numel():
                out[sel] = self.experts[e].forward([h.index_select(0, sel).contiguous() for h in hs])           # swin.py:105-108
        self.out = out
        return {"global_feat": out.float().mean(1), "local_feat": out, "router_probs": self.probs, "top_expert": top}

    def backward(self, d_global: Optional[torch.Tensor], d_local: Optional[torch.Tensor], labels: Optional[torch.Tensor] = None,
                 cls_weight: float = 0.0, d_probs: Optional[torch.Tensor] = None, zero_grad: bool = True,
                 loss_parts: Optional[torch.Tensor] = None, after_moe=None) -> Dict[str, torch.Tensor]:
        """d_global fp32 [B, 768], d_local bf16 [B, 3136, 768] (either may be None); labels + cls_weight: the reference's classifier term
        cls_weight * mean CE(router probabilities, label) as medmoe_module.py:235-237 writes it (cross-entropy applied to the PROBABILITIES);
        d_probs: an external gradient w.r.t. the router probabilities instead (fp32 [B, E]: the autograd mirror src/models/components/swin.py).
        zero_grad=False accumulates onto the arenas' contents; loss_parts: fp32 [>= 2] that receives the classifier loss (mean CE, unweighted)
        and accuracy (added to its contents) instead of a fresh tensor; after_moe: called once the MoE arena's gradients are complete, before
        the tower's backward is launched (data parallel: their all-reduce runs underneath it).  Returns fp32 gradients under the constructor's names."""
        dev, w, E, B = self.dev, self.w, self.E, self.B
        P, Do = self.out.shape[1], self.out.shape[2]
        if zero_grad:
            self.store.zero_grad()                                  # every MoE gradient (an expert no sample selected keeps zeros)
        grads = self.store.grads()
        if self.grouped is not None:
            dl = torch.zeros(B, P, Do, device=dev, dtype=BF) if d_local is None else d_local.to(BF)
            d_hs = self.grouped.backward(dl, None if d_global is None else d_global.float().contiguous())
        else:
            dy = torch.zeros(B, P, Do, device=dev, dtype=BF) if d_local is None else d_local.to(BF).clone()
            if d_global is not None:
                dy += (d_global.float() / P).to(BF)[:, None, :]
            d_hs = [torch.empty_like(h) for h in self.hs]           # top-1: every sample belongs to exactly one expert's selection
            for e in range(E):
                sel = self.sel[e]
                if sel.numel() == 0:
                    continue
                dfe, _ = self.experts[e].backward(dy.index_select(0, sel).contiguous())
                for s in range(4):
                    d_hs[s].index_copy_(0, sel, dfe[s].to(BF))
        # router: top-1 gates are 1 (swin.py:108 does not scale), so only the classifier term reaches it
        Hd, Dv = self.hidden, self.router_in.shape[1]
        d_last = None
        if (labels is not None and cls_weight != 0.0) or d_probs is not None:
            dlogits = torch.empty(B, E, device=dev); drh = torch.empty(B, Hd, device=dev); parts = torch.zeros(8, device=dev) if loss_parts is None else loss_parts
            lab = labels.to(I32).contiguous() if (labels is not None and cls_weight != 0.0) else None
            ops.call("router_bwd", self.probs, self.router_h, w["moe.router.2.weight"], self.idx, None, lab,
                     d_probs.to(F32).contiguous() if d_probs is not None else None, cls_weight / B, dlogits, drh, parts, B, Hd, E, 1)
            ones = torch.ones(B, device=dev)
            sg = lambda *a: ops.call("sgemm", *a)
            sg(dlogits, self.router_h, grads["moe.router.2.weight"], E, Hd, B, 1, E, Hd, 1, Hd, 1.0, 1.0)
            sg(ones, dlogits, grads["moe.router.2.bias"], 1, E, B, 0, 1, E, 1, E, 1.0, 1.0)
            sg(drh, self.router_in, grads["moe.router.0.weight"], Hd, Dv, B, 1, Hd, Dv, 1, Dv, 1.0, 1.0)
            sg(ones, drh, grads["moe.router.0.bias"], 1, Hd, B, 0, 1, Hd, 1, Hd, 1.0, 1.0)
            d_rin = torch.empty(B, Dv, device=dev)
            sg(drh, w["moe.router.0.weight"], d_rin, B, Dv, Hd, Hd, 1, Dv, 1, Dv, 1.0, 0.0)
            L = self.tower.res_last ** 2
            d_last = torch.empty(B, L, Dv, device=dev, dtype=BF)
            ops.call("broadcast_tokens", d_rin, d_last, B, L, Dv, 0, L, 1.0 / L)                               # mean over the 49 tokens
            self.classifier_loss = parts
        if after_moe is not None:
            join_side_stream(self.side)                             # the experts' weight gradients are complete too
            after_moe()
        tg = self.tower.backward(d_hs, d_last, zero_grad=zero_grad)
        grads.update({"model." + k: v for k, v in tg.items()})
        return grads
