"""Fused training step for the REFERENCE'S OWN model (configs/experiment/pretraining_medmoe.yaml + configs/model/med-moe.yaml: HF Swin-T tower,
six pyramid experts over its four stages, 56 x 56 = 3136 local regions, frozen BERT-geometry text tower): medmoe_module.py:284-316 `model_step`
+ backward + clip_grad_norm_ + Adam as ONE hand-scheduled launch sequence, no torch autograd and no torch optimizer -

    text tower (frozen; second stream)   Engine.forward_text                      -> words, txt_g, caption lengths
    image encoder                        SwinMoEEncoder.forward (swin.py:119-149) -> global_feat, local_feat [B, 3136, 768], router probabilities
    GLoRIA global loss, fwd + bwd        Engine.global_loss (losses.py:766-794; all-gather + local-rows InfoNCE under data parallelism)
    GLoRIA local loss, fwd + bwd         GenericLocalLoss (losses.py:961-1026 at 3136 regions)
    classifier term + encoder backward   SwinMoEEncoder.backward into the two flat gradient arenas (medmoe_amd.flat.FlatStore)
    clip + Adam                          one norm over both arenas, the fused Adam kernel on each, bf16 copies refreshed in place

The same losses and the same update rule as the torch-autograd mirror (src/models/components/swin.SWIN + src.losses + torch.optim.Adam:
tests/test_swin_engine_gpu.py steps both from one initial state).  The `Engine` passed in hosts the text tower, the global loss and the loss
configuration (cfg.temp*, w_*, lr, weight_decay, clip); its own ViT is the unused placeholder med_moe.py builds for arch = swin_t.

Data parallel: the embeddings are all-gathered for the global loss inside Engine.global_loss, the local loss stays rank-local (the reference's
own behaviour), the two gradient arenas are averaged over ranks - the MoE arena while the tower's backward still runs."""
from typing import Dict, Optional

import torch

from . import ops
from .engine import Engine
from .local_generic import GenericLocalLoss
from .swin_moe import SwinMoEEncoder

BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class SwinEngine:
    def __init__(self, engine: Engine, encoder: SwinMoEEncoder, drop_path_rate: float = 0.1):
        if engine.train_text:
            raise NotImplementedError("SwinEngine: the text tower stays frozen (med-moe.yaml:35); freeze_bert: false trains with the ViT towers")
        if engine.cfg.soft_label and engine.dist:
            raise NotImplementedError("soft_label with more than one rank (losses.py:826-883 has no gather)")
        self.eng, self.enc, self.cfg = engine, encoder, engine.cfg
        self.device = encoder.dev
        self.drop_path_rate = float(drop_path_rate)
        self.training = True
        self._loc: Optional[GenericLocalLoss] = None
        self._gsim = None
        self._normsq = torch.zeros(1, device=self.device, dtype=F32)
        self._keep = None

    def _local(self, B: int, HW: int, T: int, D: int) -> GenericLocalLoss:
        if self._loc is None or (self._loc.B, self._loc.HW, self._loc.T, self._loc.D) != (B, HW, T, D):
            self._loc = GenericLocalLoss(B, HW, T, D, self.device)
            self._gsim = torch.empty(B, B, device=self.device, dtype=F32)
        return self._loc

    def _drop_path_masks(self, B: int):
        """Train-mode stochastic depth (SwinDropPath.forward): per block a [B] keep mask, None where the probability is 0; sampled on the
        device in one launch (the module path samples on the host, one copy per block)."""
        if not self.training or self.drop_path_rate == 0.0:
            return None
        if self._keep is None or self._keep[0] != self.drop_path_rate:
            rates = self.enc.tower.drop_path_rates(self.drop_path_rate)
            self._keep = (self.drop_path_rate, rates, torch.tensor([1.0 - r for r in rates], device=self.device).unsqueeze(1))
        _, rates, keep = self._keep
        masks = torch.floor(torch.rand(len(rates), B, device=self.device) + keep)
        return [None if r == 0.0 else masks[i] for i, r in enumerate(rates)]

    def forward(self, batch: Dict[str, torch.Tensor], drop_path=None):
        """Both towers' forward.  Returns the encoder's output dict; the text outputs are in the engine's workspace."""
        eng, enc = self.eng, self.enc
        images = batch["image"]
        B = images.shape[0]
        eng._alloc(B)
        eng.prefetch_cap_lens(batch["ids"])
        main = torch.cuda.current_stream()
        side = eng._side_stream()
        ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):                               # the frozen text tower is independent of the image encoder
            eng.forward_text(batch["ids"], batch["attn_mask"], batch.get("token_type"))
            done = torch.cuda.Event(); done.record(side)
        out = enc.forward(images.contiguous(), drop_path=drop_path, drop_path_rate=self.drop_path_rate)
        main.wait_event(done)
        return out

    def train_step(self, batch: Dict[str, torch.Tensor], optimizer: bool = True, zero_grad: bool = True, loss_scale: float = 1.0):
        """One micro-batch; the arguments and the returned names are Engine.train_step's (gradient accumulation: optimizer=False for all but
        the last micro-batch, zero_grad=False for all but the first, loss_scale = 1 / number of micro-batches)."""
        eng, enc, c = self.eng, self.enc, self.cfg
        B = batch["image"].shape[0]
        out = self.forward(batch, self._drop_path_masks(B))
        ws = eng.ws
        local = out["local_feat"]                                   # bf16 [B, 3136, 768]
        HW, D = local.shape[1], local.shape[2]
        if D != c.d_out or ws["img_g"].shape[1] != D:
            raise ValueError(f"SwinEngine: the encoder's embedding width {D} != the engine's d_out {c.d_out}")
        ws["img_g"].copy_(out["global_feat"])
        eng.global_loss(loss_scale)                                 # -> loss_parts[2], ws["d_img_g"]
        lp = ws["loss_parts"]
        loc = self._local(B, HW, c.max_len, D)
        sim = loc.forward(local.view(B * HW, D), ws["words"], eng.cap_lens, c.temp1, c.temp2)
        wl = c.w_local * loss_scale / B
        eng._head(sim, self._gsim, B, 1, wl, 0, lp[3:])
        eng._head(sim, self._gsim, 1, B, wl, 1, lp[3:])
        d_local = loc.backward(self._gsim)
        if eng.dist and optimizer:
            import torch.distributed as dist
            pending = []

            def moe_ready():                                        # the MoE arena is complete: its all-reduce runs under the tower's backward
                if dist.get_backend() != "nccl":
                    torch.cuda.synchronize()                        # gloo reads the buffer from the host right away (tests)
                pending.append(dist.all_reduce(enc.store.g32, async_op=True))
            enc.backward(ws["d_img_g"], d_local.view(B, HW, D), labels=batch["label"], cls_weight=c.w_cls * loss_scale, zero_grad=zero_grad,
                         loss_parts=lp, after_moe=moe_ready)
            if dist.get_backend() != "nccl":
                torch.cuda.synchronize()
            dist.all_reduce(enc.tower.store.g32)
            pending[0].wait()
            enc.store.g32.div_(eng.world); enc.tower.store.g32.div_(eng.world)
        else:
            enc.backward(ws["d_img_g"], d_local.view(B, HW, D), labels=batch["label"], cls_weight=c.w_cls * loss_scale, zero_grad=zero_grad,
                         loss_parts=lp)
        if optimizer:
            self.optimizer_step()
        cls = lp[0] * loss_scale
        return {"loss": c.w_cls * cls + lp[2] + lp[3], "classifier_loss": cls, "classifier_acc": lp[1],
                "g_loss": lp[2] / c.w_global, "l_loss": lp[3] / c.w_local}

    def optimizer_step(self, lr: Optional[float] = None):
        """clip_grad_norm_(cfg.clip) over BOTH arenas + torch.optim.Adam(lr, weight_decay), fused; the bf16 working copies follow."""
        c, st_t, st_m = self.cfg, self.enc.tower.store, self.enc.store
        lr = c.lr if lr is None else lr
        torch.add(st_t.sumsq(), st_m.sumsq(), out=self._normsq)
        st_t.adam_step(self._normsq, lr, c.weight_decay, c.clip)
        st_m.adam_step(self._normsq, lr, c.weight_decay, c.clip)
        self.enc.tower.refresh(cast=False)                          # patch-embedding pad form, bias tables

    def eval_step(self, batch: Dict[str, torch.Tensor]):
        """Forward + losses without gradients reaching the parameters (validation): the same launch sequence minus the encoder backward."""
        was = self.training
        self.training = False
        try:
            eng, c = self.eng, self.cfg
            B = batch["image"].shape[0]
            out = self.forward(batch, None)
            ws = eng.ws
            local = out["local_feat"]
            HW, D = local.shape[1], local.shape[2]
            ws["img_g"].copy_(out["global_feat"])
            eng.global_loss(1.0)
            lp = ws["loss_parts"]
            loc = self._local(B, HW, c.max_len, D)
            sim = loc.forward(local.view(B * HW, D), ws["words"], eng.cap_lens, c.temp1, c.temp2)
            wl = c.w_local / B
            eng._head(sim, self._gsim, B, 1, wl, 0, lp[3:])
            eng._head(sim, self._gsim, 1, B, wl, 1, lp[3:])
            probs = out["router_probs"]
            lab = batch["label"].long()
            cls = torch.nn.functional.cross_entropy(probs, lab)    # medmoe_module.py:235-237: CE applied to the probabilities
            acc = (probs.argmax(1) == lab).float().mean()
            return {"loss": c.w_cls * cls + lp[2] + lp[3], "classifier_loss": cls, "classifier_acc": acc,
                    "g_loss": lp[2] / c.w_global, "l_loss": lp[3] / c.w_local}
        finally:
            self.training = was
