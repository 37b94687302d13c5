"""Mirror of reference src/models/medmoe_module.py:172-339 (`MedMoEPretrainingLightningModule`):
same constructor arguments and `model_step` composition.  Lightning is optional in this image: with
`lightning` importable the class is a LightningModule, otherwise a plain nn.Module with the same
methods (so the parity test runs without it)."""
from typing import Any, Dict

import torch
import torch.nn.functional as F
from torch import nn

try:                                                     # pragma: no cover - not installed in this image
    from lightning import LightningModule as _Base
except Exception:                                        # noqa: BLE001
    _Base = nn.Module


def _get(cfg: Any, key: str, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


class MedMoEPretrainingLightningModule(_Base):
    def __init__(self, model: nn.Module, loss: Any, optimizer: Any = None, scheduler: Any = None,
                 compile: bool = False, num_classes: int = 5, fused_step: bool = False):
        """`fused_step` (MI355X build, `model.fused_step` in the config tree): training steps run `Engine.train_step` - the hand-scheduled
        forward / losses / backward with the embedding all-gather, the reduce-scatter of the gathered-key gradients, the per-layer
        gradient all-reduce overlapped with backward and the fused clip + Adam - instead of torch autograd + a torch optimizer.
        Same losses, same update rule (tests/test_fused_module_gpu.py); needs the ViT image tower, the two GLoRIA losses (or their
        Soft variants) and torch.optim.Adam in the config, and refuses anything else at construction."""
        super().__init__()
        self.model = model
        self.loss_cfg = loss
        self.local_loss = _get(loss, "local_loss")
        self.global_loss = _get(loss, "global_loss")
        self.local_loss_weight = _get(loss, "local_loss_weight", 0.4)          # medmoe_module.py:186-188
        self.global_loss_weight = _get(loss, "global_loss_weight", 0.4)
        self.classifier_loss_weight = _get(loss, "classifier_loss_weight", 0.2)
        self._optimizer, self._scheduler = optimizer, scheduler
        self.soft_label = bool(_get(loss, "soft_label", False))                 # :207-210: the reference loads `tool_bert` here
        self.fused_step = bool(fused_step)
        self._fused_acc, self._fused_clip, self._fused_opt = 1, None, None
        if self.fused_step:
            self.automatic_optimization = False                                  # Lightning: manual optimisation (the engine steps itself)
            self._configure_engine()

    def forward(self, batch):
        return self.model(batch)

    def _calc_global_loss(self, img_emb_g, text_emb_g, idx=None, probs=None):           # :212-218
        return self.global_loss(img_emb_g, text_emb_g, temp3=_get(self.loss_cfg, "temp3", 4.0), idx=idx, probs=probs)

    def _calc_local_loss(self, img_emb_l, text_emb_l, sents, idx=None, probs=None):     # :220-233
        cap_lens = [len([w for w in sent if not w.startswith("[")]) + 1 for sent in sents]
        out = self.local_loss(img_emb_l, text_emb_l, cap_lens, temp1=_get(self.loss_cfg, "temp1", 4.0),
                              temp2=_get(self.loss_cfg, "temp2", 5.0), temp3=_get(self.loss_cfg, "temp3", 10.0),
                              idx=idx, probs=probs)
        return out.loss0 + out.loss1

    def _calc_classifier_loss(self, router_logits, labels):                             # :235-237
        return F.cross_entropy(router_logits, labels)

    def _calc_classifier_acc(self, router_logits, labels):                              # :239-241
        return (torch.argmax(router_logits, dim=1) == labels).float().mean()

    def get_text_soft_target(self, raw_txt, topK, threshold):                            # :258-281
        """(caption-to-caption scores [B, B], thresholds).  The reference runs a second frozen pretrained BertModel over `raw_txt`; with
        the text tower frozen that is the tower's own BERT, so the scores come from the text pass `forward` just ran
        (`Engine.text_soft_target`: [CLS] of the last layer, L2-normalised, pairwise products); `raw_txt` / `topK` are unused, as in
        the reference (its top-k filter is commented out, :278-280)."""
        with torch.no_grad():
            return self.model.text_soft_target(), threshold

    def model_step(self, batch: Dict[str, Any]):                                         # :284-316
        img_emb_g, img_emb_l, text_emb_g, text_emb_l, sents, router_logits = self.forward(batch)
        if self.soft_label:                                                              # :291-296
            idx, filt = self.get_text_soft_target(batch["caption"], _get(self.loss_cfg, "topk", 5),
                                                  (_get(self.loss_cfg, "threshold0", 0.98), _get(self.loss_cfg, "threshold1", 0.97)))
            l_loss = self._calc_local_loss(img_emb_l, text_emb_l, sents, idx, filt)
            g_loss = self._calc_global_loss(img_emb_g, text_emb_g, idx, filt)
        else:
            l_loss = self._calc_local_loss(img_emb_l, text_emb_l, sents)
            g_loss = self._calc_global_loss(img_emb_g, text_emb_g)
        classifier_loss = self._calc_classifier_loss(router_logits, batch["label"])
        classifier_acc = self._calc_classifier_acc(router_logits, batch["label"])
        loss = self.local_loss_weight * l_loss + self.global_loss_weight * g_loss + self.classifier_loss_weight * classifier_loss
        return {"loss": loss, "l_loss": l_loss, "g_loss": g_loss, "classifier_loss": classifier_loss,
                "classifier_acc": classifier_acc}

    def training_step(self, batch, batch_idx: int = 0):                                  # :318-339
        if self.fused_step:
            acc = self._fused_acc
            return self.fused_training_step(batch, optimizer_step=(batch_idx + 1) % acc == 0, zero_grad=batch_idx % acc == 0,
                                            loss_scale=1.0 / acc)["loss"]
        return self.model_step(batch)["loss"]

    # ---- fused mode --------------------------------------------------------------------------------------------------
    def _configure_engine(self):
        """Carry the module's loss / optimiser configuration into the engine's (medmoe_amd.config.MedMoEConfig); refuse what the fused
        step does not compute."""
        import functools

        import src.losses as L
        eng = getattr(self.model, "engine", None)
        if eng is None:
            raise NotImplementedError("fused_step needs the HIP engine behind self.model (src.models.components.med_moe.MedMoE)")
        # arch = swin_t (the reference's own encoder): medmoe_amd.swin_engine.SwinEngine, built on the first step (the encoder's arenas exist
        # once the module sits on the GPU); the ViT towers: Engine.train_step
        self._swin_engine = None
        soft = (type(self.global_loss) is L.SoftGLORIAGlobalContrastiveLoss, type(self.local_loss) is L.SoftGLORIALocalContrastiveLoss)
        hard = (type(self.global_loss) is L.GLORIAGlobalContrastiveLoss, type(self.local_loss) is L.GLORIALocalContrastiveLoss)
        if not (all(hard) or all(soft)) or all(soft) != self.soft_label:
            raise NotImplementedError("fused_step computes GLORIA{Global,Local}ContrastiveLoss (or both Soft variants with loss.soft_label: true); "
                                      f"got {type(self.global_loss).__name__} / {type(self.local_loss).__name__}, soft_label={self.soft_label}")
        if _get(self.loss_cfg, "agg", "sum") != "sum":
            raise NotImplementedError("fused_step: only loss.agg = 'sum' (the reference default)")
        opt = self._optimizer
        if not isinstance(opt, functools.partial) or opt.func is not torch.optim.Adam or opt.args \
                or set(opt.keywords) - {"lr", "weight_decay", "betas", "eps"} \
                or tuple(opt.keywords.get("betas", (0.9, 0.999))) != (0.9, 0.999) or float(opt.keywords.get("eps", 1e-8)) != 1e-8:
            raise NotImplementedError("fused_step fuses torch.optim.Adam(lr, weight_decay) with betas (0.9, 0.999), eps 1e-8 (the experiment's "
                                      "optimizer, med-moe_pretraining.yaml:7-11)")
        c = eng.cfg
        c.temp1, c.temp2 = float(_get(self.loss_cfg, "temp1", 4.0)), float(_get(self.loss_cfg, "temp2", 5.0))
        c.temp3 = float(_get(self.loss_cfg, "temp3", 10.0))
        c.w_local, c.w_global, c.w_cls = float(self.local_loss_weight), float(self.global_loss_weight), float(self.classifier_loss_weight)
        c.soft_label = self.soft_label
        c.local_loss_global = bool(_get(self.loss_cfg, "local_loss_global", False))
        c.threshold0, c.threshold1 = float(_get(self.loss_cfg, "threshold0", 0.98)), float(_get(self.loss_cfg, "threshold1", 0.97))
        c.lr, c.weight_decay = float(opt.keywords.get("lr", 1e-3)), float(opt.keywords.get("weight_decay", 0.0))

    def configure_fused(self, accumulate_grad_batches: int = 1, gradient_clip_val=None):
        """The trainer's two keys the fused step has to honour itself (trainer.accumulate_grad_batches, trainer.gradient_clip_val;
        pretraining_medmoe.yaml:23-24).  gradient_clip_val None / 0 = no clipping."""
        self._fused_acc = max(1, int(accumulate_grad_batches))
        self._fused_clip = float(gradient_clip_val) if gradient_clip_val else None
        self.model.engine.cfg.clip = self._fused_clip if self._fused_clip is not None else 0.0

    def fused_training_step(self, batch: Dict[str, Any], optimizer_step: bool = True, zero_grad: bool = True, loss_scale: float = 1.0):
        """One micro-batch through Engine.train_step; returns the reference's loss names (device scalars)."""
        if not self.fused_step:
            raise RuntimeError("fused_training_step: construct the module with fused_step=True (model.fused_step=true)")
        m = self.model
        if getattr(m, "swin", None) is not None:
            from medmoe_amd.swin_engine import SwinEngine
            enc = m.swin._encoder()                                  # (re)builds the arenas / refreshes the bf16 copies after an external edit
            if self._swin_engine is None or self._swin_engine.enc is not enc:
                self._swin_engine = SwinEngine(m.engine, enc, drop_path_rate=m.swin.drop_path_rate)
            eng = self._swin_engine
            eng.training = self.training
        else:
            m.refresh_working_copies()                               # a load_state_dict / external edit of the flat parameter since the last step
            eng = m.engine
        if self._fused_opt is not None:                              # the scheduler acts on this optimizer's lr; the engine applies it
            eng.cfg.lr = float(self._fused_opt.param_groups[0]["lr"])
        cap = batch["caption"]
        if isinstance(cap, dict):
            ids, mask, tt = cap["ids"], cap["attn_mask"], cap.get("token_type")
        elif torch.is_tensor(cap):
            ids, mask, tt = cap, (cap != 0).long(), None
        else:
            raise NotImplementedError("fused_step takes pre-tokenised captions (dict(ids, attn_mask) or an ids tensor)")
        eb = {"image": batch["image"].contiguous(), "ids": ids, "attn_mask": mask, "label": batch["label"]}
        if tt is not None:
            eb["token_type"] = tt
        out = eng.train_step(eb, optimizer=optimizer_step, zero_grad=zero_grad, loss_scale=loss_scale)
        return {"loss": out["loss"], "l_loss": out["l_loss"], "g_loss": out["g_loss"], "classifier_loss": out["classifier_loss"],
                "classifier_acc": out["classifier_acc"]}

    # ---- fused mode: the optimiser state lives in the engine's flat stores, not in a torch optimizer -----------------------------------
    def _fused_stores(self) -> Dict[str, Any]:
        """name -> flat store holding Adam moments (`m`, `v`, `step_count`) of the fused step."""
        m = self.model
        if getattr(m, "swin", None) is not None:
            enc = m.swin._encoder()
            return {"swin_tower": enc.tower.store, "swin_moe": enc.store}
        out = {"image": m.engine.params}
        if m.engine.tstore is not None:
            out["text"] = m.engine.tstore
        return out

    def on_save_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        """Lightning hook (the stand-in trainer calls it too): Adam's moments and step counts of the fused step travel with the checkpoint
        (`fused_adam`: per store `exp_avg` / `exp_avg_sq` in the store's flat layout + `step`), what `optimizer_states` holds for the
        torch-optimizer path, so that `fit(ckpt_path=...)` resumes the SAME optimisation."""
        if not self.fused_step:
            return
        state = {}
        for name, st in self._fused_stores().items():
            if getattr(st, "m", None) is None:
                continue                                             # no optimiser step taken yet
            state[name] = {"step": int(st.step_count), "numel": int(st.m.numel()), "exp_avg": st.m.detach().cpu().clone(),
                           "exp_avg_sq": st.v.detach().cpu().clone()}
        checkpoint["fused_adam"] = state

    def on_load_checkpoint(self, checkpoint: Dict[str, Any]) -> None:
        state = checkpoint.get("fused_adam") if self.fused_step else None
        if not state:
            return
        stores = self._fused_stores()
        for name, rec in state.items():
            if name not in stores:
                raise KeyError(f"checkpoint holds fused Adam state for {name!r}; this module has {sorted(stores)}")
            st = stores[name]
            if getattr(st, "m", None) is None:
                st.m, st.v = torch.zeros_like(st.p32), torch.zeros_like(st.p32)
            if int(rec["numel"]) != st.m.numel():
                raise ValueError(f"fused Adam state {name!r}: {rec['numel']} elements in the checkpoint, {st.m.numel()} in this model")
            st.m.copy_(rec["exp_avg"]); st.v.copy_(rec["exp_avg_sq"])
            st.step_count = int(rec["step"])

    def configure_optimizers(self):                                                      # :148-169
        opt = self._optimizer(params=self.parameters())
        if self.fused_step:
            self._fused_opt = opt                                    # never stepped: it carries lr for the scheduler / checkpoints
        if self._scheduler is None:
            return {"optimizer": opt}
        return {"optimizer": opt, "lr_scheduler": {"scheduler": self._scheduler(optimizer=opt), "monitor": "val/loss",
                                                   "interval": "epoch", "frequency": 1}}
