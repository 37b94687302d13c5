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
                 compile: bool = False, num_classes: int = 5):
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
        return self.model_step(batch)["loss"]

    def configure_optimizers(self):                                                      # :148-169
        opt = self._optimizer(params=self.parameters())
        if self._scheduler is None:
            return {"optimizer": opt}
        return {"optimizer": opt, "lr_scheduler": {"scheduler": self._scheduler(optimizer=opt), "monitor": "val/loss",
                                                   "interval": "epoch", "frequency": 1}}
