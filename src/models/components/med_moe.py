"""Mirror of reference src/models/components/med_moe.py:21-108: `MedMoE(vision, text, lora)` with
`encode_image`, `encode_text`, `forward(batch) -> (img_emb_g, img_emb_l, text_emb_g, text_emb_l,
sents, router_probs)`, running on the MI355X engine (medmoe_amd).

Differences forced by the environment (documented in DESIGN.md): images arrive as a normalised
[B,3,H,W] tensor (PIL + AutoImageProcessor are a "next" row) and captions as pre-tokenised ids
(`batch['caption']` = dict(ids, attn_mask[, token_type]) or an ids tensor; the HF tokenizer hookup
is a "next" row).  All parameters of the image tower + MoE are ONE flat `nn.Parameter` (the engine's
fp32 master buffer), so any torch optimizer / Lightning sees and updates them.  The GEMMs read bf16
working copies of that buffer: `encode_image` refreshes them whenever the parameter's version counter
moved (an optimizer step, `load_state_dict`, any in-place edit) or the module was moved / re-typed, so
an external optimizer trains every weight (`Engine.train_step` fuses clip + Adam + the refresh itself).
"""
from typing import Any, Dict

import torch
from torch import nn

from medmoe_amd.config import MedMoEConfig, config_by_name
from medmoe_amd.engine import Engine


def _get(cfg: Any, key: str, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if not hasattr(cfg, "get") else cfg.get(key, default)


_ARCH = {"vit_ti16": dict(d_v=192, n_layer_v=12, n_head_v=3, ff_v=768), "vit_b16": dict(),
         "vit_l14": dict(patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096),
         "vit_l14_336": dict(img_size=336, patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096),
         # the reference's own tower (swin.py:119-149, model_name 'swin'): Swin-T + pyramid experts on src.models.components.swin.SWIN behind torch
         # autograd; the engine then only hosts the (frozen) text tower - its ViT is a one-layer ViT-Ti placeholder that is never run
         "swin_t": dict(d_v=192, n_layer_v=1, n_head_v=3, ff_v=768),
         }          # unit-test geometries: `vision.config_name: tiny` (medmoe_amd.config.config_by_name)

# reference keys (configs/model/med-moe.yaml:18-44) whose other values select code outside the hot path: rejected loudly
_FIXED = {"vision": {"use_moe": True, "projection": False, "lora": False},
          "text": {"aggregate_method": "sum", "agg_tokens": True, "norm": False, "projection": False}}


def config_from_hydra(vision: Any, text: Any) -> MedMoEConfig:
    """Extends the reference keys (configs/model/med-moe.yaml:18-44) with arch / num_experts / top_k / expert_dtype (vision)
    and n_layer (text); `config_name` picks a named geometry of medmoe_amd.config outright."""
    for side, cfg_ in (("vision", vision), ("text", text)):
        for k, want in _FIXED[side].items():
            got = _get(cfg_, k, want)
            if got != want:
                raise NotImplementedError(f"model.model.{side}.{k}={got!r}: only {want!r} is on the pretraining_medmoe hot path")
    name = _get(vision, "config_name")
    if name:
        c = config_by_name(name)
        c.freeze_text = bool(_get(text, "freeze_bert", True))
        return c
    c = MedMoEConfig(**_ARCH[_get(vision, "arch", "vit_b16")])
    c.n_expert = int(_get(vision, "num_experts", 6))          # swin.py:83 default K=6 modalities
    c.top_k = int(_get(vision, "top_k", 1))
    c.d_out = int(_get(vision, "embed_dim", c.d_out))
    c.max_len = int(_get(text, "max_length", 25))             # med-moe.yaml:40
    c.n_layer_t = int(_get(text, "n_layer", 12))
    c.last_n_layers = int(_get(text, "last_n_layers", 4))
    c.d_t = int(_get(text, "embed_dim", c.d_t))
    c.freeze_text = bool(_get(text, "freeze_bert", True))     # false: the text tower trains too (text_encoder.py:27-30) - fused step only
    dt = str(_get(vision, "expert_dtype", "bf16"))
    if dt not in ("bf16", "fp8"):
        raise NotImplementedError(f"vision.expert_dtype={dt!r}: bf16 or fp8 (e4m3 expert weights, BASELINE configs[4])")
    c.expert_fp8 = dt == "fp8"
    return c


class _ImageTowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights: torch.Tensor, images: torch.Tensor, engine: Engine):
        engine.forward_image(images)
        ctx.engine = engine
        ws = engine.ws
        return ws["img_g"].clone(), ws["img_l"].float(), ws["probs"].clone()

    @staticmethod
    def backward(ctx, d_img_g, d_img_l, d_probs):
        eng = ctx.engine
        ws = eng.ws
        ws["d_img_g"].copy_(d_img_g if d_img_g is not None else torch.zeros_like(ws["d_img_g"]))
        ws["d_img_l"].copy_(d_img_l if d_img_l is not None else torch.zeros_like(ws["d_img_l"]))
        eng.params.zero_grad()
        ws["loss_parts"].zero_()
        eng.backward(None, dprobs_ext=d_probs.float().contiguous() if d_probs is not None else None)
        return eng.params.g32.clone(), None, None


class MedMoE(nn.Module):
    def __init__(self, vision: Any, text: Any, lora: bool = False) -> None:
        super().__init__()
        if lora:
            raise NotImplementedError("lora: true is not on the pretraining_medmoe path (med-moe.yaml:27)")
        self.text, self.vision = text, vision
        self.cfg = config_from_hydra(vision, text)
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if self.device is None:
            raise RuntimeError("MedMoE (MI355X build) needs a GPU: there is no CPU fallback")
        self.engine = Engine(self.cfg, self.device)
        self.swin = None
        if _get(vision, "arch", "vit_b16") == "swin_t" and not _get(vision, "config_name"):
            from .swin import SWIN
            self.swin = SWIN(num_experts=self.cfg.n_expert, state_dict=_get(vision, "state_dict")).to(self.device)
        self.weights = nn.Parameter(self.engine.params.p32, requires_grad=self.swin is None)          # flat fp32 master, shared storage
        self._synced_version = self.weights._version                 # bf16 working copies are current for this version
        # medmoe_module.py:196 calls .image_encoder.train(), :208 reads .text_encoder.tokenizer: both towers live in this
        # one object.  Plain attributes, NOT registered submodules (a module that contains itself would recurse in
        # state_dict() / named_modules()).
        object.__setattr__(self, "image_encoder", self)
        object.__setattr__(self, "text_encoder", self)
        self.idxtoword = None                                        # set_vocabulary(): tokenizer vocabulary hookup
        self.tokenizer = None
        # checkpoints carry the reference's key layout (`image_encoder.moe.experts.{e}.proj_convs.{s}.0.weight`, `image_encoder.moe.router.*`,
        # swin.py:83-92 under med_moe.py:32; `image_encoder.model.*` for the Swin tower, `image_encoder.vit.*` for the ViT one;
        # `text_encoder.*` for the frozen text tower) instead of the flat buffer's private layout
        self._register_state_dict_hook(MedMoE._to_reference_keys)
        self._register_load_state_dict_pre_hook(self._from_reference_keys)

    @staticmethod
    def _to_reference_keys(module, state_dict, prefix, local_metadata):
        flat = state_dict.pop(prefix + "weights", None)
        if module.swin is None:
            if flat is not None:
                for k, v in module.engine.params.named_views(flat.detach()).items():
                    state_dict[prefix + "image_encoder." + k] = v
        else:
            for k in [k for k in state_dict if k.startswith(prefix + "swin.")]:
                state_dict[prefix + "image_encoder." + k[len(prefix + "swin."):]] = state_dict.pop(k)
        for k, v in module.engine.params.text.items():
            state_dict[prefix + "text_encoder." + k] = v
        return state_dict

    def _from_reference_keys(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        ie, te = prefix + "image_encoder.", prefix + "text_encoder."
        named = {k[len(ie):]: state_dict.pop(k) for k in [k for k in state_dict if k.startswith(ie)]}
        text = {"text." + k[len(te):]: state_dict.pop(k) for k in [k for k in state_dict if k.startswith(te)]}
        p = self.engine.params
        if self.swin is not None:
            for k, v in named.items():
                state_dict[prefix + "swin." + k] = v
            named = {}
        if named or text:
            if p.p32.data_ptr() != self.weights.data_ptr():
                self.refresh_working_copies()                        # re-attach the engine to the parameter's storage first
            if named:
                want = set(p.named_views())
                got = set(named)
                if strict and want - got:
                    error_msgs.append(f"MedMoE: checkpoint lacks image-encoder keys {sorted(want - got)[:3]} ... ({len(want - got)} names)")
                    return
                unexpected_keys.extend(ie + k for k in sorted(got - want))
                views = p.named_views()
                for k in want & got:
                    views[k].copy_(named[k].to(views[k].device, views[k].dtype).reshape(views[k].shape))
            if text:
                if self.engine.tstore is not None:                # trainable text tower: into its flat master buffer
                    self.engine.tstore.load_named(text)
                else:
                    p.load_named_text(text)
            p.sync_working_copies()
        # the flat parameter itself: legacy checkpoints hold it under `weights`; otherwise present the (just updated) buffer so that the
        # default loader finds its key
        if prefix + "weights" not in state_dict:
            state_dict[prefix + "weights"] = self.weights.detach().clone()

    def set_vocabulary(self, idxtoword, tokenizer=None):
        """Hook a tokenizer vocabulary up (text_encoder.py:23 idxtoword): the on-device word-piece aggregation then merges
        its '##' pieces and `sents` holds the real merged words.  `tokenizer`: optional callable
        (list[str], max_length) -> dict(ids, attn_mask[, token_type]) for raw caption strings (med_moe.py:72-85)."""
        from medmoe_amd.text import vocab_tables
        self.idxtoword = idxtoword
        self.tokenizer = tokenizer
        self.engine.vocab = vocab_tables(idxtoword, self.device)

    def refresh_working_copies(self):
        """fp32 master -> bf16 working copies (and their transposes) when the master changed since the last refresh."""
        p = self.engine.params
        w = self.weights
        if w.data_ptr() != p.p32.data_ptr():
            # the module was moved / cast (`.to()`, `.cuda()`): re-attach the engine to the parameter's new storage
            if w.dtype != torch.float32 or w.device != p.p32.device or w.numel() != p.numel:
                raise RuntimeError("MedMoE.weights must stay a flat fp32 tensor on the engine's GPU")
            p.p32 = w.data
            self._synced_version = -1
        if w._version != self._synced_version:
            p.sync_working_copies()
            self._synced_version = w._version

    def encode_image(self, images: torch.Tensor):
        """med_moe.py:67-70 -> (img_feat_g [B,D], local_feats [B,D,H,W], router_probs [B,E])."""
        if self.swin is not None:                                    # (global [B,768], local [B,768,56,56], router probabilities), swin.py:149
            self.engine._alloc(images.shape[0])                      # the text pass' buffers
            return self.swin(images)
        self.refresh_working_copies()
        img_g, img_l, probs = _ImageTowerFn.apply(self.weights, images.contiguous(), self.engine)
        B, P, D = img_l.shape
        h = int(P ** 0.5)
        return img_g, img_l.transpose(1, 2).reshape(B, D, h, h), probs

    def encode_text(self, texts: Any):
        """med_moe.py:72-100 -> (text_emb_l [B,D,T], text_emb_g [B,D], sents)."""
        if isinstance(texts, dict):
            ids, mask, tt = texts["ids"], texts["attn_mask"], texts.get("token_type")
        elif torch.is_tensor(texts):
            ids, mask, tt = texts, (texts != 0).long(), None
        elif self.tokenizer is not None:
            tok = self.tokenizer(list(texts), self.cfg.max_len)
            ids, mask, tt = tok["ids"].to(self.device), tok["attn_mask"].to(self.device), tok.get("token_type")
        else:
            raise NotImplementedError("raw caption strings need a tokenizer: set_vocabulary(idxtoword, tokenizer) or pass token ids")
        if not self.cfg.freeze_text and torch.is_grad_enabled():
            raise NotImplementedError("text.freeze_bert: false trains through the fused step (model.fused_step: true -> Engine.train_step has the "
                                      "text backward); the torch-autograd mirror keeps the text tower frozen")
        with torch.no_grad():                                        # freeze_bert: true (med-moe.yaml:35)
            self.engine.forward_text(ids, mask, tt)
        ws = self.engine.ws
        cap = self.engine.cap_lens.tolist()                          # the reference contract hands host lists around
        T = ids.shape[1]
        if self.idxtoword is not None:
            from medmoe_amd.text import merge_sents
            sents = merge_sents(ids.cpu(), self.idxtoword)
        else:
            sents = [["w"] * (c - 1) + ["[SEP]"] + ["[PAD]"] * (T - c) for c in cap]
        return ws["words32"].transpose(1, 2).clone(), ws["txt_g"].clone(), sents

    def text_soft_target(self) -> torch.Tensor:
        """fp32 [B, B] caption-to-caption scores for the Soft-GLoRIA losses (medmoe_module.py:258-281), from the text pass just run."""
        return self.engine.text_soft_target()

    def forward(self, batch: Dict[str, Any]):
        images, text = batch["image"], batch["caption"]
        # the engine's text pass needs the image pass' batch allocation: run the image tower first
        img_emb_g, img_emb_l, router_logits = self.encode_image(images)
        text_emb_l, text_emb_g, sents = self.encode_text(text)
        return img_emb_g, img_emb_l, text_emb_g, text_emb_l, sents, router_logits
