"""Model / loss configuration of the MedMoE hot path (mirrors the reference's Hydra keys:
configs/model/med-moe.yaml, configs/model/med-moe_pretraining.yaml; ViT/MoE geometry per
BASELINE.json configs)."""
from dataclasses import dataclass
from typing import List


@dataclass
class MedMoEConfig:
    # image tower: pre-norm ViT blocks (reference transformer.py:98-114), eps 1e-6, final LN
    img_size: int = 224
    patch: int = 16
    d_v: int = 768
    n_layer_v: int = 12
    n_head_v: int = 12
    ff_v: int = 3072
    eps_v: float = 1e-6
    # text tower: post-norm BERT-geometry blocks (transformer.py:116-130), eps 1e-12, frozen
    vocab: int = 28996
    max_len: int = 77
    d_t: int = 768
    n_layer_t: int = 12
    n_head_t: int = 12
    ff_t: int = 3072
    eps_t: float = 1e-12
    last_n_layers: int = 4
    freeze_text: bool = True      # configs/model/med-moe.yaml:35 freeze_bert: true (the experiment); False = the text tower trains too
                                  # (text_encoder.py:27-30): padded text pass with saved activations, text backward, word gradients of the local loss
    # MoE (swin.py:82-92)
    n_expert: int = 4
    top_k: int = 1
    router_hidden: int = 128
    d_out: int = 768
    expert_fp8: bool = False      # BASELINE configs[4]: e4m3 expert weights (per-output-channel scales) on the fp8 MFMA
    # losses (med-moe_pretraining.yaml:20-41)
    temp1: float = 4.0
    temp2: float = 5.0
    temp3: float = 10.0
    w_local: float = 0.5
    w_global: float = 0.5
    w_cls: float = 2.0
    # Soft-GLoRIA (med-moe_pretraining.yaml:25-28; losses.py:814-883, 1111-1214): positives = captions whose frozen-BERT [CLS] cosine with the
    # row's caption exceeds threshold0, negatives = those at or below threshold1 (medmoe_module.py:258-281, 290-295)
    # data-parallel variant of the local loss (NOT the reference's behaviour, which keeps the local loss
    # rank-local): every rank scores its images against the captions of ALL ranks (words all-gathered: 15 MB per rank at batch 128), the
    # [B_g, B_g] similarity matrix is assembled by one more all-gather and both cross-entropies run over the global batch - the
    # W-rank step then equals the one-process step on the concatenated batch for the local loss too.  196 / 64 regions only.
    local_loss_global: bool = False
    soft_label: bool = False
    threshold0: float = 0.98
    threshold1: float = 0.97
    # optimiser (med-moe_pretraining.yaml:7-11, pretraining_medmoe.yaml:23)
    lr: float = 5e-5
    weight_decay: float = 0.0
    clip: float = 0.25

    @property
    def n_patch(self) -> int:
        return (self.img_size // self.patch) ** 2

    @property
    def n_tok_v(self) -> int:
        return self.n_patch + 1

    @property
    def patch_dim(self) -> int:
        return 3 * self.patch * self.patch

    @property
    def patch_dim_pad(self) -> int:
        """im2col row pitch: the patch row padded to the GEMM k-step (patch 14: 588 -> 640); pad columns are zero."""
        return (self.patch_dim + 63) // 64 * 64

    def stage_layers(self) -> List[int]:
        L = self.n_layer_v
        return [max(1, (L * (s + 1)) // 4) for s in range(4)]

    def validate(self):
        if self.d_v % self.n_head_v or self.d_v // self.n_head_v != 64:
            raise ValueError("image tower head_dim must be 64")
        if self.d_t % self.n_head_t or self.d_t // self.n_head_t != 64:
            raise ValueError("text tower head_dim must be 64")
        if self.d_t != self.d_out:
            raise ValueError("text width must equal the expert output width (no projection in the reference path)")
        for d in (self.d_v, self.ff_v, self.d_t, self.ff_t, self.d_out):
            if d % 64:
                raise ValueError(f"GEMM contraction dims must be multiples of 64, got {d}")
        if (self.d_out // 2) % 64:
            raise ValueError("expert attention hidden (d_out/2) must be a multiple of 64")
        if self.patch_dim % 4:
            raise ValueError("3*patch^2 must be a multiple of 4")
        if int(self.n_patch ** 0.5) ** 2 != self.n_patch:
            raise ValueError("patch grid must be square")


def config_by_name(name: str) -> MedMoEConfig:
    """BASELINE.json configs[0..2], the bf16 geometry of configs[4] (ViT-L/14, 16 experts) + unit-test scales."""
    if name == "cfg0":
        return MedMoEConfig(d_v=192, n_layer_v=12, n_head_v=3, ff_v=768, max_len=25, n_layer_t=2,
                            n_expert=2, top_k=1)
    if name == "cfg1":
        return MedMoEConfig(n_expert=4, top_k=1)
    if name == "cfg2":
        return MedMoEConfig(n_expert=8, top_k=2)
    if name == "cfg4":
        return MedMoEConfig(patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096, n_expert=16, top_k=2, expert_fp8=True)
    if name == "cfg4_bf16":  # the same geometry with bf16 expert weights
        return MedMoEConfig(patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096, n_expert=16, top_k=2)
    if name == "tinyL8":     # tinyL with fp8 expert weights (configs[4]'s expert arithmetic at unit-test width)
        c = config_by_name("tinyL")
        c.expert_fp8 = True
        return c
    if name == "tinyL":       # cfg4's geometry (patch 14 -> 256 regions, 257 tokens) at unit-test width
        return MedMoEConfig(img_size=224, patch=14, d_v=64, n_layer_v=4, n_head_v=1, ff_v=128, vocab=97,
                            max_len=40, d_t=128, n_layer_t=2, n_head_t=2, ff_t=256, n_expert=3, top_k=2,
                            d_out=128)
    if name == "tiny":
        return MedMoEConfig(img_size=64, patch=8, d_v=64, n_layer_v=4, n_head_v=1, ff_v=128, vocab=97,
                            max_len=16, d_t=128, n_layer_t=4, n_head_t=2, ff_t=256, n_expert=3, top_k=1,
                            d_out=128)
    if name == "tiny2":
        c = config_by_name("tiny")
        c.top_k = 2
        return c
    if name == "cfg3":       # BASELINE.json configs[3]: ViT-L/14 at 336 px (577 tokens, 576 regions), 8 experts top-2
        return MedMoEConfig(img_size=336, patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096, n_expert=8, top_k=2)
    if name == "tinyL336":   # cfg3's token geometry (336 px / patch 14 -> 576 regions, 577 tokens) at unit-test width
        return MedMoEConfig(img_size=336, patch=14, d_v=64, n_layer_v=4, n_head_v=1, ff_v=128, vocab=97,
                     max_len=40, d_t=128, n_layer_t=2, n_head_t=2, ff_t=256, n_expert=3, top_k=2, d_out=128)
    if name == "tiny5":       # top-1 over five experts: the routing tests need >= 3 active experts AND an empty one
        c = config_by_name("tiny")
        c.n_expert = 5
        return c
    raise KeyError(name)
