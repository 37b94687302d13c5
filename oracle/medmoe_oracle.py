"""CPU oracle for the MedMoE contrastive hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch fp32 restatement (torch CPU + numpy) of the reference
algorithm for the path named in BASELINE.json.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (``medmoe_amd``) never does and fails loudly without its HIP library.

Parity status: PINNED.  ``oracle/gen_golden.py`` runs the reference's own importable
modules (SURVEY.md section 8c recipe) in the build container and commits their outputs
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function below
against them (fp32 rtol 1e-5 / atol 1e-6, router indices exact).

Every function cites the reference file:line (relative to /root/reference) it restates.
Pieces the reference does not define (ViT patch/CLS/pos embedding, BERT-style text
embedding, top-k>1 gating, stage-feature taps of a ViT) are marked BUILD-DEFINED.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    # image tower (BUILD-DEFINED geometry; block semantics = transformer.py pre-norm)
    img_size: int = 224
    patch: int = 16
    d_v: int = 768
    n_layer_v: int = 12
    n_head_v: int = 12
    ff_v: int = 3072
    eps_v: float = 1e-6
    # text tower (BERT geometry; block semantics = transformer.py post-norm)
    vocab: int = 28996
    max_len: int = 77
    d_t: int = 768
    n_layer_t: int = 12
    n_head_t: int = 12
    ff_t: int = 3072
    eps_t: float = 1e-12
    last_n_layers: int = 4          # configs/model/med-moe.yaml:36
    # MoE (swin.py:82-92)
    n_expert: int = 4
    top_k: int = 1
    router_hidden: int = 128        # swin.py:89
    d_out: int = 768                # swin.py:83 output_dim
    freeze_text: bool = True        # configs/model/med-moe.yaml:35 freeze_bert: true; False: text_encoder.py:27-30 leaves the tower trainable
    expert_fp8: bool = False        # BUILD-DEFINED (BASELINE configs[4]): e4m3 expert weights + per-row e4m3 activations on the fp8 MFMA
    # losses (configs/model/med-moe_pretraining.yaml:20-41)
    temp1: float = 4.0
    temp2: float = 5.0
    temp3: float = 10.0
    w_local: float = 0.5
    w_global: float = 0.5
    w_cls: float = 2.0
    soft_label: bool = False      # Soft-GLoRIA (med-moe_pretraining.yaml:25-28)
    threshold0: float = 0.98
    threshold1: float = 0.97

    @property
    def n_patch(self) -> int:
        return (self.img_size // self.patch) ** 2

    @property
    def n_tok_v(self) -> int:
        return self.n_patch + 1

    def stage_layers(self) -> List[int]:
        """BUILD-DEFINED: hidden states after layers L/4, L/2, 3L/4, L
        stand in for Swin's `hidden_states[0:4]` (swin.py:139)."""
        L = self.n_layer_v
        return [max(1, (L * (s + 1)) // 4) for s in range(4)]


def config_by_name(name: str) -> OracleConfig:
    if name == "cfg0":   # BASELINE.json configs[0]
        return OracleConfig(d_v=192, n_layer_v=12, n_head_v=3, ff_v=768, max_len=25,
                            n_layer_t=2, n_expert=2, top_k=1)
    if name == "cfg1":
        return OracleConfig(n_expert=4, top_k=1)
    if name == "cfg2":
        return OracleConfig(n_expert=8, top_k=2)
    if name == "cfg4":   # BASELINE.json configs[4]: fp8 expert weights modelled by fake quantisation (fake_quant_rows below)
        return OracleConfig(patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096, n_expert=16, top_k=2, expert_fp8=True)
    if name == "cfg4_bf16":  # the same geometry with bf16 expert weights
        return OracleConfig(patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096, n_expert=16, top_k=2)
    if name == "tinyL8":     # tinyL with fp8 expert weights (configs[4]'s expert arithmetic at unit-test width)
        c = config_by_name("tinyL")
        c.expert_fp8 = True
        return c
    if name == "tinyL":  # cfg4's token geometry (patch 14: 256 regions) at unit-test width (matches medmoe_amd.config)
        return OracleConfig(img_size=224, patch=14, d_v=64, n_layer_v=4, n_head_v=1, ff_v=128,
                            vocab=97, max_len=40, d_t=128, n_layer_t=2, n_head_t=2, ff_t=256,
                            n_expert=3, top_k=2, d_out=128)
    if name == "tiny":   # unit-test scale (must match medmoe_amd.config "tiny")
        return OracleConfig(img_size=64, patch=8, d_v=64, n_layer_v=4, n_head_v=1, ff_v=128,
                            vocab=97, max_len=16, d_t=128, n_layer_t=4, n_head_t=2, ff_t=256,
                            n_expert=3, top_k=1, d_out=128)
    if name == "tiny2":
        c = config_by_name("tiny")
        c.top_k = 2
        return c
    if name == "cfg3":       # BASELINE.json configs[3]: ViT-L/14 at 336 px (577 tokens, 576 regions), 8 experts top-2
        return OracleConfig(img_size=336, patch=14, d_v=1024, n_layer_v=24, n_head_v=16, ff_v=4096, n_expert=8, top_k=2)
    if name == "tinyL336":   # cfg3's token geometry (336 px / patch 14 -> 576 regions, 577 tokens) at unit-test width
        return OracleConfig(img_size=336, patch=14, d_v=64, n_layer_v=4, n_head_v=1, ff_v=128, vocab=97,
                     max_len=40, d_t=128, n_layer_t=2, n_head_t=2, ff_t=256, n_expert=3, top_k=2, d_out=128)
    if name == "tiny5":       # top-1 over five experts: the routing tests need >= 3 active experts AND an empty one
        c = config_by_name("tiny")
        c.n_expert = 5
        return c
    raise KeyError(name)


# --------------------------------------------------------------------------------------
# parameter construction (init rule: multimodal_transformer.py:298-312 — normal(0, 0.02),
# biases 0, LayerNorm weight 1 / bias 0)
# --------------------------------------------------------------------------------------
def init_params(cfg: OracleConfig, seed: int = 0, std: float = 0.02) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f):
        p[name + ".weight"] = torch.randn(out_f, in_f, generator=g) * std
        p[name + ".bias"] = torch.zeros(out_f)

    def ln(name, d):
        p[name + ".weight"] = torch.ones(d)
        p[name + ".bias"] = torch.zeros(d)

    def enc(prefix, n_layer, d, ff):
        for i in range(n_layer):
            b = f"{prefix}.layer.{i}"
            ln(b + ".attention_layernorm", d)
            lin(b + ".attention.input_proj", 3 * d, d)
            lin(b + ".attention.output_proj", d, d)
            ln(b + ".feedforward_layernorm", d)
            lin(b + ".feedforward.model.0", ff, d)
            lin(b + ".feedforward.model.2", d, ff)

    # image tower
    lin("vit.patch_embed", cfg.d_v, 3 * cfg.patch * cfg.patch)
    p["vit.cls_token"] = torch.randn(cfg.d_v, generator=g) * std
    p["vit.pos_embed"] = torch.randn(cfg.n_tok_v, cfg.d_v, generator=g) * std
    enc("vit", cfg.n_layer_v, cfg.d_v, cfg.ff_v)
    ln("vit.final_layer_norm", cfg.d_v)
    # MoE (swin.py:82-92, 11-30)
    lin("moe.router.0", cfg.router_hidden, cfg.d_v)
    lin("moe.router.2", cfg.n_expert, cfg.router_hidden)
    for e in range(cfg.n_expert):
        for s in range(4):
            lin(f"moe.experts.{e}.proj_convs.{s}.0", cfg.d_out, cfg.d_v)
        lin(f"moe.experts.{e}.attn_proj.0", cfg.d_out // 2, cfg.d_out)
        lin(f"moe.experts.{e}.attn_proj.2", 1, cfg.d_out // 2)
    # text tower
    p["text.word_embeddings"] = torch.randn(cfg.vocab, cfg.d_t, generator=g) * std
    p["text.word_embeddings"][0].zero_()          # padding_idx row (init rule :306-307)
    p["text.position_embeddings"] = torch.randn(cfg.max_len, cfg.d_t, generator=g) * std
    p["text.token_type_embeddings"] = torch.randn(2, cfg.d_t, generator=g) * std
    ln("text.emb_layernorm", cfg.d_t)
    enc("text", cfg.n_layer_t, cfg.d_t, cfg.ff_t)
    return p


# --------------------------------------------------------------------------------------
# transformer blocks
# --------------------------------------------------------------------------------------
def fp32_layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """normalizations.py:8-19 — statistics in fp32, cast back to x's dtype."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).type_as(x)


def mhsa(x: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor, n_head: int,
         key_mask: Optional[Tensor] = None) -> Tensor:
    """multi_head_attention.py:40-81.  key_mask: bool [B, T], True = key takes part
    (the reference takes a [B,1|H,T,T] mask; the path only ever uses key padding)."""
    B, N, D = x.shape
    hd = D // n_head
    qkv = F.linear(x, w_in, b_in)                                   # :61
    q, k, v = qkv.chunk(3, dim=-1)                                  # :62
    q = q.view(B, N, n_head, hd).transpose(1, 2)                    # :65-69
    k = k.view(B, N, n_head, hd).transpose(1, 2)
    v = v.view(B, N, n_head, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)        # SDPA :75-77
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, N, D)         # :78
    return F.linear(o, w_out, b_out)                                # :80


def encoder_layer(x: Tensor, p: Dict[str, Tensor], prefix: str, n_head: int, eps: float,
                  norm_first: bool, key_mask: Optional[Tensor] = None) -> Tensor:
    """transformer.py:98-130 with GELU feed-forward (mlp.py:13-66, dropout 0)."""
    g = lambda n: p[f"{prefix}.{n}"]

    def attn(h):
        return mhsa(h, g("attention.input_proj.weight"), g("attention.input_proj.bias"),
                    g("attention.output_proj.weight"), g("attention.output_proj.bias"),
                    n_head, key_mask)

    def ffn(h):
        h = F.gelu(F.linear(h, g("feedforward.model.0.weight"), g("feedforward.model.0.bias")))
        return F.linear(h, g("feedforward.model.2.weight"), g("feedforward.model.2.bias"))

    ln1 = lambda h: fp32_layer_norm(h, g("attention_layernorm.weight"),
                                    g("attention_layernorm.bias"), eps)
    ln2 = lambda h: fp32_layer_norm(h, g("feedforward_layernorm.weight"),
                                    g("feedforward_layernorm.bias"), eps)
    if norm_first:                       # _forward_prenorm :98-114
        r = attn(ln1(x)) + x
        return r + ffn(ln2(r))
    r = ln1(attn(x) + x)                 # _forward_postnorm :116-130
    return ln2(r + ffn(r))


def encoder(x: Tensor, p: Dict[str, Tensor], prefix: str, n_layer: int, n_head: int, eps: float,
            norm_first: bool, key_mask: Optional[Tensor] = None,
            final_ln: bool = False) -> Tuple[Tensor, List[Tensor]]:
    """transformer.py:224-256: returns (last_hidden_state, hidden_states[L+1]); the last
    list entry is the last block's output BEFORE the final LN."""
    hs = []
    for i in range(n_layer):
        hs.append(x)
        x = encoder_layer(x, p, f"{prefix}.layer.{i}", n_head, eps, norm_first, key_mask)
    hs.append(x)
    if final_ln:
        x = fp32_layer_norm(x, p[f"{prefix}.final_layer_norm.weight"],
                            p[f"{prefix}.final_layer_norm.bias"], eps)
    return x, hs


# --------------------------------------------------------------------------------------
# image tower (BUILD-DEFINED embedding; blocks = transformer.py)
# --------------------------------------------------------------------------------------
def patchify(images: Tensor, patch: int) -> Tensor:
    """[B,3,H,W] -> [B, P, 3*patch*patch], inner order (c, py, px) = Conv2d weight order."""
    B, C, H, W = images.shape
    x = images.view(B, C, H // patch, patch, W // patch, patch)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(B, (H // patch) * (W // patch), C * patch * patch)


def vit_forward(images: Tensor, p: Dict[str, Tensor], cfg: OracleConfig):
    """Returns (last_hidden [B,N,Dv] after final LN, hidden_states list)."""
    x = F.linear(patchify(images, cfg.patch), p["vit.patch_embed.weight"], p["vit.patch_embed.bias"])
    cls = p["vit.cls_token"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1) + p["vit.pos_embed"][None]
    return encoder(x, p, "vit", cfg.n_layer_v, cfg.n_head_v, cfg.eps_v, True, None, True)


# --------------------------------------------------------------------------------------
# MoE (swin.py:11-117)
# --------------------------------------------------------------------------------------
def router_probs(x: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """swin.py:98-99: softmax(W2 relu(W1 x + b1) + b2)."""
    h = F.relu(F.linear(x, p["moe.router.0.weight"], p["moe.router.0.bias"]))
    return torch.softmax(F.linear(h, p["moe.router.2.weight"], p["moe.router.2.bias"]), dim=-1)


def topk_lowest_index(probs: Tensor, k: int) -> Tensor:
    """swin.py:100 argmax of the PROBABILITIES; k>1 BUILD-DEFINED: repeated first-max
    (ties -> lowest index), order = descending probability."""
    pr = probs.clone()
    out = []
    for _ in range(k):
        i = torch.argmax(pr, dim=-1)            # first occurrence of the max
        out.append(i)
        pr.scatter_(1, i[:, None], -1.0)
    return torch.stack(out, dim=1)


def gates_from_probs(probs: Tensor, idx: Tensor) -> Tensor:
    """k=1: gate == 1 (swin.py:108 — selected expert output is NOT scaled);
    k>1 BUILD-DEFINED: selected probabilities renormalised to sum 1."""
    if idx.shape[1] == 1:
        return torch.ones(idx.shape, dtype=probs.dtype)
    sel = torch.gather(probs, 1, idx)
    return sel / sel.sum(dim=1, keepdim=True)


def fake_quant_rows(x: Tensor) -> Tensor:
    """BUILD-DEFINED (BASELINE configs[4], no reference counterpart): OCP e4m3fn fake quantisation with ONE scale per row
    (last dimension) = amax / 448, round to nearest even; straight-through gradient.  Used for the expert weights (rows =
    output channels) and for the activation rows entering an expert projection, exactly as medmoe_amd/csrc/fp8.hip does."""
    amax = x.detach().abs().amax(dim=-1, keepdim=True)
    s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (x.detach() * (1.0 / s)).to(torch.float8_e4m3fn).float() * s
    return x + (q - x.detach())


def expert_forward(feats: Sequence[Tensor], p: Dict[str, Tensor], e: int, fp8: bool = False) -> Tensor:
    """swin.py:32-80.  feats: 4 x [n, P_s, D_s] -> [n, P, d_out].  fp8: the two projections run on fake-quantised
    weights and activation rows (BUILD-DEFINED, see fake_quant_rows); the 384 -> 1 logit layer stays fp32."""
    fq = fake_quant_rows if fp8 else (lambda t: t)
    max_len = max(f.shape[1] for f in feats)
    ups = []
    for s, f in enumerate(feats):
        w = p[f"moe.experts.{e}.proj_convs.{s}.0.weight"]
        w = w.reshape(w.shape[0], -1)
        g = F.relu(F.linear(fq(f), fq(w), p[f"moe.experts.{e}.proj_convs.{s}.0.bias"]))   # :41 (k=1 conv)
        if g.shape[1] != max_len:                                                # :42
            g = F.interpolate(g.transpose(1, 2), size=max_len, mode="linear",
                              align_corners=False).transpose(1, 2)
        ups.append(g)
    fused = torch.stack(ups, dim=2)                                              # [n,P,S,D] :50-54
    h = F.relu(F.linear(fq(fused), fq(p[f"moe.experts.{e}.attn_proj.0.weight"]),
                        p[f"moe.experts.{e}.attn_proj.0.bias"]))
    logit = F.linear(h, p[f"moe.experts.{e}.attn_proj.2.weight"],
                     p[f"moe.experts.{e}.attn_proj.2.bias"]).squeeze(-1)         # [n,P,S] :62-63
    w = torch.softmax(logit, dim=-1)                                             # :67
    return (fused * w.unsqueeze(-1)).sum(dim=2)                                  # :78-80


def moe_forward(feats: Sequence[Tensor], router_in: Tensor, p: Dict[str, Tensor],
                n_expert: int, top_k: int, fp8: bool = False):
    """swin.py:94-117, computing only the selected experts (the dense-all-experts +
    gather of :105-108 gives identical values for the selected rows).
    Returns (global [B,D], local [B,D,H,W], probs [B,E], idx [B,k])."""
    probs = router_probs(router_in, p)
    idx = topk_lowest_index(probs.detach(), top_k)
    gates = gates_from_probs(probs, idx)
    B = router_in.shape[0]
    out = None
    for e in range(n_expert):
        for j in range(top_k):
            sel = (idx[:, j] == e).nonzero(as_tuple=True)[0]
            if sel.numel() == 0:
                continue
            y = expert_forward([f[sel] for f in feats], p, e, fp8)
            if out is None:
                out = torch.zeros(B, y.shape[1], y.shape[2], dtype=y.dtype)
            out = out.index_add(0, sel, y * gates[sel, j][:, None, None])
    P, D = out.shape[1], out.shape[2]
    H = int(P ** 0.5)
    return out.mean(dim=1), out.transpose(1, 2).reshape(B, D, H, H), probs, idx   # :110-113


def image_tower(images: Tensor, p: Dict[str, Tensor], cfg: OracleConfig):
    """swin.py:130-149 with the ViT mapping of SURVEY 8a row a3."""
    last, hs = vit_forward(images, p, cfg)
    router_in = last[:, 1:, :].mean(dim=1)                       # swin.py:137 (patch tokens)
    feats = [hs[l][:, 1:, :] for l in cfg.stage_layers()]        # swin.py:139
    return moe_forward(feats, router_in, p, cfg.n_expert, cfg.top_k, cfg.expert_fp8)


# --------------------------------------------------------------------------------------
# text tower (text_encoder.py:32-144)
# --------------------------------------------------------------------------------------
def text_hidden_states(ids: Tensor, attn_mask: Tensor, token_type: Tensor,
                       p: Dict[str, Tensor], cfg: OracleConfig) -> List[Tensor]:
    """BUILD-DEFINED BERT-style embeddings + transformer.py post-norm encoder."""
    T = ids.shape[1]
    x = p["text.word_embeddings"][ids] + p["text.position_embeddings"][:T][None] \
        + p["text.token_type_embeddings"][token_type]
    x = fp32_layer_norm(x, p["text.emb_layernorm.weight"], p["text.emb_layernorm.bias"], cfg.eps_t)
    _, hs = encoder(x, p, "text", cfg.n_layer_t, cfg.n_head_t, cfg.eps_t, False,
                    attn_mask.bool(), False)
    return hs


@dataclass
class Vocab:
    """What aggregate_tokens needs from a tokenizer vocabulary (text_encoder.py:23,47-74)."""
    is_continuation: np.ndarray      # bool[V]: token string starts with '##'
    starts_bracket: np.ndarray       # bool[V]: token string starts with '['
    sep_id: int = 2
    cls_id: int = 1
    pad_id: int = 0

    @staticmethod
    def synthetic(vocab: int, n_continuation: int = 0) -> "Vocab":
        """ids 0,1,2 = [PAD],[CLS],[SEP]; the LAST n_continuation ids are '##' pieces."""
        cont = np.zeros(vocab, dtype=bool)
        if n_continuation:
            cont[vocab - n_continuation:] = True
        br = np.zeros(vocab, dtype=bool)
        br[:3] = True
        return Vocab(cont, br)


def segment_map(ids: np.ndarray, vocab: Vocab) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Word index of every token per text_encoder.py:45-74 (-1 = token dropped).
    Returns (seg [B,T] int32, n_words [B], cap_lens [B]) where cap_lens follows
    medmoe_module.py:221-223 (#words not starting with '[' , + 1)."""
    B, T = ids.shape
    seg = -np.ones((B, T), dtype=np.int32)
    n_words = np.zeros(B, dtype=np.int32)
    cap = np.zeros(B, dtype=np.int32)
    for b in range(B):
        w = -1                      # index of the word whose bank is open
        flushed = 0
        first_br = []
        closed = False
        for t in range(T):
            tid = int(ids[b, t])
            if tid == vocab.sep_id:
                # flush the open bank (:50-54), then [SEP] is its own word (:56-58)
                w += 1
                seg[b, t] = w
                flushed = w + 1
                first_br.append(True)
                closed = True
                break
            if not vocab.is_continuation[tid]:
                w += 1              # :60-71 new word (first token opens bank 0)
                first_br.append(bool(vocab.starts_bracket[tid]))
            elif w < 0:
                w = 0               # '##' piece with an empty bank just joins it (:72-75)
                first_br.append(False)
            seg[b, t] = w
        if not closed:
            # no [SEP]: the loop ends without flushing the last bank (:45-76) -> dropped
            seg[b][seg[b] == w] = -1
            flushed = max(w, 0)
            first_br = first_br[:flushed]
        n_words[b] = flushed
        cap[b] = sum(1 for x in first_br[:flushed] if not x) + 1
    return seg, n_words, cap


def aggregate_last_layers(hs: List[Tensor], seg: np.ndarray, last_n: int):
    """text_encoder.py:97-117 with aggregate_method='sum', agg_tokens=True.
    Returns (word_emb [B,D,T], sent_emb [B,D])."""
    emb = torch.stack(hs[-last_n:], dim=1)                     # [B, L4, T, D]
    B, L4, T, D = emb.shape
    segt = torch.from_numpy(seg.astype(np.int64))
    agg = torch.zeros(B, L4, T + 1, D, dtype=emb.dtype)
    idx = torch.where(segt >= 0, segt, torch.full_like(segt, T))   # dropped -> trash slot
    agg = agg.scatter_add(2, idx[:, None, :, None].expand(B, L4, T, D), emb)[:, :, :T]
    sent = agg.mean(dim=2).sum(dim=1)                          # :110,114
    word = agg.sum(dim=1)                                      # :113
    return word.permute(0, 2, 1), sent                         # :130


def text_tower(ids: Tensor, attn_mask: Tensor, token_type: Tensor, p, cfg: OracleConfig,
               vocab: Vocab):
    hs = text_hidden_states(ids, attn_mask, token_type, p, cfg)
    seg, n_words, cap = segment_map(ids.numpy(), vocab)
    word, sent = aggregate_last_layers(hs, seg, cfg.last_n_layers)
    return word, sent, cap


# --------------------------------------------------------------------------------------
# losses (losses.py)
# --------------------------------------------------------------------------------------
def gloria_global(img_g: Tensor, txt_g: Tensor, temp3: float = 10.0, eps: float = 1e-8) -> Tensor:
    """losses.py:766-794: returns loss0 + loss1 (SUM of the two CE means)."""
    n_i = img_g.norm(dim=-1, keepdim=True)
    n_t = txt_g.norm(dim=-1, keepdim=True)
    s = (img_g @ txt_g.t()) / (n_i @ n_t.t()).clamp(min=eps) * temp3
    lab = torch.arange(img_g.shape[0])
    return F.cross_entropy(s, lab) + F.cross_entropy(s.t(), lab)


def gloria_local_sim(img_l: Tensor, words: Tensor, cap_lens: Sequence[int], temp1: float,
                     temp2: float, eps: float = 1e-8):
    """losses.py:979-1012 + attention_fn :698-736 + cosine_similarity :690-695,
    vectorised over the caption loop.  img_l [B,D,H,W], words [Bc,D,T].
    Returns (sim [B_img, B_cap] BEFORE temp3, att [B_img, B_cap, T, HW])."""
    B, D = img_l.shape[:2]
    ctx = img_l.reshape(B, D, -1)
    T = words.shape[2]
    cl = torch.as_tensor(np.asarray(cap_lens), dtype=torch.long)
    tmask = torch.arange(T)[None, :] < cl[:, None]                       # [Bc, T]
    s = torch.einsum("bdh,idt->biht", ctx, words)                        # :713
    s = s.masked_fill(~tmask[None, :, None, :], float("-inf"))
    a1 = torch.softmax(s, dim=-1)                                        # :716 over words
    a2 = torch.softmax(a1 * temp1, dim=2)                                # :724-725 over HW
    wctx = torch.einsum("bdh,biht->bidt", ctx, a2)                       # :732
    w12 = (words[None] * wctx).sum(dim=2)                                # cosine :692-695
    den = (words.norm(dim=1)[None] * wctx.norm(dim=2)).clamp(min=eps)
    e = torch.exp(w12 / den * temp2) * tmask[None]                       # :1005
    sim = torch.log(e.sum(dim=-1))                                       # :1006-1010 (agg=sum)
    return sim, a2.permute(0, 1, 3, 2)


def gloria_local(img_l: Tensor, words: Tensor, cap_lens: Sequence[int], temp1: float = 4.0,
                 temp2: float = 5.0, temp3: float = 10.0):
    """losses.py:961-1026 -> (loss0, loss1, att_maps list of [1,T_i,H,W])."""
    sim, att = gloria_local_sim(img_l, words, cap_lens, temp1, temp2)
    sim = sim * temp3                                                    # :1015
    lab = torch.arange(img_l.shape[0])
    loss0 = F.cross_entropy(sim, lab)                                    # :1020
    loss1 = F.cross_entropy(sim.t(), lab)                                # :1021
    H, W = img_l.shape[2:]
    maps = [att[i, i, : int(cap_lens[i])].reshape(1, int(cap_lens[i]), H, W)
            for i in range(words.shape[0])]                              # :993-995
    return loss0, loss1, maps


def contrastive_with_temperature(a_local: Tensor, b_local: Tensor, a_all: Tensor, b_all: Tensor,
                                 logit_scale: Tensor, rank: int = 0):
    """losses.py:527-592 given already-gathered embeddings (:503-524); labels are
    B_loc*rank + arange(B_loc) (:516-518).  Returns (loss, logits_a, logits_b, loss_a, loss_b)."""
    t = torch.exp(logit_scale)
    la = a_local @ b_all.t() * t
    lb = b_local @ a_all.t() * t
    n = a_local.shape[0]
    lab = n * rank + torch.arange(n)
    loss_a = F.cross_entropy(la, lab)
    loss_b = F.cross_entropy(lb, lab)
    return (loss_a + loss_b) / 2, la, lb, loss_a, loss_b


def soft_gloria_head(x: Tensor, soft: Tensor, t1: float, t2: float):
    """The loop both Soft-GLoRIA losses end with (losses.py:856-883 global, :1175-1208 local), row by row as the reference writes it:
    for caption-similarity row `soft[layer]`, positives = columns above t1, negatives = columns at or below t2; every positive j is scored
    against the negatives with softXEnt(one-hot at 0, [x_j, x_neg]) = -log_softmax(.)[0] / (1 + #neg) (losses.py:796-803: the division is by
    logits.shape[0] of a 1-D vector), averaged over the positives, then over the rows.  x = image-to-text similarities already multiplied by
    temp3; its transpose gives the text-to-image term.  Returns (loss0, loss1)."""
    B = x.shape[0]
    x1 = x.t()
    loss0 = x.new_zeros(())
    loss1 = x.new_zeros(())
    for layer in range(B):
        pos = (soft[layer] > t1).nonzero().squeeze(-1)
        neg = (soft[layer] <= t2).nonzero().squeeze(-1)
        li = x.new_zeros(())
        lt = x.new_zeros(())
        for j in pos:
            a = torch.cat([x[layer][j].unsqueeze(-1), x[layer][neg]])
            b = torch.cat([x1[layer][j].unsqueeze(-1), x1[layer][neg]])
            li = li - torch.log_softmax(a, dim=-1)[0] / a.shape[0]
            lt = lt - torch.log_softmax(b, dim=-1)[0] / b.shape[0]
        loss0 = loss0 + li / len(pos)
        loss1 = loss1 + lt / len(pos)
    return loss0 / B, loss1 / B


def soft_gloria_global(img_g: Tensor, txt_g: Tensor, soft: Tensor, thresholds, temp3: float = 10.0, eps: float = 1e-8) -> Tensor:
    """SoftGLORIAGlobalContrastiveLoss.forward, losses.py:826-883 (idx = soft scores, probs = (threshold1, threshold2)): loss0 + loss1."""
    n_i = img_g.norm(dim=-1, keepdim=True)
    n_t = txt_g.norm(dim=-1, keepdim=True)
    s = (img_g @ txt_g.t()) / (n_i @ n_t.t()).clamp(min=eps) * temp3
    l0, l1 = soft_gloria_head(s, soft, thresholds[0], thresholds[1])
    return l0 + l1


def soft_gloria_local(img_l: Tensor, words: Tensor, cap_lens: Sequence[int], soft: Tensor, thresholds, temp1: float = 4.0,
                      temp2: float = 5.0, temp3: float = 10.0):
    """SoftGLORIALocalContrastiveLoss.forward, losses.py:1122-1214: the similarities of GLORIALocalContrastiveLoss (:1139-1172 repeat
    :979-1012) under the soft head -> (loss0, loss1, att_maps)."""
    sim, att = gloria_local_sim(img_l, words, cap_lens, temp1, temp2)
    l0, l1 = soft_gloria_head(sim * temp3, soft, thresholds[0], thresholds[1])
    H, W = img_l.shape[2:]
    maps = [att[i, i, : int(cap_lens[i])].reshape(1, int(cap_lens[i]), H, W) for i in range(words.shape[0])]
    return l0, l1, maps


def text_soft_target(last_hidden: Tensor) -> Tensor:
    """medmoe_module.py:258-281 get_text_soft_target: the frozen `tool_bert`'s last hidden state, pooled by token_pooling (:243-244: the
    [CLS] position), L2-normalised, caption-to-caption products.  With `freeze_bert: true` tool_bert and the text encoder's BERT are the
    same pretrained weights, so `last_hidden` is the text tower's last layer."""
    f = F.normalize(last_hidden[:, 0], p=2, dim=1)
    return f @ f.t()


def hard_negative(imgs: Tensor, caps: Tensor, margin: float = 0.2, nmax: int = 1) -> Tensor:
    """HardNegativeContrastiveLoss.forward, losses.py:885-927 (the VSE++-style alternative of the global loss listed in
    med-moe_pretraining.yaml:33): cosine scores, the diagonal pushed down by twice its value so that it is not picked (:903), the nmax
    largest entries per column / row against the positive with a margin, summed."""
    caps = F.normalize(caps, dim=-1)
    imgs = F.normalize(imgs, dim=-1)
    scores = imgs @ caps.t()
    diag = scores.diag()
    scores = scores - 2 * torch.diag(scores.diag())
    sorted_cap, _ = torch.sort(scores, 0, descending=True)
    sorted_img, _ = torch.sort(scores, 1, descending=True)
    max_c = sorted_cap[:nmax, :]
    max_i = sorted_img[:, :nmax]
    neg_cap = torch.clamp(max_c + (margin - diag).view(1, -1).expand_as(max_c), min=0).sum()
    neg_img = torch.clamp(max_i + (margin - diag).view(-1, 1).expand_as(max_i), min=0).sum()
    return neg_cap + neg_img


def router_ce(probs: Tensor, labels: Tensor) -> Tensor:
    """medmoe_module.py:235-237 — CE applied to ALREADY-SOFTMAXED probabilities."""
    return F.cross_entropy(probs, labels)


# --------------------------------------------------------------------------------------
# the whole step (medmoe_module.py:284-316, med_moe.py:102-108)
# --------------------------------------------------------------------------------------
def model_step(batch: Dict[str, Tensor], p: Dict[str, Tensor], cfg: OracleConfig, vocab: Vocab):
    img_g, img_l, probs, idx = image_tower(batch["image"], p, cfg)
    with torch.set_grad_enabled(not getattr(cfg, "freeze_text", True)):      # freeze_bert: true (configs/model/med-moe.yaml:35) unless told otherwise
        txt_l, txt_g, cap = text_tower(batch["ids"], batch["attn_mask"], batch["token_type"],
                                       p, cfg, vocab)
    soft = None
    if getattr(cfg, "soft_label", False):                                # :291-296
        with torch.no_grad():
            soft = text_soft_target(text_hidden_states(batch["ids"], batch["attn_mask"], batch["token_type"], p, cfg)[-1])
        thr = (cfg.threshold0, cfg.threshold1)
        l0, l1, _ = soft_gloria_local(img_l, txt_l, cap, soft, thr, cfg.temp1, cfg.temp2, cfg.temp3)
        g_loss = soft_gloria_global(img_g, txt_g, soft, thr, cfg.temp3)
    else:
        l0, l1, _ = gloria_local(img_l, txt_l, cap, cfg.temp1, cfg.temp2, cfg.temp3)
        g_loss = gloria_global(img_g, txt_g, cfg.temp3)                  # :213-217
    l_loss = l0 + l1                                                     # :233
    c_loss = router_ce(probs, batch["label"])                            # :305
    acc = (probs.argmax(dim=1) == batch["label"]).float().mean()         # :239-241
    loss = cfg.w_local * l_loss + cfg.w_global * g_loss + cfg.w_cls * c_loss   # :308
    return {"loss": loss, "l_loss": l_loss, "g_loss": g_loss, "classifier_loss": c_loss,
            "classifier_acc": acc, "img_g": img_g, "img_l": img_l, "txt_g": txt_g,
            "txt_l": txt_l, "probs": probs, "idx": idx, "cap_lens": cap, "soft": soft}


def synthetic_batch(cfg: OracleConfig, B: int, seed: int = 12345, min_len: int = 8):
    """SURVEY 8d synthetic inputs: randn images, ids with [CLS]=1 first, [SEP]=2 at
    len-1, [PAD]=0 after, len ~ U{min_len..T}; labels randint(0,E)."""
    g = torch.Generator().manual_seed(seed)
    T = cfg.max_len
    img = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=g)
    lens = torch.randint(min(min_len, T), T + 1, (B,), generator=g)
    ids = torch.randint(3, cfg.vocab, (B, T), generator=g)
    pos = torch.arange(T)[None]
    ids[:, 0] = 1
    ids = torch.where(pos == (lens[:, None] - 1), torch.full_like(ids, 2), ids)
    ids = torch.where(pos >= lens[:, None], torch.zeros_like(ids), ids)
    mask = (pos < lens[:, None]).long()
    label = torch.randint(0, cfg.n_expert, (B,), generator=g)
    return {"image": img, "ids": ids, "attn_mask": mask, "token_type": torch.zeros_like(ids),
            "label": label}


# --------------------------------------------------------------------------------------
# bit-exact router restatement (fixed fp32 summation order; numpy, no FMA contraction)
# --------------------------------------------------------------------------------------
def router_fixed_order(x: np.ndarray, w1: np.ndarray, b1: np.ndarray, w2: np.ndarray,
                       b2: np.ndarray, k: int):
    """swin.py:98-100 with every fp32 operation order pinned so a GPU kernel can match the
    LOGITS bit for bit: acc = bias; for j ascending: acc = fl(acc + fl(x[j]*w[.,j])).
    softmax = exp(l - max) / sum (sum ascending in e).  Returns (probs f32 [B,E], idx i32 [B,k], logits, hidden)."""
    x = x.astype(np.float32)
    B = x.shape[0]
    h = np.broadcast_to(b1.astype(np.float32), (B, w1.shape[0])).copy()
    for j in range(x.shape[1]):
        h = (h + (x[:, j:j + 1] * w1[None, :, j].astype(np.float32)).astype(np.float32)).astype(np.float32)
    h = np.maximum(h, np.float32(0))
    l = np.broadcast_to(b2.astype(np.float32), (B, w2.shape[0])).copy()
    for j in range(h.shape[1]):
        l = (l + (h[:, j:j + 1] * w2[None, :, j].astype(np.float32)).astype(np.float32)).astype(np.float32)
    m = l.max(axis=1, keepdims=True)
    e = np.exp((l - m).astype(np.float32)).astype(np.float32)
    s = np.zeros((B, 1), dtype=np.float32)
    for j in range(e.shape[1]):
        s = (s + e[:, j:j + 1]).astype(np.float32)
    probs = (e / s).astype(np.float32)
    pr = probs.copy()
    idx = np.zeros((B, k), dtype=np.int32)
    for j in range(k):
        i = pr.argmax(axis=1)           # first max = lowest index on ties
        idx[:, j] = i
        pr[np.arange(B), i] = -1.0
    return probs, idx, l, h
