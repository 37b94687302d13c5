"""Parameter storage for the MedMoE hot path on one MI355X.

Trainable (image tower + MoE) parameters live in ONE flat fp32 master buffer with matching
flat fp32 grad / Adam-m / Adam-v buffers (a single fused clip+Adam launch, a single RCCL
all-reduce), plus a flat bf16 working copy in the reference's nn.Linear [out,in] layout and a
second bf16 buffer holding every GEMM weight TRANSPOSED ([in,out]) so dgrad is the same NT MFMA
kernel as the forward.  Expert weights are stacked [E, ...] for the grouped GEMMs.

Names follow the reference modules (transformer.py / multi_head_attention.py / mlp.py /
swin.py) so state_dicts map 1:1; `load_named` / `export_named` translate the per-expert
reference names (moe.experts.{e}.proj_convs.{s}.0.weight ...) to the stacked storage.
The text tower is frozen (configs/model/med-moe.yaml:35) and kept separately.
"""
from typing import Dict, List, Tuple

import torch

from . import ops
from .config import MedMoEConfig

_ALIGN = 8


class ParamStore:
    def __init__(self, cfg: MedMoEConfig, device, seed: int = 0, std: float = 0.02):
        cfg.validate()
        self.cfg = cfg
        self.device = torch.device(device)
        self.specs: List[Tuple[str, Tuple[int, ...], str]] = []   # (name, shape, kind) kind: w|wt|v
        c = cfg
        pp = c.patch_dim_pad           # padded to the GEMM k-step; the pad columns stay zero (zero im2col columns -> zero grads)
        E, Do, Dh, Dv = c.n_expert, c.d_out, c.d_out // 2, c.d_v

        def add(name, shape, kind):
            self.specs.append((name, tuple(shape), kind))

        add("vit.patch_embed.weight", (Dv, pp), "w")
        add("vit.patch_embed.bias", (Dv,), "v")
        add("vit.cls_token", (Dv,), "v")
        add("vit.pos_embed", (c.n_tok_v, Dv), "v")
        for i in range(c.n_layer_v):
            b = f"vit.layer.{i}"
            add(b + ".attention_layernorm.weight", (Dv,), "v"); add(b + ".attention_layernorm.bias", (Dv,), "v")
            add(b + ".attention.input_proj.weight", (3 * Dv, Dv), "wt"); add(b + ".attention.input_proj.bias", (3 * Dv,), "v")
            add(b + ".attention.output_proj.weight", (Dv, Dv), "wt"); add(b + ".attention.output_proj.bias", (Dv,), "v")
            add(b + ".feedforward_layernorm.weight", (Dv,), "v"); add(b + ".feedforward_layernorm.bias", (Dv,), "v")
            add(b + ".feedforward.model.0.weight", (c.ff_v, Dv), "wt"); add(b + ".feedforward.model.0.bias", (c.ff_v,), "v")
            add(b + ".feedforward.model.2.weight", (Dv, c.ff_v), "wt"); add(b + ".feedforward.model.2.bias", (Dv,), "v")
        add("vit.final_layer_norm.weight", (Dv,), "v"); add("vit.final_layer_norm.bias", (Dv,), "v")
        add("moe.router.0.weight", (c.router_hidden, Dv), "v"); add("moe.router.0.bias", (c.router_hidden,), "v")
        add("moe.router.2.weight", (E, c.router_hidden), "v"); add("moe.router.2.bias", (E,), "v")
        for s in range(4):
            add(f"moe.proj.{s}.weight", (E, Do, Dv), "wt"); add(f"moe.proj.{s}.bias", (E, Do), "v")
        add("moe.attn0.weight", (E, Dh, Do), "wt"); add("moe.attn0.bias", (E, Dh), "v")
        add("moe.attn2.weight", (E, Dh), "v"); add("moe.attn2.bias", (E,), "v")

        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shape, _ in self.specs:
            self.offsets[name] = off
            n = 1
            for d in shape:
                n *= d
            off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        dev = self.device
        self.p32 = torch.zeros(off, device=dev, dtype=torch.float32)
        self.g32 = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.v = torch.zeros(off, device=dev, dtype=torch.float32)
        self.p16 = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        self.p16t = torch.zeros(off, device=dev, dtype=torch.bfloat16)
        self.shapes = {n: s for n, s, _ in self.specs}
        self.kinds = {n: k for n, _, k in self.specs}
        # transpose table: one entry per 2-D matrix (per expert for stacked weights)
        rows = []
        for name, shape, kind in self.specs:
            if kind != "wt":
                continue
            o = self.offsets[name]
            if len(shape) == 2:
                rows.append([o, o, shape[0], shape[1]])
            else:
                for e in range(shape[0]):
                    oo = o + e * shape[1] * shape[2]
                    rows.append([oo, oo, shape[1], shape[2]])
        self.tr_table = torch.tensor(rows, device=dev, dtype=torch.int64)
        self.tr_max_tiles = max(((r[2] + 63) // 64) * ((r[3] + 63) // 64) for r in rows)
        self.normsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.norm_scratch = torch.zeros(2049, device=dev, dtype=torch.float32)      # per-block partials + arrival counter (medmoe_sumsq_det)
        self.step_count = 0
        # fp8 expert weights (BASELINE configs[4]): e4m3 copies [E,N,K] + transposes [E,K,N] + per-output-channel scales [E,N] of
        # the five expert projections, re-derived from the fp32 master after every update (requantise_experts)
        self.fp8: Dict[str, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = {}
        if cfg.expert_fp8:
            for name in [f"moe.proj.{s_}.weight" for s_ in range(4)] + ["moe.attn0.weight"]:
                Eg, N, K = self.shapes[name]
                self.fp8[name] = (torch.empty(Eg, N, K, device=dev, dtype=torch.uint8), torch.empty(Eg, K, N, device=dev, dtype=torch.uint8),
                                  torch.empty(Eg, N, device=dev, dtype=torch.float32))
        self._init_random(seed, std)
        # ---- frozen text tower ----
        self.text: Dict[str, torch.Tensor] = {}
        self._init_text(seed + 1, std)

    # -- views ---------------------------------------------------------------------------------
    def _view(self, flat, name, transposed=False):
        shape = self.shapes[name]
        n = 1
        for d in shape:
            n *= d
        t = flat[self.offsets[name]: self.offsets[name] + n]
        if transposed:
            shape = shape[:-2] + (shape[-1], shape[-2])
        return t.view(shape)

    def f32(self, name): return self._view(self.p32, name)
    def grad(self, name): return self._view(self.g32, name)
    def w16(self, name): return self._view(self.p16, name)
    def w16t(self, name): return self._view(self.p16t, name, transposed=True)

    # -- init / sync ---------------------------------------------------------------------------
    def _init_random(self, seed, std):
        g = torch.Generator(device="cpu").manual_seed(seed)
        for name, shape, kind in self.specs:
            if name.endswith("layernorm.weight") or name.endswith("layer_norm.weight"):
                val = torch.ones(shape)
            elif name.endswith(".bias"):
                val = torch.zeros(shape)
            elif name == "vit.patch_embed.weight":
                val = self._pad_patch(torch.randn(shape[0], self.cfg.patch_dim, generator=g) * std)
            else:
                val = torch.randn(shape, generator=g) * std     # init rule multimodal_transformer.py:298-312
            self.f32(name).copy_(val.to(self.device))
        self.sync_working_copies()

    def _init_text(self, seed, std):
        c, dev = self.cfg, self.device
        g = torch.Generator(device="cpu").manual_seed(seed)
        t = self.text

        def rn(*shape):
            return (torch.randn(*shape, generator=g) * std).to(dev)
        t["word_embeddings"] = rn(c.vocab, c.d_t)
        t["word_embeddings"][0].zero_()
        t["position_embeddings"] = rn(c.max_len, c.d_t)
        t["token_type_embeddings"] = rn(2, c.d_t)
        t["emb_layernorm.weight"] = torch.ones(c.d_t, device=dev); t["emb_layernorm.bias"] = torch.zeros(c.d_t, device=dev)
        for i in range(c.n_layer_t):
            b = f"layer.{i}"
            for nm, (o, k) in (("attention.input_proj", (3 * c.d_t, c.d_t)), ("attention.output_proj", (c.d_t, c.d_t)),
                               ("feedforward.model.0", (c.ff_t, c.d_t)), ("feedforward.model.2", (c.d_t, c.ff_t))):
                t[f"{b}.{nm}.weight"] = rn(o, k).to(torch.bfloat16)
                t[f"{b}.{nm}.bias"] = torch.zeros(o, device=dev)
            for nm in ("attention_layernorm", "feedforward_layernorm"):
                t[f"{b}.{nm}.weight"] = torch.ones(c.d_t, device=dev); t[f"{b}.{nm}.bias"] = torch.zeros(c.d_t, device=dev)

    def _pad_patch(self, w):
        c = self.cfg
        w = w.reshape(c.d_v, c.patch_dim)
        if c.patch_dim_pad == c.patch_dim:
            return w
        return torch.cat([w, w.new_zeros(c.d_v, c.patch_dim_pad - c.patch_dim)], 1)

    def sync_working_copies(self):
        """fp32 master -> bf16 [out,in] copy and the transposed bf16 copy (+ the e4m3 expert copies)."""
        ops.call("cast_bf16", self.p32, self.p16, self.numel)
        ops.call("transpose_many", self.p16, self.p16t, self.tr_table, self.tr_table.shape[0], self.tr_max_tiles)
        self.requantise_experts()

    def requantise_experts(self):
        for name, (q, qT, s) in self.fp8.items():
            Eg, N, K = self.shapes[name]
            ops.call("quant_weights_e4m3", self.f32(name), q, qT, s, Eg, N, K)

    def q8(self, name): return self.fp8[name][0]
    def q8t(self, name): return self.fp8[name][1]
    def s8(self, name): return self.fp8[name][2]

    # -- reference-style names ------------------------------------------------------------------
    def load_named(self, named: Dict[str, torch.Tensor]):
        """Load a reference-style name->tensor dict (e.g. oracle.init_params / a state_dict)."""
        c = self.cfg
        dev = self.device
        used = set()
        for name in self.shapes:
            if name.startswith("moe.proj.") or name.startswith("moe.attn0") or name.startswith("moe.attn2"):
                continue
            if name in named:
                v = named[name].to(dev)
                if name == "vit.patch_embed.weight":
                    v = self._pad_patch(v)
                self.f32(name).copy_(v.reshape(self.shapes[name])); used.add(name)
        for e in range(c.n_expert):
            for s in range(4):
                w = named[f"moe.experts.{e}.proj_convs.{s}.0.weight"].to(dev)
                self.f32(f"moe.proj.{s}.weight")[e].copy_(w.reshape(c.d_out, c.d_v))
                self.f32(f"moe.proj.{s}.bias")[e].copy_(named[f"moe.experts.{e}.proj_convs.{s}.0.bias"].to(dev))
            self.f32("moe.attn0.weight")[e].copy_(named[f"moe.experts.{e}.attn_proj.0.weight"].to(dev))
            self.f32("moe.attn0.bias")[e].copy_(named[f"moe.experts.{e}.attn_proj.0.bias"].to(dev))
            self.f32("moe.attn2.weight")[e].copy_(named[f"moe.experts.{e}.attn_proj.2.weight"].to(dev).reshape(-1))
            self.f32("moe.attn2.bias")[e].copy_(named[f"moe.experts.{e}.attn_proj.2.bias"].to(dev).reshape(()))
        self.load_named_text(named)
        self.sync_working_copies()

    def load_named_text(self, named: Dict[str, torch.Tensor]):
        """`text.<name>` entries of a reference-style dict into the frozen text tower (in place: views handed out earlier stay valid)."""
        for k, v in named.items():
            if k.startswith("text."):
                kk = k[len("text."):]
                if kk not in self.text:
                    raise KeyError(f"unknown text parameter {k}")
                if tuple(v.shape) != tuple(self.text[kk].shape):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(self.text[kk].shape)}")
                self.text[kk] = v.to(self.device).to(self.text[kk].dtype).contiguous()

    def named_views(self, flat=None) -> Dict[str, torch.Tensor]:
        """Reference-style names -> VIEWS of the flat buffer on the device (flat = p32 for weights, g32 for grads): the per-expert
        names of swin.py:12-30 are slices of the stacked [E, ...] storage; the patch-embedding weight loses its k-step padding."""
        flat = self.p32 if flat is None else flat
        c = self.cfg
        out = {}
        for name in self.shapes:
            v = self._view(flat, name)
            if name.startswith("moe.proj."):
                s = int(name.split(".")[2]); leaf = name.split(".")[3]
                for e in range(c.n_expert):
                    t = v[e]
                    out[f"moe.experts.{e}.proj_convs.{s}.0.{leaf}"] = t.reshape(c.d_out, c.d_v, 1) if leaf == "weight" else t
            elif name.startswith("moe.attn0."):
                leaf = name.split(".")[2]
                for e in range(c.n_expert):
                    out[f"moe.experts.{e}.attn_proj.0.{leaf}"] = v[e]
            elif name.startswith("moe.attn2."):
                leaf = name.split(".")[2]
                for e in range(c.n_expert):
                    out[f"moe.experts.{e}.attn_proj.2.{leaf}"] = v[e].reshape(1, -1) if leaf == "weight" else v[e].reshape(1)
            elif name == "vit.patch_embed.weight":
                out[name] = v[:, :c.patch_dim]
            else:
                out[name] = v
        return out

    def export_named(self, flat=None) -> Dict[str, torch.Tensor]:
        """Reference-style names -> fp32 CPU tensors (flat = p32 for weights, g32 for grads)."""
        return {k: v.detach().float().cpu().contiguous() for k, v in self.named_views(flat).items()}

    # -- optimiser ------------------------------------------------------------------------------
    def zero_grad(self):
        self.g32.zero_()

    def adam_step(self, lr=None, grad_scale: float = 1.0, extra_normsq=None):
        """clip_grad_norm_(clip) + torch.optim.Adam step, fused, then refresh the bf16 copies.  extra_normsq: the squared gradient norm of
        parameters that live in another store (the trainable text tower) - the clip coefficient is formed from the norm over ALL of them, and
        self.normsq holds that total afterwards."""
        c = self.cfg
        self.step_count += 1
        ops.call("sumsq_det", self.g32, self.numel, self.normsq, self.norm_scratch)     # fixed order: identical on every rank
        if extra_normsq is not None:
            self.normsq.add_(extra_normsq)
        ops.call("adam_step", self.p32, self.g32, self.m, self.v, self.p16, self.numel, c.lr if lr is None else lr,
                 0.9, 0.999, 1e-8, c.weight_decay, self.step_count, self.normsq, c.clip, grad_scale)
        ops.call("transpose_many", self.p16, self.p16t, self.tr_table, self.tr_table.shape[0], self.tr_max_tiles)
        self.requantise_experts()
