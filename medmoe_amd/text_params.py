"""Trainable text tower (reference `freeze_bert: false`, text_encoder.py:27-30 leaves the BERT parameters trainable; configs/model/med-moe.yaml:35
freezes them for the pretraining experiment): the tower's parameters as ONE flat fp32 master with flat gradient / Adam-m / Adam-v buffers, a bf16
working copy in the nn.Linear [out, in] layout and a second one with every GEMM weight transposed (dgrad = the same NT kernel as forward) - the
layout of `ParamStore` for the image tower, so the fused clip + Adam kernels and one more all-reduce bucket serve it.

Names are the text tower's own (`layer.{i}.attention.input_proj.weight`, ..., `word_embeddings`, `position_embeddings`,
`token_type_embeddings`, `emb_layernorm.*`: transformer.py:116-130 post-norm blocks under a BERT-style embedding front-end).  `as_dict()` hands
the engine's forward pass the same name -> tensor mapping the frozen tower uses (bf16 GEMM weights, fp32 everything else), as VIEWS of the flat
buffers: an optimiser step updates them in place."""
from typing import Dict, List, Tuple

import torch

from . import ops
from .config import MedMoEConfig

_ALIGN = 8


class TextStore:
    def __init__(self, cfg: MedMoEConfig, device, init: Dict[str, torch.Tensor]):
        self.cfg, self.device = cfg, torch.device(device)
        c = cfg
        D, ff = c.d_t, c.ff_t
        specs: List[Tuple[str, Tuple[int, ...], str]] = [("word_embeddings", (c.vocab, D), "v"), ("position_embeddings", (c.max_len, D), "v"),
                                                        ("token_type_embeddings", (2, D), "v"), ("emb_layernorm.weight", (D,), "v"),
                                                        ("emb_layernorm.bias", (D,), "v")]
        for i in range(c.n_layer_t):
            b = f"layer.{i}."
            specs += [(b + "attention.input_proj.weight", (3 * D, D), "wt"), (b + "attention.input_proj.bias", (3 * D,), "v"),
                      (b + "attention.output_proj.weight", (D, D), "wt"), (b + "attention.output_proj.bias", (D,), "v"),
                      (b + "attention_layernorm.weight", (D,), "v"), (b + "attention_layernorm.bias", (D,), "v"),
                      (b + "feedforward.model.0.weight", (ff, D), "wt"), (b + "feedforward.model.0.bias", (ff,), "v"),
                      (b + "feedforward.model.2.weight", (D, ff), "wt"), (b + "feedforward.model.2.bias", (D,), "v"),
                      (b + "feedforward_layernorm.weight", (D,), "v"), (b + "feedforward_layernorm.bias", (D,), "v")]
        self.specs = specs
        self.shapes = {n: s for n, s, _ in specs}
        self.kinds = {n: k for n, _, k in specs}
        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shape, _ in specs:
            self.offsets[name] = off
            n = 1
            for d in shape:
                n *= d
            off += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        dev = self.device
        z = lambda dt: torch.zeros(off, device=dev, dtype=dt)
        self.p32, self.g32, self.m, self.v = z(torch.float32), z(torch.float32), z(torch.float32), z(torch.float32)
        self.p16, self.p16t = z(torch.bfloat16), z(torch.bfloat16)
        rows = [[self.offsets[n], self.offsets[n], s[0], s[1]] for n, s, k in specs if k == "wt"]
        self.tr_table = torch.tensor(rows, device=dev, dtype=torch.int64)
        self.tr_max_tiles = max(((r[2] + 63) // 64) * ((r[3] + 63) // 64) for r in rows)
        self.normsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.norm_scratch = torch.zeros(2049, device=dev, dtype=torch.float32)
        self.step_count = 0
        missing = [n for n in self.shapes if n not in init]
        if missing:
            raise KeyError(f"TextStore: no initial value for {missing[:3]} ...")
        for name in self.shapes:
            self.f32(name).copy_(init[name].to(dev).float().reshape(self.shapes[name]))
        self.sync_working_copies()

    def _view(self, flat, name, transposed=False):
        shape = self.shapes[name]
        n = 1
        for d in shape:
            n *= d
        t = flat[self.offsets[name]: self.offsets[name] + n]
        if transposed:
            shape = (shape[1], shape[0])
        return t.view(shape)

    def f32(self, name): return self._view(self.p32, name)
    def grad(self, name): return self._view(self.g32, name)
    def w16(self, name): return self._view(self.p16, name)
    def w16t(self, name): return self._view(self.p16t, name, transposed=True)

    def as_dict(self) -> Dict[str, torch.Tensor]:
        """name -> the tensor the forward pass reads: bf16 [out, in] views for the GEMM weights, fp32 views for embeddings, biases, LayerNorms."""
        return {n: (self.w16(n) if self.kinds[n] == "wt" else self.f32(n)) for n in self.shapes}

    def load_named(self, named: Dict[str, torch.Tensor]):
        """`text.<name>` entries of a reference-style dict into the master buffer (then refresh the working copies)."""
        for k, v in named.items():
            if not k.startswith("text."):
                continue
            kk = k[len("text."):]
            if kk not in self.shapes:
                raise KeyError(f"unknown text parameter {k}")
            if tuple(v.shape) != tuple(self.shapes[kk]):
                raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(self.shapes[kk])}")
            self.f32(kk).copy_(v.to(self.device).float())
        self.sync_working_copies()

    def export_named(self, flat=None) -> Dict[str, torch.Tensor]:
        flat = self.p32 if flat is None else flat
        return {"text." + n: self._view(flat, n).detach().float().cpu().contiguous() for n in self.shapes}

    def sync_working_copies(self):
        ops.call("cast_bf16", self.p32, self.p16, self.numel)
        ops.call("transpose_many", self.p16, self.p16t, self.tr_table, self.tr_table.shape[0], self.tr_max_tiles)

    def zero_grad(self):
        self.g32.zero_()

    def sumsq(self) -> torch.Tensor:
        """Sum of squares of the gradient in a fixed order (identical on every rank): the text tower's share of the global clip norm."""
        ops.call("sumsq_det", self.g32, self.numel, self.normsq, self.norm_scratch)
        return self.normsq

    def adam_step(self, normsq_total: torch.Tensor, lr=None, grad_scale: float = 1.0):
        """clip (against the norm over BOTH towers' gradients, as clip_grad_norm_ over all parameters computes it) + Adam, fused."""
        c = self.cfg
        self.step_count += 1
        ops.call("adam_step", self.p32, self.g32, self.m, self.v, self.p16, self.numel, c.lr if lr is None else lr,
                 0.9, 0.999, 1e-8, c.weight_decay, self.step_count, normsq_total, c.clip, grad_scale)
        ops.call("transpose_many", self.p16, self.p16t, self.tr_table, self.tr_table.shape[0], self.tr_max_tiles)
