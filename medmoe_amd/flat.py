"""Flat parameter arena for the Swin-T encoder (`SwinTower`, `SwinMoEEncoder`): every parameter of a name -> tensor dict as a VIEW of one fp32
buffer, with a gradient buffer of the same layout (one memset zeroes every gradient of a step), a bf16 working copy and a second one holding
every GEMM weight transposed ([in, out]: dgrad runs on the same NT kernel as forward).  `refresh()` is two kernels (cast + batched transpose)
whatever the number of parameters - the per-parameter torch casts, transposes and zero fills it replaces were 1.4 ms of fills and ~270 small
launches of a 19 ms step (profiles/r03_notes.md, ref_swin).

`groups` lays parameters out back to back so that a concatenation the kernels want is a free view: the q / k / v projections of a Swin block
([C, C] each) are one [3C, C] GEMM weight and one [3C] bias (modeling_swin.py SwinSelfAttention keeps them as three nn.Linear).

The layout mirrors `ParamStore` / `TextStore` (the ViT towers' stores); those carry tower-specific specs, this one takes any dict."""
from typing import Dict, List, Sequence, Tuple

import torch

from . import ops

_ALIGN = 8


def _numel(shape) -> int:
    n = 1
    for d in shape:
        n *= int(d)
    return n


class FlatStore:
    def __init__(self, weights: Dict[str, torch.Tensor], device, groups: Sequence[Tuple[str, List[str]]] = (), gemm: Sequence[str] = ()):
        """weights: name -> floating tensor.  groups: (alias, member names) - members are stored contiguously in that order, `alias` then names the
        concatenation along dim 0.  gemm: names or aliases that are GEMM weights ([out, in] or [out, in, 1]): they get a transposed bf16 copy
        (16-byte accesses: a GEMM weight inside a group must have a multiple of 8 elements, as must every member before it)."""
        self.device = dev = torch.device(device)
        self.shapes: Dict[str, Tuple[int, ...]] = {n: tuple(v.shape) for n, v in weights.items()}
        member_of = {m: a for a, ms in groups for m in ms}
        order: List[str] = []
        for a, ms in groups:
            order += ms
        order += [n for n in weights if n not in member_of]
        self.offsets: Dict[str, int] = {}
        off = 0
        pad = lambda o: (o + _ALIGN - 1) // _ALIGN * _ALIGN
        for a, ms in groups:                                          # members back to back (no padding between them), the group padded as a whole
            for n in ms:
                self.offsets[n] = off
                off += _numel(self.shapes[n])
            off = pad(off)
        for n in order:
            if n not in member_of:
                self.offsets[n] = off
                off = pad(off + _numel(self.shapes[n]))
        for a, ms in groups:
            first = self.shapes[ms[0]]
            if any(self.shapes[m][1:] != first[1:] for m in ms):
                raise ValueError(f"FlatStore: group {a} concatenates parameters of different trailing shapes")
            self.shapes[a] = (sum(self.shapes[m][0] for m in ms),) + first[1:]
            self.offsets[a] = self.offsets[ms[0]]
        self.names = list(weights)
        self.numel = off
        self._views: Dict[tuple, tuple] = {}
        z = lambda dt: torch.zeros(off, device=dev, dtype=dt)
        self.p32, self.g32, self.p16, self.p16t = z(torch.float32), z(torch.float32), z(torch.bfloat16), z(torch.bfloat16)
        rows = []
        self._mat: Dict[str, Tuple[int, int]] = {}
        for n in gemm:
            if self.offsets[n] % _ALIGN:
                raise ValueError(f"FlatStore: GEMM weight {n} starts at element {self.offsets[n]}, not a multiple of {_ALIGN}")
            s = self.shapes[n]
            r, c = int(s[0]), _numel(s[1:])
            self._mat[n] = (r, c)
            rows.append([self.offsets[n], self.offsets[n], r, c])
        self.tr_table = torch.tensor(rows, device=dev, dtype=torch.int64) if rows else None
        self.tr_max_tiles = max(((r[2] + 63) // 64) * ((r[3] + 63) // 64) for r in rows) if rows else 0
        for n in self.names:
            self.f32(n).copy_(weights[n].detach().to(dev, torch.float32))
        self.refresh()

    def _view(self, flat, name, shape=None):
        shape = self.shapes[name] if shape is None else shape
        o = self.offsets[name]
        return flat[o: o + _numel(shape)].view(shape)

    def _cached(self, kind, flat, name, shape=None):
        """Views are built once per buffer (a backward asks for a few hundred of them per step)."""
        key = (kind, name)
        hit = self._views.get(key)
        if hit is None or hit[0] is not flat:
            hit = self._views[key] = (flat, self._view(flat, name, shape))
        return hit[1]

    def f32(self, name): return self._cached(0, self.p32, name)
    def grad(self, name): return self._cached(1, self.g32, name)

    def w16(self, name):
        """bf16 [out, in] view of a GEMM weight."""
        return self._cached(2, self.p16, name, self._mat[name])

    def w16t(self, name):
        """bf16 [in, out] view of a GEMM weight (the transposed copy)."""
        r, c = self._mat[name]
        return self._cached(3, self.p16t, name, (c, r))

    def grad2d(self, name):
        return self._cached(4, self.g32, name, self._mat[name])

    def refresh(self):
        """bf16 working copies after the fp32 master changed (an optimizer step, a loaded checkpoint)."""
        ops.call("cast_bf16", self.p32, self.p16, self.numel)
        if self.tr_table is not None:
            ops.call("transpose_many", self.p16, self.p16t, self.tr_table, self.tr_table.shape[0], self.tr_max_tiles)

    def zero_grad(self):
        self.g32.zero_()

    def new_grad_arena(self):
        """A fresh gradient buffer (the previous one stays alive through whoever still holds views of it: parameter .grad tensors that were
        not released before this backward)."""
        self.g32 = torch.zeros_like(self.g32)

    def grads(self) -> Dict[str, torch.Tensor]:
        return {n: self.grad(n) for n in self.names}

    # ---- fused clip + Adam (medmoe_amd.swin_engine): the kernels of ParamStore / TextStore on this arena --------------------------------
    def sumsq(self) -> torch.Tensor:
        """Sum of squares of the gradient arena, summed in a fixed order (identical on every rank): this arena's share of the clip norm."""
        if getattr(self, "normsq", None) is None:
            self.normsq = torch.zeros(1, device=self.device, dtype=torch.float32)
            self.norm_scratch = torch.zeros(2049, device=self.device, dtype=torch.float32)
        ops.call("sumsq_det", self.g32, self.numel, self.normsq, self.norm_scratch)
        return self.normsq

    def adam_step(self, normsq_total: torch.Tensor, lr: float, weight_decay: float, clip: float, grad_scale: float = 1.0):
        """clip (against `normsq_total`, the squared norm over ALL arenas of the model, as clip_grad_norm_ over all parameters computes it)
        + torch.optim.Adam's update (betas 0.9 / 0.999, eps 1e-8, L2 weight decay) on the fp32 master, the bf16 copy written by the same
        kernel; the transposed copies follow.  Gradients must be in THIS arena's g32 (no new_grad_arena() since the backward)."""
        if getattr(self, "m", None) is None:
            self.m, self.v, self.step_count = torch.zeros_like(self.p32), torch.zeros_like(self.p32), 0
        self.step_count += 1
        ops.call("adam_step", self.p32, self.g32, self.m, self.v, self.p16, self.numel, lr, 0.9, 0.999, 1e-8, weight_decay, self.step_count,
                 normsq_total, clip, grad_scale)
        if self.tr_table is not None:
            ops.call("transpose_many", self.p16, self.p16t, self.tr_table, self.tr_table.shape[0], self.tr_max_tiles)
