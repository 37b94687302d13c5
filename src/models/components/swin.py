"""Mirror of reference src/models/components/swin.py:119-149 (`SWIN`): the Swin-T image encoder with the modality MoE on its four stages,
`forward(x) -> (global_feat [B, 768], local_feat [B, 768, 56, 56], router_logits [B, E])` (router_logits are the softmaxed probabilities, as
the reference returns them, swin.py:99), running on the HIP kernels (`medmoe_amd.swin_moe.SwinMoEEncoder`) behind torch autograd: the
parameters are ordinary `nn.Parameter`s, any torch optimizer trains them; `state_dict()` / `load_state_dict()` use the reference's names
(`model.*` = HF SwinModel, `moe.*`).

Differences forced by the environment: `x` is the already normalised [B, 3, 224, 224] tensor (the reference passes PIL images through
`AutoImageProcessor`; the device-side preprocessing lives in medmoe_amd.data), and `pretrained=True` cannot fetch
'microsoft/swin-tiny-patch4-window7-224' offline - pass `state_dict=` (a SwinModel / reference checkpoint) or get the random init of the
published geometry.  `lora=True` and `use_moe=False` are not built (the pretraining_medmoe experiment uses neither)."""
from typing import Dict, Optional

import torch
from torch import nn

from medmoe_amd.swin_moe import SwinMoEEncoder


def _random_init(num_experts: int, seed: int) -> Dict[str, torch.Tensor]:
    from transformers import SwinConfig, SwinModel
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    w = {"model." + k: v for k, v in SwinModel(SwinConfig()).state_dict().items() if v.dtype.is_floating_point}
    lin = lambda o, i: (torch.randn(o, i, generator=g) * i ** -0.5, torch.zeros(o))
    w["moe.router.0.weight"], w["moe.router.0.bias"] = lin(128, 768)                       # swin.py:88-92
    w["moe.router.2.weight"], w["moe.router.2.bias"] = lin(num_experts, 128)
    for e in range(num_experts):
        for s, d in enumerate((96, 192, 384, 768)):                                       # swin.py:14-21: Conv1d(d, 768, 1) + ReLU per stage
            ww, bb = lin(768, d)
            w[f"moe.experts.{e}.proj_convs.{s}.0.weight"], w[f"moe.experts.{e}.proj_convs.{s}.0.bias"] = ww.unsqueeze(-1), bb
        w[f"moe.experts.{e}.attn_proj.0.weight"], w[f"moe.experts.{e}.attn_proj.0.bias"] = lin(384, 768)
        w[f"moe.experts.{e}.attn_proj.2.weight"], w[f"moe.experts.{e}.attn_proj.2.bias"] = lin(1, 384)
    return w


class _SwinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        enc = module._encoder()
        rate = module.drop_path_rate                                  # stochastic depth as SwinConfig.drop_path_rate trains it
        masks = enc.tower.sample_drop_path(x.shape[0], rate) if (module.training and rate > 0.0) else None
        out = enc.forward(x.detach().to(torch.bfloat16).contiguous(), drop_path=masks, drop_path_rate=rate)
        module._generation += 1
        ctx.module, ctx.generation = module, module._generation
        B, P, D = out["local_feat"].shape
        side = int(P ** 0.5)
        return out["global_feat"], out["local_feat"].transpose(1, 2).reshape(B, D, side, side), out["router_probs"].clone()

    @staticmethod
    def backward(ctx, d_global, d_local, d_probs):
        module = ctx.module
        if ctx.generation != module._generation:
            raise RuntimeError("SWIN: backward of an earlier forward - the encoder keeps ONE forward's activations (run backward before the "
                               "next forward of the module, or run the other pass under torch.no_grad() on a second instance)")
        enc = module._enc
        if any(p.grad is not None for p in module.params):           # gradients of an earlier backward still in use (accumulation, or
            enc.new_grad_arenas()                                    # zero_grad(set_to_none=False)): they may be views of the arena - keep it
        B, D = d_local.shape[0], d_local.shape[1]
        grads = enc.backward(d_global, d_local.reshape(B, D, -1).transpose(1, 2).to(torch.bfloat16).contiguous(), d_probs=d_probs)
        return (None, None) + tuple(grads[n].reshape(p.shape).to(p.dtype) for n, p in zip(module._names, module.params))


class SWIN(nn.Module):
    def __init__(self, pretrained: bool = True, lora: bool = False, lora_r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.1,
                 use_moe: bool = True, num_experts: int = 6, state_dict: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0):
        super().__init__()
        if lora or not use_moe:
            raise NotImplementedError("SWIN (HIP): lora=True / use_moe=False are outside the pretraining_medmoe path")
        w = _random_init(num_experts, seed)
        if state_dict is not None:
            missing = [k for k in w if k not in state_dict]
            if missing:
                raise KeyError(f"SWIN: state_dict lacks {missing[:3]} ... ({len(missing)} names; expected `model.*` SwinModel and `moe.*` keys)")
            w = {k: state_dict[k].detach().float() for k in w}
        self.num_experts = num_experts
        self.drop_path_rate = 0.1                                     # SwinConfig.drop_path_rate of 'microsoft/swin-tiny-patch4-window7-224'
        self._names = sorted(w)
        self.params = nn.ParameterList([nn.Parameter(w[n].clone()) for n in self._names])
        self._enc: Optional[SwinMoEEncoder] = None
        self._seen = None
        self._views = []
        self._plist = None
        self._generation = 0                                          # forward passes so far: a backward must belong to the latest one
        # state_dict under the reference's names (`model.*` = HF SwinModel, `moe.*`; swin.py:119-128) instead of `params.<i>`
        self._register_state_dict_hook(SWIN._named_keys)
        self._register_load_state_dict_pre_hook(self._indexed_keys)

    @staticmethod
    def _named_keys(module, state_dict, prefix, local_metadata):
        for i, n in enumerate(module._names):
            if prefix + f"params.{i}" in state_dict:
                state_dict[prefix + n] = state_dict.pop(prefix + f"params.{i}")
        return state_dict

    def _indexed_keys(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for i, n in enumerate(self._names):
            if prefix + n in state_dict:
                state_dict[prefix + f"params.{i}"] = state_dict.pop(prefix + n)

    def named_weights(self) -> Dict[str, torch.Tensor]:
        return {n: p for n, p in zip(self._names, self.params)}

    def _encoder(self) -> SwinMoEEncoder:
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("SWIN (HIP): move the module to the GPU first; there is no CPU path")
        views = self._views
        plist = self._plist = self._plist if self._plist is not None else list(self.params)        # (ParameterList indexing is slow)
        aliased = self._enc is not None and self._enc.dev == dev and all(p.data_ptr() == v.data_ptr() for p, v in zip(plist, views))
        if not aliased:                                               # first use on this device (or the parameters were replaced): build the
            self._enc = SwinMoEEncoder({n: p.data for n, p in zip(self._names, self.params)}, self.num_experts, dev)     # arenas from them and
            pv = self._enc.parameter_views()                          # make the nn.Parameters views of the fp32 arena: an optimizer step
            self._views = views = [pv[n] for n in self._names]        # updates it in place, no copy back
            for p, v in zip(plist, views):
                p.data = v.view(p.shape)
            self._seen = None
        stamp = tuple(p._version for p in plist)
        if self._seen is not None and stamp != self._seen:            # an optimizer step / load_state_dict: new bf16 working copies
            self._enc.refresh()
        self._seen = stamp
        return self._enc

    def forward(self, x: torch.Tensor):
        return _SwinFn.apply(self, x, *self.params)
