"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*_mfma.npz: the REFERENCE's own modules (oracle/_ref_import.py)
run on seeded fp32 CPU inputs at geometries the MFMA kernels accept (width 128, head_dim 64, 64 regions), so that the GPU tests can
feed the reference's numbers STRAIGHT to the HIP path (tests/test_ref_fixtures_gpu.py) instead of HIP -> oracle -> fixture.
The older D = 64 / 4-head / D = 24 fixtures (gen_golden.py) pin the oracle; these pin the kernels.
Run in the build container only:  python oracle/gen_golden_mfma.py
The fixtures hold data (inputs, weights, expected outputs / gradients) - no reference source.  Weights and inputs are rounded to
bf16 values (stored as fp32) so that the bf16 kernels see EXACTLY the operands the reference computed with.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def npd(d):
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def sd(mod, prefix=""):
    return {prefix + k: v.detach().clone() for k, v in mod.state_dict().items()}


def bf(t):
    return t.to(torch.bfloat16).float()


def round_matrices_(mod):
    """GEMM weights (dim >= 2) to bf16 values: what the engine's bf16 working copies hold."""
    with torch.no_grad():
        for p in mod.parameters():
            if p.dim() >= 2:
                p.copy_(bf(p))


def main():
    R = _ref_import.load()
    nn = torch.nn
    os.makedirs(OUT, exist_ok=True)

    # (1) MultiHeadSelfAttention(128, 2): head_dim 64, N = 24 tokens, with / without a bool key-padding mask; forward + every gradient
    torch.manual_seed(9001)
    m = R.mha.MultiHeadSelfAttention(128, 2)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)                    # default init gives near-uniform attention: sharpen it so the softmax matters
    round_matrices_(m)
    x = bf(torch.randn(3, 24, 128)).requires_grad_(True)
    km = torch.ones(3, 24, dtype=torch.bool)
    km[0, 17:] = False
    km[2, 5:] = False
    d = {**sd(m), "x": x, "key_mask": km}
    for tag, mask in (("nomask", None), ("mask", km[:, None, None, :])):
        for p in m.parameters():
            p.grad = None
        x.grad = None
        y = m(x, attn_mask=mask)
        gy = bf(torch.randn(3, 24, 128, generator=torch.Generator().manual_seed(77)))
        (y * gy).sum().backward()
        d[f"y_{tag}"] = y
        d[f"gy_{tag}"] = gy
        d[f"gx_{tag}"] = x.grad.clone()
        for k, v in m.named_parameters():
            d[f"grad_{tag}." + k] = v.grad.clone()
    np.savez(os.path.join(OUT, "mhsa_mfma.npz"), **npd(d))

    # (2) TransformerEncoder, 2 layers at D = 128 / 2 heads / ff 256: pre-norm (ViT style, final LN, 17 tokens = 4 x 4 patches + CLS) and
    # post-norm (text style, 16 tokens, key-padding mask); hidden states, last state, every gradient
    for name, nf, eps, fin, N, seed in (("enc_prenorm_mfma", True, 1e-6, 1e-6, 17, 9002), ("enc_postnorm_mfma", False, 1e-12, None, 16, 9003)):
        torch.manual_seed(seed)
        enc = R.transformer.TransformerEncoder(2, 128, 2, 256, 0.0, nn.GELU, eps, nf, fin)
        for mod in enc.modules():
            if isinstance(mod, nn.LayerNorm):
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.uniform_(-0.2, 0.2)
            elif isinstance(mod, nn.Linear):
                mod.weight.data.mul_(2.0)
                mod.bias.data.uniform_(-0.1, 0.1)
        round_matrices_(enc)
        x = bf(torch.randn(4, N, 128)).requires_grad_(True)
        kmk = torch.ones(4, N, dtype=torch.bool)
        if not nf:
            kmk[0, 11:] = False
            kmk[1, 3:] = False
            kmk[3, 15:] = False
        mask = None if nf else kmk[:, None, None, :]
        out = enc(x, attention_mask=mask, return_hidden_states=True)
        gy = bf(torch.randn_like(out.last_hidden_state))
        if not nf:
            gy = gy * kmk[:, :, None]                      # padding positions carry no gradient (they never reach a result)
        (out.last_hidden_state * gy).sum().backward()
        d = {**sd(enc), "x": x, "key_mask": kmk, "gy": gy, "last": out.last_hidden_state, "gx": x.grad}
        for i, h in enumerate(out.hidden_states):
            d[f"hs{i}"] = h
        for k, v in enc.named_parameters():
            d["grad." + k] = v.grad
        np.savez(os.path.join(OUT, name + ".npz"), **npd(d))

    # (3) router at router_input_dim 128: probabilities + argmax, and a constructed exact tie between experts 1 and 3
    torch.manual_seed(9034)
    moe = R.swin.MoE(num_experts=4, hidden_dims=[128] * 4, output_dim=128, router_input_dim=128)
    for b in moe.router:
        if isinstance(b, nn.Linear):
            b.weight.data.mul_(4.0)
    xr = torch.randn(32, 128)
    pr = torch.softmax(moe.router(xr), dim=-1)
    moe_t = R.swin.MoE(num_experts=4, hidden_dims=[128] * 4, output_dim=128, router_input_dim=128)
    moe_t.router[2].weight.data[1] = moe_t.router[2].weight.data[3]
    moe_t.router[2].bias.data[1] = moe_t.router[2].bias.data[3]
    # rows where the tied pair wins: push the other two logits down
    moe_t.router[2].bias.data[0] -= 5.0
    moe_t.router[2].bias.data[2] -= 5.0
    pr_t = torch.softmax(moe_t.router(xr), dim=-1)
    np.savez(os.path.join(OUT, "router_mfma.npz"), **npd({
        **sd(moe.router, "router."), "x": xr, "probs": pr, "top1": torch.argmax(pr, -1),
        **sd(moe_t.router, "tie.router."), "tie.probs": pr_t, "tie.top1": torch.argmax(pr_t, -1)}))

    # (4) MoE(num_experts=4, hidden_dims=[128]*4, output_dim=128) end to end, 16 tokens per scale, forward + every gradient
    # (dense-all-experts + gather in the reference; the router weights are scaled so that several experts are hit)
    round_matrices_(moe.experts)
    Bm = 8
    feats = [bf(torch.randn(Bm, 16, 128)).requires_grad_(True) for _ in range(4)]
    rin = torch.randn(Bm, 128, requires_grad=True)
    g, l, prm = moe(feats, rin)
    gen = torch.Generator().manual_seed(78)
    gg, gl, gp = torch.randn(g.shape, generator=gen), bf(torch.randn(l.shape, generator=gen)), torch.randn(prm.shape, generator=gen)
    ((g * gg).sum() + (l * gl).sum() + (prm * gp).sum()).backward()
    d = {**sd(moe), "rin": rin, "global": g, "local": l, "probs": prm, "top1": torch.argmax(prm, -1),
         "gg": gg, "gl": gl, "gp": gp, "g_rin": rin.grad}
    for s, f in enumerate(feats):
        d[f"f{s}"] = f
        d[f"gf{s}"] = f.grad
    for k, v in moe.named_parameters():
        d["grad." + k] = v.grad if v.grad is not None else torch.zeros_like(v)
    np.savez(os.path.join(OUT, "moe_mfma.npz"), **npd(d))

    # (5) GLORIALocalContrastiveLoss at D = 128, 8 x 8 = 64 regions, T = 16, B = 6, ragged cap_lens; losses, att_maps, both gradients
    torch.manual_seed(9005)
    il = bf(torch.randn(6, 128, 8, 8)).requires_grad_(True)
    wl = bf(torch.randn(6, 128, 16)).requires_grad_(True)
    cl = [16, 3, 9, 1, 12, 7]
    o = R.losses.GLORIALocalContrastiveLoss()(il, wl, cl, temp1=4.0, temp2=5.0, temp3=10.0)
    (o.loss0 + o.loss1).backward()
    d = {"img_l": il, "words": wl, "cap_lens": np.array(cl), "loss0": o.loss0, "loss1": o.loss1, "g_img_l": il.grad, "g_words": wl.grad}
    for i, mp in enumerate(o.att_maps):
        d[f"att{i}"] = mp
    np.savez(os.path.join(OUT, "gloria_local_mfma.npz"), **npd(d))

    # (6) GLORIAGlobalContrastiveLoss at B = 8, D = 128
    torch.manual_seed(9006)
    a = torch.randn(8, 128, requires_grad=True)
    t = torch.randn(8, 128, requires_grad=True)
    lo = R.losses.GLORIAGlobalContrastiveLoss()(a, t, temp3=10.0)
    lo.backward()
    np.savez(os.path.join(OUT, "gloria_global_mfma.npz"), **npd({"img": a, "txt": t, "loss": lo, "g_img": a.grad, "g_txt": t.grad}))

    for fn in sorted(os.listdir(OUT)):
        if "_mfma" in fn:
            print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))} B")


if __name__ == "__main__":
    main()
