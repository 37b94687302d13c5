"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REFERENCE's own
importable modules (see oracle/_ref_import.py) on seeded fp32 CPU inputs.
Run in the build container only:  python oracle/gen_golden.py
The fixtures hold data (inputs, weights, expected outputs/grads) - no reference source.
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
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
            for k, v in d.items()}


def sd(mod, prefix=""):
    return {prefix + k: v.detach().clone() for k, v in mod.state_dict().items()}


def main():
    R = _ref_import.load()
    torch.manual_seed(1234)
    os.makedirs(OUT, exist_ok=True)
    nn = torch.nn

    # (1) MultiHeadSelfAttention D=64,H=4,N=9 with / without bool key-padding mask
    m = R.mha.MultiHeadSelfAttention(64, 4)
    x = torch.randn(3, 9, 64)
    km = torch.ones(3, 9, dtype=torch.bool)
    km[0, 6:] = False
    km[2, 3:] = False
    np.savez(os.path.join(OUT, "mhsa.npz"), **npd({
        **sd(m), "x": x, "key_mask": km, "y_nomask": m(x),
        "y_mask": m(x, attn_mask=km[:, None, None, :])}))

    # (2)+(3) TransformerEncoder pre-norm (ViT-style, final LN) and post-norm (text-style)
    for name, nf, eps, fin in (("enc_prenorm", True, 1e-6, 1e-6), ("enc_postnorm", False, 1e-12, None)):
        enc = R.transformer.TransformerEncoder(2, 64, 4, 128, 0.0, nn.GELU, eps, nf, fin)
        for mod in enc.modules():
            if isinstance(mod, nn.LayerNorm):
                mod.weight.data.uniform_(0.5, 1.5)
                mod.bias.data.uniform_(-0.2, 0.2)
        x = torch.randn(3, 9, 64, requires_grad=True)
        mask = None if nf else km[:, None, None, :]
        out = enc(x, attention_mask=mask, return_hidden_states=True)
        gy = torch.randn_like(out.last_hidden_state)
        (out.last_hidden_state * gy).sum().backward()
        d = {**sd(enc), "x": x, "key_mask": km, "gy": gy, "last": out.last_hidden_state, "gx": x.grad}
        for i, h in enumerate(out.hidden_states):
            d[f"hs{i}"] = h
        for k, v in enc.named_parameters():
            d["grad." + k] = v.grad
        np.savez(os.path.join(OUT, name + ".npz"), **npd(d))

    # (4) MoE router probs + argmax incl. a constructed exact tie
    moe = R.swin.MoE(num_experts=4, hidden_dims=[32] * 4, output_dim=48, router_input_dim=32)
    xr = torch.randn(16, 32)
    lg = torch.softmax(moe.router(xr), dim=-1)
    moe_t = R.swin.MoE(num_experts=4, hidden_dims=[32] * 4, output_dim=48, router_input_dim=32)
    moe_t.router[2].weight.data[1] = moe_t.router[2].weight.data[3]
    moe_t.router[2].bias.data[1] = moe_t.router[2].bias.data[3]
    lg_t = torch.softmax(moe_t.router(xr), dim=-1)
    np.savez(os.path.join(OUT, "router.npz"), **npd({
        **sd(moe.router, "router."), "x": xr, "probs": lg, "top1": torch.argmax(lg, -1),
        **sd(moe_t.router, "tie.router."), "tie.probs": lg_t, "tie.top1": torch.argmax(lg_t, -1)}))

    # (5) Expert: ViT geometry (equal P) and pyramid geometry (exercises interpolate)
    ex = R.swin.Expert([32] * 4, 48)
    feats = [torch.randn(3, 16, 32, requires_grad=True) for _ in range(4)]
    y = ex(feats)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    d = {**sd(ex), "y": y, "gy": gy}
    for s, f in enumerate(feats):
        d[f"f{s}"] = f
        d[f"gf{s}"] = f.grad
    for k, v in ex.named_parameters():
        d["grad." + k] = v.grad
    np.savez(os.path.join(OUT, "expert_vit.npz"), **npd(d))
    exp_ = R.swin.Expert([8, 16, 32, 64], 48)
    featsp = [torch.randn(2, 64, 8), torch.randn(2, 16, 16), torch.randn(2, 4, 32), torch.randn(2, 1, 64)]
    d = {**sd(exp_), "y": exp_(featsp)}
    for s, f in enumerate(featsp):
        d[f"f{s}"] = f
    np.savez(os.path.join(OUT, "expert_pyramid.npz"), **npd(d))

    # (6) MoE E=4 end to end, fwd + grads (dense-all-experts + gather in the reference)
    for b in moe.router:
        if isinstance(b, nn.Linear):
            b.weight.data.mul_(20.0)   # spread the routing so several experts are hit
    feats = [torch.randn(6, 16, 32, requires_grad=True) for _ in range(4)]
    rin = torch.randn(6, 32, requires_grad=True)
    g, l, pr = moe(feats, rin)
    gg, gl, gp = torch.randn_like(g), torch.randn_like(l), torch.randn_like(pr)
    ((g * gg).sum() + (l * gl).sum() + (pr * gp).sum()).backward()
    d = {**sd(moe), "rin": rin, "global": g, "local": l, "probs": pr, "top1": torch.argmax(pr, -1),
         "gg": gg, "gl": gl, "gp": gp, "g_rin": rin.grad}
    for s, f in enumerate(feats):
        d[f"f{s}"] = f
        d[f"gf{s}"] = f.grad
    for k, v in moe.named_parameters():
        d["grad." + k] = v.grad if v.grad is not None else torch.zeros_like(v)
    np.savez(os.path.join(OUT, "moe.npz"), **npd(d))

    # (7) GLORIA global B=5
    a = torch.randn(5, 48, requires_grad=True)
    t = torch.randn(5, 48, requires_grad=True)
    lo = R.losses.GLORIAGlobalContrastiveLoss()(a, t, temp3=10.0)
    lo.backward()
    np.savez(os.path.join(OUT, "gloria_global.npz"), **npd(
        {"img": a, "txt": t, "loss": lo, "g_img": a.grad, "g_txt": t.grad}))

    # (8) GLORIA local B=4, HW=3x3, ragged cap_lens, incl. att_maps
    il = torch.randn(4, 24, 3, 3, requires_grad=True)
    wl = torch.randn(4, 24, 7, requires_grad=True)
    cl = [7, 3, 5, 2]
    o = R.losses.GLORIALocalContrastiveLoss()(il, wl, cl, temp1=4.0, temp2=5.0, temp3=10.0)
    (o.loss0 + o.loss1).backward()
    d = {"img_l": il, "words": wl, "cap_lens": np.array(cl), "loss0": o.loss0, "loss1": o.loss1,
         "g_img_l": il.grad, "g_words": wl.grad}
    for i, mp in enumerate(o.att_maps):
        d[f"att{i}"] = mp
    np.savez(os.path.join(OUT, "gloria_local.npz"), **npd(d))

    # (9) contrastive_loss_with_temperature, non-distributed
    a = torch.randn(6, 16)
    b = torch.randn(6, 16)
    ls = torch.nn.Parameter(torch.tensor(R.losses.DEFAULT_LOGIT_SCALE))
    o = R.losses.contrastive_loss_with_temperature(a, b, ls)
    np.savez(os.path.join(OUT, "contrastive_temp.npz"), **npd(
        {"a": a, "b": b, "logit_scale": ls, "loss": o.loss, "logits_a": o.logits_a,
         "logits_b": o.logits_b, "loss_a": o.loss_a, "loss_b": o.loss_b}))

    # (10) BertEncoder.forward/aggregate_tokens with a synthetic vocab, __init__ bypassed
    from transformers import BertConfig, BertModel
    V = 40
    words = ["[PAD]", "[CLS]", "[SEP]"] + [f"w{i}" for i in range(3, 30)] + [f"##p{i}" for i in range(30, V)]
    be = R.text_encoder.BertEncoder.__new__(R.text_encoder.BertEncoder)
    nn.Module.__init__(be)
    be.model = BertModel(BertConfig(vocab_size=V, hidden_size=32, num_hidden_layers=4,
                                    num_attention_heads=2, intermediate_size=64,
                                    max_position_embeddings=16, hidden_dropout_prob=0.0,
                                    attention_probs_dropout_prob=0.0,
                                    output_hidden_states=True, return_dict=False)).eval()
    be.idxtoword = dict(enumerate(words))
    be.last_n_layers, be.aggregate_method, be.norm = 4, "sum", False
    be.embed_dim, be.agg_tokens, be.emb_global, be.emb_local = 32, True, None, None
    ids = torch.tensor([[1, 5, 31, 32, 7, 2, 0, 0, 0, 0],
                        [1, 9, 10, 33, 11, 12, 34, 35, 13, 2],
                        [1, 4, 2, 0, 0, 0, 0, 0, 0, 0]])
    am = (ids != 0).long()
    hs_ref = be.model(ids, am, torch.zeros_like(ids))[2]
    w, s, sents = be(ids, am, torch.zeros_like(ids))
    cap = [len([x for x in sent if not x.startswith("[")]) + 1 for sent in sents]
    d = {"ids": ids, "word": w, "sent": s, "cap_lens": np.array(cap),
         "is_cont": np.array([x.startswith("##") for x in words]),
         "starts_bracket": np.array([x.startswith("[") for x in words])}
    for i, h in enumerate(hs_ref[-4:]):
        d[f"h{i}"] = h
    np.savez(os.path.join(OUT, "bert_aggregate.npz"), **npd(d))
    with open(os.path.join(OUT, "bert_aggregate_sents.txt"), "w") as f:
        for sent in sents:
            f.write(" ".join(sent) + "\n")

    # (11) composite loss scalars: reference components wired as medmoe_module.py:284-316
    ref_moe = R.swin.MoE(num_experts=3, hidden_dims=[32] * 4, output_dim=24, router_input_dim=32)
    feats = [torch.randn(4, 9, 32) for _ in range(4)]
    rin = torch.randn(4, 32)
    g, l, pr = ref_moe(feats, rin)
    tl = torch.randn(4, 24, 7)
    tg = torch.randn(4, 24)
    lab = torch.tensor([0, 2, 1, 1])
    lo = R.losses.GLORIALocalContrastiveLoss()(l, tl, cl, temp1=4.0, temp2=5.0, temp3=10.0)
    l_loss = lo.loss0 + lo.loss1
    g_loss = R.losses.GLORIAGlobalContrastiveLoss()(g, tg, temp3=10.0)
    c_loss = torch.nn.functional.cross_entropy(pr, lab)
    total = 0.5 * l_loss + 0.5 * g_loss + 2.0 * c_loss
    d = {**sd(ref_moe), "rin": rin, "txt_l": tl, "txt_g": tg, "label": lab, "cap_lens": np.array(cl),
         "l_loss": l_loss, "g_loss": g_loss, "c_loss": c_loss, "loss": total}
    for s_, f in enumerate(feats):
        d[f"f{s_}"] = f
    np.savez(os.path.join(OUT, "composite.npz"), **npd(d))

    # (12) Expert, pyramid geometry at MFMA-friendly widths (token counts 64 / 16 / 4 / 1 as (5), channels 64 / 64 / 128 / 128,
    # output 128): forward + every gradient.  Appended LAST with its own seed so that fixtures (1)-(11) regenerate bit-equal.
    torch.manual_seed(4321)
    exm = R.swin.Expert([64, 64, 128, 128], 128)
    featsm = [torch.randn(3, 64, 64, requires_grad=True), torch.randn(3, 16, 64, requires_grad=True),
              torch.randn(3, 4, 128, requires_grad=True), torch.randn(3, 1, 128, requires_grad=True)]
    ym = exm(featsm)
    gym = torch.randn_like(ym)
    (ym * gym).sum().backward()
    d = {**sd(exm), "y": ym, "gy": gym}
    for s_, f in enumerate(featsm):
        d[f"f{s_}"] = f
        d[f"gf{s_}"] = f.grad
    for k, v in exm.named_parameters():
        d["grad." + k] = v.grad
    np.savez(os.path.join(OUT, "expert_pyramid_mfma.npz"), **npd(d))
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn}: {os.path.getsize(os.path.join(OUT, fn))} B")


if __name__ == "__main__":
    main()
