"""CPU: the oracle (oracle/medmoe_oracle.py) against golden vectors captured from the
reference's own modules by oracle/gen_golden.py.  Tolerance fp32 rtol 1e-5 / atol 1e-6
(a few cases accumulate over hundreds of terms and use 2e-5/2e-6); indices exact."""
import os

import numpy as np
import pytest
import torch

import medmoe_oracle as O

RT, AT = 1e-5, 1e-6


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def close(a, b, rtol=RT, atol=AT):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def enc_params(z, prefix="enc"):
    return {f"{prefix}.{k}": v for k, v in z.items() if k.startswith("layer.") or k.startswith("final_")}


def test_mhsa(golden_dir):
    z = load(golden_dir, "mhsa")
    args = (z["input_proj.weight"], z["input_proj.bias"], z["output_proj.weight"], z["output_proj.bias"], 4)
    close(O.mhsa(z["x"], *args), z["y_nomask"])
    close(O.mhsa(z["x"], *args, key_mask=z["key_mask"]), z["y_mask"])


@pytest.mark.parametrize("name,norm_first,eps,final", [("enc_prenorm", True, 1e-6, True),
                                                       ("enc_postnorm", False, 1e-12, False)])
def test_encoder_fwd_bwd(golden_dir, name, norm_first, eps, final):
    z = load(golden_dir, name)
    p = {k: v.clone().requires_grad_(True) for k, v in enc_params(z).items()}
    x = z["x"].clone().requires_grad_(True)
    km = None if norm_first else z["key_mask"]
    last, hs = O.encoder(x, p, "enc", 2, 4, eps, norm_first, km, final)
    close(last, z["last"], 2e-5, 2e-6)
    for i, h in enumerate(hs):
        close(h, z[f"hs{i}"], 2e-5, 2e-6)
    (last * z["gy"]).sum().backward()
    close(x.grad, z["gx"], 1e-4, 1e-5)
    for k, v in p.items():
        close(v.grad, z["grad." + k[len("enc."):]], 1e-4, 1e-5)


def test_router_probs_and_argmax(golden_dir):
    z = load(golden_dir, "router")
    for pre in ("", "tie."):
        p = {"moe.router.0.weight": z[pre + "router.0.weight"], "moe.router.0.bias": z[pre + "router.0.bias"],
             "moe.router.2.weight": z[pre + "router.2.weight"], "moe.router.2.bias": z[pre + "router.2.bias"]}
        pr = O.router_probs(z["x"], p)
        close(pr, z[pre + "probs"])
        assert torch.equal(O.topk_lowest_index(pr, 1)[:, 0], z[pre + "top1"])
        # the fixed-order numpy restatement picks the same experts (bit-exact index contract)
        _, idx, _, _ = O.router_fixed_order(z["x"].numpy(), p["moe.router.0.weight"].numpy(),
                                         p["moe.router.0.bias"].numpy(), p["moe.router.2.weight"].numpy(),
                                         p["moe.router.2.bias"].numpy(), 1)
        assert np.array_equal(idx[:, 0], z[pre + "top1"].numpy())
    # the constructed tie must resolve to the LOWEST index (1, never 3)
    assert not (z["tie.top1"] == 3).any()


def expert_params(z, e=0):
    return {f"moe.experts.{e}.{k}": v for k, v in z.items() if k.startswith("proj_convs") or k.startswith("attn_proj")}


def test_expert_vit_fwd_bwd(golden_dir):
    z = load(golden_dir, "expert_vit")
    p = {k: v.clone().requires_grad_(True) for k, v in expert_params(z).items()}
    feats = [z[f"f{s}"].clone().requires_grad_(True) for s in range(4)]
    y = O.expert_forward(feats, p, 0)
    close(y, z["y"])
    (y * z["gy"]).sum().backward()
    for s in range(4):
        close(feats[s].grad, z[f"gf{s}"], 1e-4, 1e-6)
    for k, v in p.items():
        close(v.grad, z["grad." + k[len("moe.experts.0."):]], 1e-4, 1e-5)


def test_expert_pyramid(golden_dir):
    z = load(golden_dir, "expert_pyramid")
    y = O.expert_forward([z[f"f{s}"] for s in range(4)], expert_params(z), 0)
    close(y, z["y"])


def test_moe_fwd_bwd(golden_dir):
    z = load(golden_dir, "moe")
    p = {"moe." + k: v.clone().requires_grad_(True) for k, v in z.items()
         if k.startswith("experts.") or k.startswith("router.")}
    feats = [z[f"f{s}"].clone().requires_grad_(True) for s in range(4)]
    rin = z["rin"].clone().requires_grad_(True)
    g, l, pr, idx = O.moe_forward(feats, rin, p, 4, 1)
    assert torch.equal(idx[:, 0], z["top1"])
    assert len(set(z["top1"].tolist())) > 1, "fixture should route to several experts"
    close(g, z["global"]); close(l, z["local"]); close(pr, z["probs"])
    ((g * z["gg"]).sum() + (l * z["gl"]).sum() + (pr * z["gp"]).sum()).backward()
    close(rin.grad, z["g_rin"], 1e-4, 1e-6)
    for s in range(4):
        close(feats[s].grad, z[f"gf{s}"], 1e-4, 1e-6)
    for k, v in p.items():
        ref = z["grad." + k[len("moe."):]]
        got = v.grad if v.grad is not None else torch.zeros_like(v)   # unselected experts: 0
        close(got, ref, 1e-4, 1e-5)


def test_gloria_global(golden_dir):
    z = load(golden_dir, "gloria_global")
    a = z["img"].clone().requires_grad_(True)
    t = z["txt"].clone().requires_grad_(True)
    lo = O.gloria_global(a, t, 10.0)
    close(lo, z["loss"])
    lo.backward()
    close(a.grad, z["g_img"], 1e-4, 1e-6); close(t.grad, z["g_txt"], 1e-4, 1e-6)


def test_gloria_local(golden_dir):
    z = load(golden_dir, "gloria_local")
    il = z["img_l"].clone().requires_grad_(True)
    wl = z["words"].clone().requires_grad_(True)
    cl = z["cap_lens"].tolist()
    l0, l1, maps = O.gloria_local(il, wl, cl, 4.0, 5.0, 10.0)
    close(l0, z["loss0"]); close(l1, z["loss1"])
    for i, m in enumerate(maps):
        close(m, z[f"att{i}"])
    (l0 + l1).backward()
    close(il.grad, z["g_img_l"], 1e-4, 1e-6); close(wl.grad, z["g_words"], 1e-4, 1e-6)


def test_soft_gloria(golden_dir):
    """oracle Soft-GLoRIA (global + local) against the reference's classes (fixture from oracle/gen_golden_soft.py): losses, attention
    maps and every input gradient; rows hold 2..7 positives and 1..5 negatives."""
    z = load(golden_dir, "soft_gloria")
    thr = [float(v) for v in z["thresholds"]]
    a = z["a"].clone().requires_grad_(True); t = z["t"].clone().requires_grad_(True)
    g = O.soft_gloria_global(a, t, z["soft"], thr, 10.0)
    close(g, z["g_loss"])
    g.backward()
    close(a.grad, z["grad_a"], 1e-4, 1e-6); close(t.grad, z["grad_t"], 1e-4, 1e-6)
    il = z["img_l"].clone().requires_grad_(True); wl = z["words"].clone().requires_grad_(True)
    l0, l1, maps = O.soft_gloria_local(il, wl, z["cap_lens"].tolist(), z["soft"], thr, 4.0, 5.0, 10.0)
    close(l0, z["loss0"]); close(l1, z["loss1"])
    for i, m in enumerate(maps):
        close(m, z[f"att{i}"])
    (l0 + 2.0 * l1).backward()
    close(il.grad, z["grad_img_l"], 1e-4, 1e-6); close(wl.grad, z["grad_words"], 1e-4, 1e-6)


def test_hard_negative(golden_dir):
    """oracle HardNegativeContrastiveLoss against the reference's class (oracle/gen_golden_soft.py): 7 of 8 rows and 6 of 8 columns active."""
    z = load(golden_dir, "hard_negative")
    a = z["imgs"].clone().requires_grad_(True); t = z["caps"].clone().requires_grad_(True)
    l = O.hard_negative(a, t)
    close(l, z["loss"])
    l.backward()
    close(a.grad, z["grad_imgs"], 1e-5, 1e-7); close(t.grad, z["grad_caps"], 1e-5, 1e-7)
    close(O.hard_negative(z["imgs"], z["caps"], margin=0.9), z["loss_margin09"])


def test_contrastive_with_temperature(golden_dir):
    z = load(golden_dir, "contrastive_temp")
    loss, la, lb, loss_a, loss_b = O.contrastive_with_temperature(
        z["a"], z["b"], z["a"], z["b"], z["logit_scale"], rank=0)
    close(loss, z["loss"]); close(la, z["logits_a"]); close(lb, z["logits_b"])
    close(loss_a, z["loss_a"]); close(loss_b, z["loss_b"])


def test_bert_aggregate(golden_dir):
    z = load(golden_dir, "bert_aggregate")
    vocab = O.Vocab(z["is_cont"].numpy().astype(bool), z["starts_bracket"].numpy().astype(bool))
    seg, n_words, cap = O.segment_map(z["ids"].numpy(), vocab)
    assert cap.tolist() == z["cap_lens"].tolist()
    word, sent = O.aggregate_last_layers([z[f"h{i}"] for i in range(4)], seg, 4)
    close(word, z["word"]); close(sent, z["sent"])
    sents = [l.split() for l in open(os.path.join(golden_dir, "bert_aggregate_sents.txt"))]
    for b, s in enumerate(sents):           # merged-word count == our n_words
        assert len([w for w in s if w != "[PAD]"]) == n_words[b]


def test_composite_loss(golden_dir):
    z = load(golden_dir, "composite")
    p = {"moe." + k: v for k, v in z.items() if k.startswith("experts.") or k.startswith("router.")}
    g, l, pr, _ = O.moe_forward([z[f"f{s}"] for s in range(4)], z["rin"], p, 3, 1)
    cl = z["cap_lens"].tolist()
    l0, l1, _ = O.gloria_local(l, z["txt_l"], cl, 4.0, 5.0, 10.0)
    g_loss = O.gloria_global(g, z["txt_g"], 10.0)
    c_loss = O.router_ce(pr, z["label"])
    close(l0 + l1, z["l_loss"]); close(g_loss, z["g_loss"]); close(c_loss, z["c_loss"])
    close(0.5 * (l0 + l1) + 0.5 * g_loss + 2.0 * c_loss, z["loss"])


def test_model_step_runs_tiny():
    cfg = O.config_by_name("tiny")
    p = O.init_params(cfg, seed=0)
    batch = O.synthetic_batch(cfg, 4, min_len=4)
    out = O.model_step(batch, p, cfg, O.Vocab.synthetic(cfg.vocab))
    assert torch.isfinite(out["loss"])
    assert out["img_l"].shape == (4, cfg.d_out, 8, 8) and out["txt_l"].shape == (4, cfg.d_t, cfg.max_len)
