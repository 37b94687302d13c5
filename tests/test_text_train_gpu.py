"""Trainable text tower (reference `freeze_bert: false`: text_encoder.py:27-30 leaves the BERT parameters trainable; the pretraining experiment
freezes them, configs/model/med-moe.yaml:35): `MedMoEConfig.freeze_text = False` - padded text pass with saved activations, word gradients of
the local loss, caption-side gradient of the global loss, text backward (aggregation, post-norm blocks, embedding front-end), one clip norm
over both towers, fused Adam on a second flat store.  Against the CPU oracle's autograd with the text parameters requiring gradients."""
import numpy as np
import pytest
import torch

import medmoe_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def bf_round(t):
    return t.to(torch.bfloat16).float()


def make(cfg_name, B, seed=0, n_continuation=0):
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine, VocabTables
    ocfg, cfg = O.config_by_name(cfg_name), config_by_name(cfg_name)
    ocfg.freeze_text = cfg.freeze_text = False
    p = O.init_params(ocfg, seed=seed, std=0.05)
    g = torch.Generator().manual_seed(seed + 7)
    for k in p:
        if k.endswith("layernorm.weight") or k.endswith("layer_norm.weight"):
            p[k] = 1 + 0.2 * torch.randn(p[k].shape, generator=g)
        elif k.endswith(".bias"):
            p[k] = 0.05 * torch.randn(p[k].shape, generator=g)
    p["moe.router.0.weight"] *= 8.0; p["moe.router.2.weight"] *= 8.0
    for k in p:      # GEMM weights the engine keeps in bf16 (both towers now) are rounded for the oracle too
        if k.endswith(".weight") and p[k].dim() >= 2 and not k.startswith("moe.router") and "embeddings" not in k:
            p[k] = bf_round(p[k])
    batch = O.synthetic_batch(ocfg, B, min_len=4)
    batch["image"] = bf_round(batch["image"])
    if n_continuation:
        gi = torch.Generator().manual_seed(seed + 5)
        ids = batch["ids"]
        cont = torch.randint(ocfg.vocab - n_continuation, ocfg.vocab, ids.shape, generator=gi)
        pick = (torch.rand(ids.shape, generator=gi) < 0.35) & (ids > 2)
        pick[:, :2] = False
        batch["ids"] = torch.where(pick, cont, ids)
    eng = Engine(cfg, "cuda:0", vocab=VocabTables.synthetic(cfg.vocab, "cuda:0", n_continuation))
    eng.params.load_named({k: v for k, v in p.items() if not k.startswith("text.")})
    eng.tstore.load_named(p)
    return ocfg, cfg, p, batch, eng, O.Vocab.synthetic(ocfg.vocab, n_continuation)


def test_text_tower_gradients_against_the_oracle():
    """tiny2 (64 regions -> the transposed local loss with word gradients), 8 captions with '##' continuation pieces (several tokens feed one
    word).  (1) forward: the padded training pass gives the embeddings / losses of the frozen pass and of the oracle; (2) the loss kernels'
    caption-side gradients (d words, d txt_g) against the oracle's losses differentiated at the ENGINE's own features; (3) the text backward
    alone: those gradients pushed through the oracle's text graph - every text parameter; (4) the whole fp32 chain (well conditioned at this
    size).  Image-side gradients are unchanged by the mode."""
    B = 8
    ocfg, cfg, p, batch, eng, vocab = make("tiny2", B, seed=3, n_continuation=12)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = O.model_step(batch, pr, ocfg, vocab)
    ref["loss"].backward()
    dev_b = {k: v.cuda() for k, v in batch.items()}
    out = eng.train_step(dev_b, optimizer=False)
    torch.cuda.synchronize()
    o = eng.outputs()
    assert np.array_equal(o["cap_lens"].cpu().numpy(), np.asarray(ref["cap_lens"]))
    assert torch.equal(o["idx"].cpu().long(), ref["idx"])
    assert rel(o["txt_g"], ref["txt_g"]) < 2e-2 and rel(o["txt_l"], ref["txt_l"]) < 2e-2
    for k in ("g_loss", "l_loss"):
        assert abs(out[k].item() - ref[k].item()) < 1e-2 * abs(ref[k].item()), k
    # (2) caption-side gradients of the two losses at the engine's own features
    P, Do, Hh, T = cfg.n_patch, cfg.d_out, int(cfg.n_patch ** 0.5), cfg.max_len
    x = eng.ws["img_l"].float().cpu().transpose(1, 2).reshape(B, Do, Hh, Hh)
    w = eng.ws["words"].float().cpu().transpose(1, 2).clone().requires_grad_(True)            # [B, D, T]
    tg = eng.ws["txt_g"].float().cpu().clone().requires_grad_(True)
    l0, l1, _ = O.gloria_local(x, w, ref["cap_lens"], ocfg.temp1, ocfg.temp2, ocfg.temp3)
    (ocfg.w_local * (l0 + l1) + ocfg.w_global * O.gloria_global(eng.ws["img_g"].float().cpu(), tg, ocfg.temp3)).backward()
    e_w = rel(eng._d_words.transpose(1, 2), w.grad)
    e_g = rel(eng.ws["d_txt_g"], tg.grad)
    print(f"caption-side loss gradients: d words {e_w:.4f}  d txt_g {e_g:.5f}")
    assert e_w < 2e-2 and e_g < 1e-3
    # (3) the text backward alone: the engine's own gradients at the tower's outputs through the oracle's text graph
    got = eng.tstore.export_named(eng.tstore.g32)
    for v in pr.values():
        v.grad = None
    word_o, sent_o, _ = O.text_tower(batch["ids"], batch["attn_mask"], batch["token_type"], pr, ocfg, vocab)
    ((word_o * eng._d_words.cpu().transpose(1, 2)).sum() + (sent_o * eng.ws["d_txt_g"].cpu()).sum()).backward()
    errs = {}
    for k, v in pr.items():
        if k.startswith("text.") and v.grad is not None and float(v.grad.norm()) > 1e-9:
            errs[k] = rel(got[k].reshape(v.grad.shape), v.grad)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("text backward worst", [(k, round(e, 4)) for k, e in worst], "median", float(np.median(list(errs.values()))), "n", len(errs))
    assert len(errs) == 5 + 12 * ocfg.n_layer_t
    assert max(errs.values()) < 6e-2 and float(np.median(list(errs.values()))) < 2e-2, worst
    # rows of the word table nobody used stay exactly zero
    used = torch.zeros(ocfg.vocab, dtype=torch.bool); used[batch["ids"].reshape(-1)] = True
    assert float(got["text.word_embeddings"][~used].abs().max()) == 0.0


def test_text_training_steps_move_both_towers_and_track_torch_adam():
    """Three optimiser steps with the text tower trainable: the loss falls, text parameters move the way torch's clip_grad_norm_ (ONE norm over
    image + text parameters) + Adam move the oracle's (direction cosine > 0.9 after the sign-like first steps), the bf16 working copies of the
    text weights follow their master, and a frozen engine on the same batch leaves its text tower untouched."""
    ocfg, cfg, p, batch, eng, vocab = make("tiny", 8, seed=2)
    eng.cfg.lr = 1e-3
    b = {k: v.cuda() for k, v in batch.items()}
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    train = list(po.values())
    opt = torch.optim.Adam(train, lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        O.model_step(batch, po, ocfg, vocab)["loss"].backward()
        torch.nn.utils.clip_grad_norm_(train, cfg.clip)
        opt.step()
    losses = [float(eng.train_step(b)["loss"]) for _ in range(3)]
    torch.cuda.synchronize()
    named = eng.tstore.export_named()
    for k in ("text.layer.0.feedforward.model.0.weight", "text.layer.1.attention.input_proj.weight", "text.position_embeddings", "text.emb_layernorm.weight"):
        dv, do = named[k].reshape(p[k].shape) - p[k], po[k].detach() - p[k]
        cos = float((dv * do).sum() / (dv.norm() * do.norm() + 1e-30))
        assert float(dv.norm()) > 0 and cos > 0.9, (k, cos)
    for _ in range(20):
        losses.append(float(eng.train_step(b)["loss"]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.1, losses[::4]
    ts = eng.tstore
    w = "layer.0.attention.input_proj.weight"
    assert torch.equal(ts.w16(w), ts.f32(w).to(torch.bfloat16)) and torch.equal(ts.w16t(w), ts.w16(w).t())
    assert eng.params.text[w].data_ptr() == ts.w16(w).data_ptr()                 # the forward pass reads the store's own working copy
    # frozen (default) engine: no text store, text weights untouched by steps
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    fz = Engine(config_by_name("tiny"), "cuda:0")
    before = {k: v.clone() for k, v in fz.params.text.items()}
    fz.train_step(b)
    assert fz.tstore is None and all(torch.equal(before[k], v) for k, v in fz.params.text.items())
