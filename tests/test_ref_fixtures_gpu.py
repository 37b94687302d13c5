"""The REFERENCE's own numbers fed straight to the HIP path (HIP -> fixture, no oracle in between).

tests/golden/*_mfma.npz were produced by oracle/gen_golden_mfma.py from the reference's modules at geometries the MFMA kernels accept
(width 128, head_dim 64, 64 regions, router input 128); weights and inputs are bf16-representable, so the kernels see exactly the
operands the reference computed with and the bars below measure the kernels' bf16 arithmetic (fp32 accumulation, bf16 activations
between kernels) only.

  a1  MultiHeadSelfAttention(128, 2)            multi_head_attention.py:40-81       fwd + every gradient, with / without key mask
  a2  TransformerEncoder pre-norm  (ViT blocks)  transformer.py:98-114,158-256       hidden states, last state, every gradient
  a2  TransformerEncoder post-norm (text blocks) transformer.py:116-130              hidden states (padded and packed variable-length paths)
  a4  MoE router                                 swin.py:88-92,98-100                probabilities, arg-max bit-exact, constructed tie
  a5/a6  MoE(4 experts, [128]*4 -> 128)          swin.py:82-117                      fwd + every gradient, three active experts + an empty one
  a9  GLORIAGlobalContrastiveLoss                losses.py:766-794                   loss + both gradients
  a10 GLORIALocalContrastiveLoss                 losses.py:961-1026                  loss0, loss1, att_maps, region-feature AND word gradients
Bars are rel-L2 unless stated and are written at each assert."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def load(golden_dir, name):
    d = np.load(os.path.join(golden_dir, name))
    return {k: torch.from_numpy(np.asarray(d[k])) for k in d.files}


def dev16(t):
    return t.to("cuda", BF).contiguous()


# ---------------------------------------------------------------------------------------------------------------
# a1: MultiHeadSelfAttention
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_mhsa_reference_fixture(golden_dir, tag):
    from medmoe_amd import ops
    f = load(golden_dir, "mhsa_mfma.npz")
    B, N, D, H = 3, 24, 128, 2
    x = dev16(f["x"]).view(B * N, D)
    w_in, w_out = dev16(f["input_proj.weight"]), dev16(f["output_proj.weight"])
    b_in, b_out = f["input_proj.bias"].cuda(), f["output_proj.bias"].cuda()
    km = f["key_mask"].to(torch.uint8).cuda().contiguous() if tag == "mask" else None
    qkv = torch.empty(B * N, 3 * D, device="cuda", dtype=BF); att = torch.empty(B * N, D, device="cuda", dtype=BF)
    lse = torch.empty(B * H * N, device="cuda"); y = torch.empty(B * N, D, device="cuda", dtype=BF)
    ops.gemm_nt(x, w_in, qkv, bias=b_in)
    ops.attn_fwd(qkv, att, lse, km, B, N, H)
    ops.gemm_nt(att, w_out, y, bias=b_out)
    e_y = rel(y.view(B, N, D), f[f"y_{tag}"])
    # backward: d att = gy Wo ; (dq, dk, dv) ; gx = dqkv Win ; weight / bias gradients by the wgrad kernel
    gy = dev16(f[f"gy_{tag}"]).view(B * N, D)
    datt = torch.empty_like(att); dqkv = torch.empty_like(qkv); delta = torch.empty(B * H * N, device="cuda"); gx = torch.empty_like(x)
    ops.gemm_nt(gy, dev16(f["output_proj.weight"].t()), datt)
    ops.attn_bwd(qkv, att, datt, lse, km, dqkv, delta, B, N, H)
    ops.gemm_nt(dqkv, dev16(f["input_proj.weight"].t()), gx)
    dw_in = torch.zeros(3 * D, D, device="cuda"); db_in = torch.zeros(3 * D, device="cuda")
    dw_out = torch.zeros(D, D, device="cuda"); db_out = torch.zeros(D, device="cuda")
    ops.gemm_tn(dqkv, x, dw_in, db=db_in, nsplit=1)
    ops.gemm_tn(gy, att, dw_out, db=db_out, nsplit=1)
    torch.cuda.synchronize()
    errs = {"y": e_y, "gx": rel(gx.view(B, N, D), f[f"gx_{tag}"]),
            "d input_proj.weight": rel(dw_in, f[f"grad_{tag}.input_proj.weight"]), "d input_proj.bias": rel(db_in, f[f"grad_{tag}.input_proj.bias"]),
            "d output_proj.weight": rel(dw_out, f[f"grad_{tag}.output_proj.weight"]), "d output_proj.bias": rel(db_out, f[f"grad_{tag}.output_proj.bias"])}
    print("mhsa", tag, {k: round(v, 5) for k, v in errs.items()})
    assert errs["y"] < 1e-2                       # one bf16 rounding each of qkv, the attention output and y
    assert errs["d output_proj.bias"] < 1e-5      # column sums of the fixture's own gy in fp32
    for k in ("gx", "d input_proj.weight", "d input_proj.bias", "d output_proj.weight"):
        assert errs[k] < 2e-2, (k, errs[k])      # bf16 dqkv / datt in between


# ---------------------------------------------------------------------------------------------------------------
# a2: TransformerEncoder, pre-norm (the ViT blocks of Engine.forward_image / Engine.backward)
# ---------------------------------------------------------------------------------------------------------------
def _enc_config(**kw):
    from medmoe_amd.config import MedMoEConfig
    base = dict(img_size=64, patch=16, d_v=128, n_layer_v=2, n_head_v=2, ff_v=256, vocab=97, max_len=16, d_t=128, n_layer_t=2, n_head_t=2,
                ff_t=256, n_expert=4, top_k=1, d_out=128)
    base.update(kw)
    return MedMoEConfig(**base)


def test_prenorm_encoder_reference_fixture(golden_dir):
    from medmoe_amd.engine import Engine
    f = load(golden_dir, "enc_prenorm_mfma.npz")
    B, N, D, L = 4, 17, 128, 2
    eng = Engine(_enc_config(), "cuda:0")
    assert eng.cfg.n_tok_v == N
    eng._alloc(B)
    p, ws = eng.params, eng.ws
    names = [k for k in f if k.startswith("layer.") or k.startswith("final_layer_norm.")]
    assert len(names) == 12 * L + 2
    for k in names:
        p.f32("vit." + k).copy_(f[k].cuda().reshape(p.shapes["vit." + k]))
    p.sync_working_copies()
    ws["x0"].copy_(dev16(f["x"]).view(B * N, D))
    eng._vit_blocks(B)
    torch.cuda.synchronize()
    e_h = [rel(ws[f"x{l}"].view(B, N, D), f[f"hs{l}"]) for l in range(L + 1)]
    e_last = rel(ws["lnf"].view(B, N, D), f["last"])
    print("prenorm hidden states", [round(v, 5) for v in e_h], "last", round(e_last, 5))
    assert e_h[0] == 0.0 and max(e_h) < 1e-2 and e_last < 1e-2
    # backward from the gradient of the final LayerNorm's output
    p.zero_grad()
    ws["dln"].copy_(dev16(f["gy"]).view(B * N, D))
    eng._wgrad_begin()
    eng._wait(eng._vit_backward(stage_grads=False))
    torch.cuda.synchronize()
    e_gx = rel(ws["dxa"].view(B, N, D), f["gx"])
    errs = {k: rel(p.grad("vit." + k), f["grad." + k].reshape(p.shapes["vit." + k])) for k in names}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("prenorm gx", round(e_gx, 5), "worst grads", [(k, round(v, 5)) for k, v in worst], "median", float(np.median(list(errs.values()))))
    assert e_gx < 2e-2
    assert max(errs.values()) < 3e-2, worst
    assert float(np.median(list(errs.values()))) < 1.5e-2


# ---------------------------------------------------------------------------------------------------------------
# a2: TransformerEncoder, post-norm (the frozen text tower's blocks), padded and packed variable-length paths
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("varlen", [False, True])
def test_postnorm_encoder_reference_fixture(golden_dir, varlen):
    from medmoe_amd.engine import Engine
    f = load(golden_dir, "enc_postnorm_mfma.npz")
    B, T, D, L = 4, 16, 128, 2
    eng = Engine(_enc_config(), "cuda:0")
    eng.text_varlen = varlen
    eng._alloc(B)
    t = eng.params.text
    for k in f:
        if k.startswith("layer."):
            assert k in t, k
            t[k] = f[k].cuda().to(t[k].dtype).contiguous()
    km = f["key_mask"]
    lens = km.sum(1)
    ids = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        n = int(lens[b])
        ids[b, :n] = torch.arange(3, 3 + n); ids[b, 0] = 1; ids[b, n - 1] = 2
    eng.forward_text(ids.cuda(), km.long().cuda(), None, embedded=dev16(f["x"]))
    torch.cuda.synchronize()
    ws = eng.ws
    valid = km.reshape(-1)
    if varlen:
        tok_row = ws["tpack"][:B * T].long().cpu()
        pick = lambda buf: buf[tok_row[valid].cuda()].float().cpu()
    else:
        pick = lambda buf: buf[:B * T].float().cpu()[valid]
    errs = [rel(pick(ws[f"ths{l}"]), f[f"hs{l}"].reshape(B * T, D)[valid]) for l in range(L + 1)]
    print("postnorm hidden states, varlen =", varlen, [round(v, 5) for v in errs])
    assert errs[0] == 0.0 and max(errs) < 1e-2
    if not varlen:          # the padded path computes the padding queries too (masked keys only), as the reference does
        assert rel(ws[f"ths{L}"][:B * T].view(B, T, D), f[f"hs{L}"]) < 1e-2
        assert rel(ws[f"ths{L}"][:B * T].view(B, T, D), f["last"]) < 1e-2       # no final LayerNorm in the text geometry


# ---------------------------------------------------------------------------------------------------------------
# a4: router
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prefix", ["", "tie."])
def test_router_reference_fixture(golden_dir, prefix):
    from medmoe_amd import ops
    f = load(golden_dir, "router_mfma.npz")
    x = f["x"].cuda().contiguous()
    B, Dv = x.shape
    w1, b1, w2, b2 = (f[prefix + "router." + n].cuda().contiguous() for n in ("0.weight", "0.bias", "2.weight", "2.bias"))
    Hd, E = w1.shape[0], w2.shape[0]
    for k in (1, 2):
        h = torch.empty(B, Hd, device="cuda"); probs = torch.empty(B, E, device="cuda")
        idx = torch.empty(B, k, device="cuda", dtype=torch.int32); gates = torch.empty(B, k, device="cuda")
        ops.call("router_fwd", x, w1, b1, w2, b2, h, probs, idx, gates, B, Dv, Hd, E, k)
        torch.cuda.synchronize()
        assert torch.allclose(probs.cpu(), f[prefix + "probs"], rtol=2e-5, atol=1e-7), float((probs.cpu() - f[prefix + "probs"]).abs().max())
        assert torch.equal(idx[:, 0].cpu().long(), f[prefix + "top1"])                     # bit-exact arg-max
        if k == 2:
            want = f[prefix + "probs"].clone()
            want[torch.arange(B), f[prefix + "top1"]] = -1.0
            assert torch.equal(idx[:, 1].cpu().long(), want.argmax(1))                      # second choice = arg-max of the rest (lowest index on ties)
    if prefix:
        pr = f["tie.probs"]
        assert bool((pr[:, 1] == pr[:, 3]).all()) and bool((f["tie.top1"] == 1).all())      # the tie is exact in the reference, lowest index wins
        assert torch.equal(probs[:, 1], probs[:, 3])                                        # ... and exact in the kernel


# ---------------------------------------------------------------------------------------------------------------
# a5 / a6: MoE (router -> dispatch -> experts -> combine), forward + every gradient
# ---------------------------------------------------------------------------------------------------------------
def test_moe_reference_fixture(golden_dir):
    from medmoe_amd.engine import Engine
    f = load(golden_dir, "moe_mfma.npz")
    B, P, D, E = 8, 16, 128, 4
    cfg = _enc_config(n_layer_v=4)
    eng = Engine(cfg, "cuda:0")
    assert cfg.stage_layers() == [1, 2, 3, 4] and cfg.n_patch == P and cfg.router_hidden == 128
    eng._alloc(B)
    p, ws = eng.params, eng.ws
    p.load_named({"moe." + k: v for k, v in f.items() if k.startswith("experts.") or k.startswith("router.")})
    Nt = cfg.n_tok_v
    for s, l in enumerate(cfg.stage_layers()):
        ws[f"x{l}"].zero_()
        ws[f"x{l}"].view(B, Nt, D)[:, 1:].copy_(dev16(f[f"f{s}"]))
    ws["router_in"].copy_(f["rin"].cuda())
    eng._moe_forward(B)
    torch.cuda.synchronize()
    top1 = f["top1"]
    assert len(set(top1.tolist())) >= 3 and len(set(top1.tolist())) < E        # three active experts and an empty one
    assert torch.equal(ws["idx"][:, 0].cpu().long(), top1)
    assert torch.allclose(ws["probs"].cpu(), f["probs"], rtol=2e-5, atol=1e-7)
    local = f["local"].reshape(B, D, P).transpose(1, 2)                        # [B, D, H, W] -> [B, P, D]
    e_l, e_g = rel(ws["img_l"], local), rel(ws["img_g"], f["global"])
    print("moe forward: local", round(e_l, 5), "global", round(e_g, 5))
    assert e_l < 1e-2 and e_g < 1e-2
    # backward: gradients of (global * gg).sum() + (local * gl).sum() + (probs * gp).sum()
    p.zero_grad(); ws["loss_parts"].zero_()
    ws["d_img_g"].copy_(f["gg"].cuda())
    ws["d_img_l"].copy_(dev16(f["gl"].reshape(B, D, P).transpose(1, 2)))
    eng._wgrad_begin()
    eng._wait(eng._moe_backward(None, 1.0, dprobs_ext=f["gp"].cuda().contiguous()))
    torch.cuda.synchronize()
    slot = ws["slot_of"].long().cpu()
    errs = {"g_rin": rel(ws["drouter_in"], f["g_rin"])}
    for s in range(4):
        got = ws["dF"][s].view(-1, P, D)[slot.cuda()]                          # top-1: slot_of[b] is sample b's only slot
        errs[f"gf{s}"] = rel(got, f[f"gf{s}"])
    named = p.export_named(p.g32)
    for k in f:
        if k.startswith("grad."):
            want = f[k]
            got = named["moe." + k[5:]].reshape(want.shape)
            if k.endswith("attn_proj.2.bias"):
                # the softmax over the four scales is invariant to a shared logit bias: the reference's gradient is rounding noise around
                # zero (<= 2e-6 here) and so is the kernel's (fp32 sum of terms that cancel)
                assert float(want.abs().max()) < 1e-5 and float(got.abs().max()) < 1e-5, (k, got, want)
            elif float(want.norm()) < 1e-12:
                assert float(got.norm()) == 0.0, k                              # the empty expert's weights get exactly nothing
            else:
                errs[k] = rel(got, want)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("moe backward worst", [(k, round(v, 5)) for k, v in worst], "median", float(np.median(list(errs.values()))))
    assert errs["g_rin"] < 1e-4                                                # fp32 router backward
    for k, v in errs.items():
        bar = 1e-4 if k.startswith("grad.router") else (6e-2 if "attn_proj" in k else 3e-2)
        assert v < bar, (k, v, worst)


# ---------------------------------------------------------------------------------------------------------------
# a9 / a10: the two GLoRIA losses through src.losses
# ---------------------------------------------------------------------------------------------------------------
def test_gloria_global_reference_fixture(golden_dir):
    from src.losses import GLORIAGlobalContrastiveLoss
    f = load(golden_dir, "gloria_global_mfma.npz")
    a = f["img"].cuda().requires_grad_(True); t = f["txt"].cuda().requires_grad_(True)
    loss = GLORIAGlobalContrastiveLoss()(a, t, temp3=10.0)
    loss.backward()
    assert abs(float(loss) - float(f["loss"])) < 1e-5 * abs(float(f["loss"]))
    assert rel(a.grad, f["g_img"]) < 1e-4 and rel(t.grad, f["g_txt"]) < 1e-4


@pytest.mark.parametrize("word_grad", [False, True])
def test_gloria_local_reference_fixture(golden_dir, word_grad):
    from src.losses import GLORIALocalContrastiveLoss
    f = load(golden_dir, "gloria_local_mfma.npz")
    img = f["img_l"].cuda().requires_grad_(True)
    words = f["words"].cuda().requires_grad_(word_grad)
    caps = [int(v) for v in f["cap_lens"]]
    out = GLORIALocalContrastiveLoss()(img, words, caps, temp1=4.0, temp2=5.0, temp3=10.0)
    (out.loss0 + out.loss1).backward()
    torch.cuda.synchronize()
    if word_grad:
        # d loss / d words (losses.py:985-1012 differentiates the word embeddings): through the scores (dS . ctx, one NT GEMM over the
        # row-major pair matrix) and through each word's own norm in the cosine; positions at or beyond a caption's length get exactly zero
        gw = f["g_words"]
        e_w = rel(words.grad, gw)
        print(f"gloria local: d words {e_w:.5f}")
        assert e_w < 2e-2
        for i, n in enumerate(caps):
            assert float(words.grad[i, :, n:].abs().max() if n < words.shape[2] else 0.0) == 0.0 and float(gw[i, :, n:].abs().max() if n < gw.shape[2] else 0.0) == 0.0
    e0 = abs(float(out.loss0) - float(f["loss0"])) / abs(float(f["loss0"]))
    e1 = abs(float(out.loss1) - float(f["loss1"])) / abs(float(f["loss1"]))
    e_att = max(rel(out.att_maps[i], f[f"att{i}"]) for i in range(len(caps)))
    e_g = rel(img.grad, f["g_img_l"])
    print(f"gloria local: loss0 {e0:.5f} loss1 {e1:.5f} att maps {e_att:.5f} d img_l {e_g:.5f}")
    assert len(out.att_maps) == len(caps) and all(tuple(out.att_maps[i].shape) == (1, caps[i], 8, 8) for i in range(len(caps)))
    assert e0 < 5e-3 and e1 < 5e-3            # fp16 log-probabilities + bf16 attention weights inside, fp32 sums
    assert e_att < 1e-2
    assert e_g < 2e-2                          # the loss-kernel bar of tests/test_engine_gpu.py::test_gradients stage (1)
