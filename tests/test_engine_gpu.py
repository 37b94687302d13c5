"""GPU parity of the whole hot path (HIP engine through the C-ABI) against the CPU oracle on
identical seeded inputs.  The engine computes in bf16 with fp32 accumulation, the oracle in
fp32 on the same bf16-rounded weights/images; tolerances (relative L2) are stated inline.
Router top-k indices must match exactly."""
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


def structured_images(base, seed, amp=2.0):
    """base/2 + a per-sample 4x4 block pattern: the mean-pooled router input then carries a sample-specific component (plain randn
    images average out over 576 patches and every sample of a small batch picks the same expert pair)."""
    B, _, size, _ = base.shape
    low = torch.randn(B, 3, 4, 4, generator=torch.Generator().manual_seed(seed))
    return base * 0.5 + amp * torch.nn.functional.interpolate(low, size=(size, size), mode="nearest")


# tinyL336 (576 patches) routes all four randn samples to ONE expert pair; seed 3 + structured images give the oracle three distinct
# pairs ({0,1}, {0,2}, {1,2}; second/third-choice margin 0.017) - asserted in the tests
SEED_OF = {"tinyL336": 3}


def make(cfg_name, B, seed=0, struct=None):
    seed = SEED_OF.get(cfg_name, seed)
    struct = (cfg_name == "tinyL336") if struct is None else struct
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    ocfg = O.config_by_name(cfg_name)
    cfg = config_by_name(cfg_name)
    p = O.init_params(ocfg, seed=seed, std=0.05)
    g = torch.Generator().manual_seed(seed + 7)
    for k in p:     # non-trivial LN affine + biases; the router weights are scaled up below (top-2 then spreads over the experts;
                    # top-1 over several experts is covered by tests/test_parity2_gpu.py::test_top1_routing_over_several_experts)
        if k.endswith("layernorm.weight") or k.endswith("layer_norm.weight"):
            p[k] = 1 + 0.2 * torch.randn(p[k].shape, generator=g)
        elif k.endswith(".bias"):
            p[k] = 0.05 * torch.randn(p[k].shape, generator=g)
    p["moe.router.0.weight"] *= 8.0
    p["moe.router.2.weight"] *= 8.0
    # weights that the engine keeps in bf16 are rounded for the oracle too
    for k in p:
        if k.endswith(".weight") and p[k].dim() >= 2 and not k.startswith("moe.router") and "embeddings" not in k:
            p[k] = bf_round(p[k])
    batch = O.synthetic_batch(ocfg, B, min_len=4)
    if struct:
        batch["image"] = structured_images(batch["image"], seed + 99)
    batch["image"] = bf_round(batch["image"])
    eng = Engine(cfg, "cuda:0")
    eng.params.load_named(p)
    return ocfg, cfg, p, batch, eng


def to_dev(batch):
    return {k: v.cuda() for k, v in batch.items()}


# tinyL = BASELINE configs[4]'s token geometry (patch 14 -> 588-wide patch rows padded to 640, 256 regions, 257 tokens)
# tinyL336 = BASELINE configs[3]'s token geometry (336 px / patch 14: 577 tokens -> the 37-tile attention kernels, 576 regions -> the
# generic-geometry local loss)
@pytest.mark.parametrize("cfg_name", ["tiny", "tiny2", "tinyL", "tinyL336"])
def test_forward_and_losses(cfg_name):
    B = 4 if cfg_name == "tinyL336" else 8
    ocfg, cfg, p, batch, eng = make(cfg_name, B)
    ref = O.model_step(batch, p, ocfg, O.Vocab.synthetic(ocfg.vocab))
    out_l = eng.train_step(to_dev(batch), optimizer=False)
    torch.cuda.synchronize()
    out = eng.outputs()
    assert np.array_equal(out["cap_lens"].cpu().numpy(), np.asarray(ref["cap_lens"]))
    assert rel(out["txt_g"], ref["txt_g"]) < 2e-2 and rel(out["txt_l"], ref["txt_l"]) < 2e-2
    # routing: identical expert choice wherever the oracle's own top-2 margin is not a near tie
    pr = ref["probs"]
    assert rel(out["probs"], pr) < 2e-2
    srt = pr.sort(dim=1, descending=True).values
    k = ocfg.top_k
    safe = (srt[:, k - 1] - srt[:, k]) > 5e-3 if k < pr.shape[1] else torch.ones(B, dtype=torch.bool)
    assert safe.float().mean() > 0.5
    assert torch.equal(out["idx"].cpu().long()[safe], ref["idx"][safe])
    if cfg_name == "tinyL336":       # top-2 dispatch at this geometry over more than one expert pair
        assert len({tuple(sorted(r.tolist())) for r in ref["idx"]}) >= 2, ref["idx"]
    if bool(safe.all()):
        assert rel(out["img_g"], ref["img_g"]) < 2e-2 and rel(out["img_l"], ref["img_l"]) < 2e-2
        c = lambda t: t.detach().float().cpu() - t.detach().float().cpu().mean(0, keepdim=True)     # batch-mean-centred
        assert rel(c(out["img_g"]), c(ref["img_g"])) < 4e-2 and rel(c(out["img_l"]), c(ref["img_l"])) < 4e-2
        assert abs(out_l["g_loss"].item() - ref["g_loss"].item()) < 5e-3 * abs(ref["g_loss"].item())
        assert abs(out_l["l_loss"].item() - ref["l_loss"].item()) < 1e-2 * abs(ref["l_loss"].item())
    assert abs(out_l["classifier_loss"].item() - ref["classifier_loss"].item()) < 1e-2
    assert abs(out_l["classifier_acc"].item() - ref["classifier_acc"].item()) < 1e-6 or not bool(safe.all())


def test_odd_batch_all_captions_full_length():
    """B = 5 with every caption at max_len: the ragged pair matrices are wider (rounded up to the GEMM k-step) than
    B*Tp; local + global losses still match the oracle."""
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    ocfg, cfg = O.config_by_name("tiny"), config_by_name("tiny")
    p = O.init_params(ocfg, seed=3, std=0.05)
    for k in p:
        if k.endswith(".weight") and p[k].dim() >= 2 and not k.startswith("moe.router") and "embeddings" not in k:
            p[k] = bf_round(p[k])
    batch = O.synthetic_batch(ocfg, 5, min_len=ocfg.max_len)
    batch["image"] = bf_round(batch["image"])
    eng = Engine(cfg, "cuda:0")
    eng.params.load_named(p)
    ref = O.model_step(batch, p, ocfg, O.Vocab.synthetic(ocfg.vocab))
    out_l = eng.train_step(to_dev(batch), optimizer=False)
    torch.cuda.synchronize()
    assert np.array_equal(eng.outputs()["cap_lens"].cpu().numpy(), np.asarray(ref["cap_lens"]))
    assert abs(out_l["l_loss"].item() - ref["l_loss"].item()) < 3e-2 * max(1.0, abs(ref["l_loss"].item()))
    assert abs(out_l["g_loss"].item() - ref["g_loss"].item()) < 3e-2 * max(1.0, abs(ref["g_loss"].item()))


def test_training_loop_reduces_loss_and_tracks_oracle_adam():
    """Forty optimisation steps on one batch: the loss goes down, nothing turns non-finite, and after the first steps the
    fused clip + Adam update moved the parameters the way torch's clip_grad_norm_ + Adam move the oracle's."""
    import copy
    ocfg, cfg, p, batch, eng = make("tiny", 8, seed=2)
    eng.cfg.lr = 1e-3
    b = to_dev(batch)
    # oracle: three steps of torch Adam on the same model
    po = {k: v.clone().requires_grad_(not k.startswith("text.")) for k, v in p.items()}
    train = [v for k, v in po.items() if not k.startswith("text.")]
    opt = torch.optim.Adam(train, lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        O.model_step(batch, po, ocfg, O.Vocab.synthetic(ocfg.vocab))["loss"].backward()
        torch.nn.utils.clip_grad_norm_(train, cfg.clip)
        opt.step()
    losses = []
    for it in range(40):
        losses.append(float(eng.train_step(b)["loss"]))
        if it == 2:
            named = eng.params.export_named()
            moved = {k: (named[k].reshape(p[k].shape).cpu() - p[k]) for k in ("vit.layer.0.feedforward.model.0.weight", "moe.router.0.weight", "vit.pos_embed")}
            for k, dv in moved.items():
                do = po[k].detach() - p[k]
                cos = float((dv * do).sum() / (dv.norm() * do.norm() + 1e-30))
                assert cos > 0.9, (k, cos)                       # Adam's sign-like update: direction agreement, bf16 forward noise
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.1, losses[::8]
    assert torch.isfinite(eng.params.p32).all()


def test_gradient_accumulation_equals_one_step():
    """train_step(zero_grad=False, loss_scale=1/2) twice on the same micro-batch leaves the same flat gradient as one
    plain step (every wgrad / bias / LN / embedding gradient kernel accumulates)."""
    ocfg, cfg, p, batch, eng = make("tiny2", 8)
    b = to_dev(batch)
    eng.train_step(b, optimizer=False)
    torch.cuda.synchronize()
    g1 = eng.params.g32.clone()
    eng.train_step(b, optimizer=False, zero_grad=True, loss_scale=0.5)
    eng.train_step(b, optimizer=False, zero_grad=False, loss_scale=0.5)
    torch.cuda.synchronize()
    g2 = eng.params.g32
    assert rel(g2, g1) < 2e-3, rel(g2, g1)


@pytest.mark.parametrize("cfg_name", ["tiny", "tiny2", "tinyL", "tinyL336"])
def test_gradients(cfg_name):
    """Parameter gradients against the oracle's autograd.  Two stages, so that the comparison measures KERNELS and not the
    conditioning of the loss (profiles/r02_notes.md): (1) the loss kernels at the engine's own bf16 tower outputs (same inputs on
    both sides); (2) the router backward by fp32 autograd on the engine's own router input; (3) experts + ViT backward with the
    engine's gradients at the tower outputs and at the router input pushed through the oracle's graph.  For tiny / tiny2 the
    whole fp32 chain is compared as well (their loss is well conditioned)."""
    B = 4 if cfg_name == "tinyL336" else 8
    ocfg, cfg, p, batch, eng = make(cfg_name, B, seed=3)
    vocab = O.Vocab.synthetic(ocfg.vocab)
    pr = {k: v.clone().requires_grad_(not k.startswith("text.")) for k, v in p.items()}
    ref = O.model_step(batch, pr, ocfg, vocab)
    ref["loss"].backward()
    eng.train_step(to_dev(batch), optimizer=False)
    torch.cuda.synchronize()
    if not torch.equal(eng.outputs()["idx"].cpu().long(), ref["idx"]):
        pytest.skip("near-tie routing differs between bf16 and fp32 towers on this seed")
    if cfg_name == "tinyL336":
        assert len({tuple(sorted(r.tolist())) for r in ref["idx"]}) >= 2, ref["idx"]
    got = eng.params.export_named(eng.params.g32)
    P, Do, Hh = cfg.n_patch, cfg.d_out, int(cfg.n_patch ** 0.5)

    def compare(tag, bar, want):
        worst = {}
        for k, v in pr.items():
            if k.startswith("text.") or not want(k):
                continue
            gref = v.grad if v.grad is not None else torch.zeros_like(v)
            g = got[k].reshape(gref.shape)
            if gref.norm() < 1e-7:
                assert g.norm() < 1e-4, k
                continue
            worst[k] = rel(g, gref)
        print(tag, "worst grads:", sorted(worst.items(), key=lambda kv: -kv[1])[:6], "median", float(np.median(list(worst.values()))))
        bad = {k: e for k, e in worst.items() if e > bar(k)}
        assert not bad, (tag, sorted(bad.items(), key=lambda kv: -kv[1])[:10])

    # (1) loss kernels in isolation: local + global loss differentiated by the oracle at the engine's own outputs
    x = eng.ws["img_l"].float().cpu().transpose(1, 2).reshape(B, Do, Hh, Hh).requires_grad_(True)
    xg = eng.ws["img_g"].float().cpu().requires_grad_(True)
    l0, l1, _ = O.gloria_local(x, eng.ws["words"].float().cpu().transpose(1, 2), ref["cap_lens"], ocfg.temp1, ocfg.temp2, ocfg.temp3)
    (ocfg.w_local * (l0 + l1) + ocfg.w_global * O.gloria_global(xg, eng.ws["txt_g"].float().cpu(), ocfg.temp3)).backward()
    e_l = rel(eng.ws["d_img_l"].float(), x.grad.reshape(B, Do, P).transpose(1, 2))
    e_g = rel(eng.ws["d_img_g"], xg.grad)
    print(f"loss kernels in isolation: d img_l {e_l:.4f}  d img_g {e_g:.4f}")
    assert e_l < 2e-2 and e_g < 1e-3, (e_l, e_g)
    if cfg_name in ("tiny", "tiny2"):
        compare("fp32 chain", lambda k: 0.15 if "attn_proj" in k else 0.08, lambda k: True)
    # (2) router backward in isolation: fp32 autograd on the engine's OWN router input, gate gradients and labels.  (Against the
    # oracle chain the router path is conditioning-limited: the test scales the router weights x8 per layer so that routing
    # spreads, and a 0.2 % difference of the pooled router input then moves d router_in by 8.7 % - tools/grad_diag.py.)
    k_top = cfg.top_k
    rin = eng.ws["router_in"].cpu().clone().requires_grad_(True)
    pw = {kk: p[kk].clone().requires_grad_(True) for kk in ("moe.router.0.weight", "moe.router.0.bias", "moe.router.2.weight", "moe.router.2.bias")}
    probs_r = O.router_probs(rin, pw)
    obj = ocfg.w_cls * O.router_ce(probs_r, batch["label"])
    if k_top > 1:
        obj = obj + (O.gates_from_probs(probs_r, ref["idx"]) * eng.ws["dgate"].cpu().view(B, k_top)).sum()
    obj.backward()
    assert rel(eng.ws["drouter_in"], rin.grad) < 1e-4
    for kk, v in pw.items():
        assert rel(got[kk].reshape(v.shape), v.grad) < 1e-4, kk
    # (3) experts + ViT backward in isolation: the engine's gradients at the tower outputs AND at the router input pushed through
    # the oracle's graph (the router itself cut out: it was checked in (2))
    for v in pr.values():
        v.grad = None
    last, hs = O.vit_forward(batch["image"], pr, ocfg)
    router_in = last[:, 1:, :].mean(dim=1)
    feats = [hs[l][:, 1:, :] for l in ocfg.stage_layers()]
    img_g2, img_l2, _, _ = O.moe_forward(feats, router_in.detach(), pr, ocfg.n_expert, ocfg.top_k)
    obj = (img_g2 * eng.ws["d_img_g"].cpu()).sum() + (img_l2.reshape(B, Do, P) * eng.ws["d_img_l"].float().cpu().transpose(1, 2)).sum() \
        + (router_in * eng.ws["drouter_in"].cpu()).sum()
    obj.backward()
    compare("experts + ViT backward", lambda k: 0.15 if "attn_proj" in k else 0.08, lambda k: not k.startswith("moe.router"))


def test_router_bit_exact():
    """router top-k indices bit-exact vs the fixed-order numpy restatement; probabilities to 1 ulp-ish."""
    from medmoe_amd import ops
    rng = np.random.default_rng(0)
    B, Dv, Hd, E, k = 64, 192, 128, 8, 2
    x = rng.standard_normal((B, Dv)).astype(np.float32)
    w1 = (rng.standard_normal((Hd, Dv)) * 0.2).astype(np.float32); b1 = (rng.standard_normal(Hd) * 0.1).astype(np.float32)
    w2 = (rng.standard_normal((E, Hd)) * 0.2).astype(np.float32); b2 = (rng.standard_normal(E) * 0.1).astype(np.float32)
    w2[5] = w2[2]; b2[5] = b2[2]                       # exact tie between experts 2 and 5 -> lowest index first
    probs_ref, idx_ref, logits_ref, h_ref = O.router_fixed_order(x, w1, b1, w2, b2, k)
    d = lambda a: torch.from_numpy(a).cuda()
    h = torch.empty(B, Hd, device="cuda"); probs = torch.empty(B, E, device="cuda")
    idx = torch.empty(B, k, device="cuda", dtype=torch.int32); gates = torch.empty(B, k, device="cuda")
    ops.call("router_fwd", d(x), d(w1), d(b1), d(w2), d(b2), h, probs, idx, gates, B, Dv, Hd, E, k)
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.array_equal(h.cpu().numpy(), h_ref)          # hidden layer bit for bit (fixed fp32 order, no FMA)
    assert np.allclose(probs.cpu().numpy(), probs_ref, rtol=1e-5, atol=1e-7)   # expf differs by a few ulp
    for b in range(B):                                  # tie: 5 never precedes 2
        row = idx_ref[b].tolist()
        if 5 in row and 2 in row:
            assert row.index(2) < row.index(5)


@pytest.mark.parametrize("B,HW,T,D,caps", [(8, 64, 16, 128, [16, 3, 9, 1, 12, 16, 7, 5]),
                                          (4, 196, 77, 768, [77, 8, 40, 23]), (2, 196, 25, 768, [25, 6])])
def test_local_loss_kernel_vs_oracle(B, HW, T, D, caps):
    """local_pair forward sim matrix and its ctx gradient against the oracle on bf16-rounded inputs."""
    from medmoe_amd import ops
    torch.manual_seed(0)
    sc = 0.5 if D == 128 else 0.2
    ctx = bf_round(torch.randn(B, HW, D) * sc); words = bf_round(torch.randn(B, T, D) * sc)
    cap = torch.tensor(caps)
    hh = int(HW ** 0.5)
    img_l = ctx.transpose(1, 2).reshape(B, D, hh, hh).clone().requires_grad_(True)
    sim_ref, _ = O.gloria_local_sim(img_l, words.transpose(1, 2), cap.tolist(), 4.0, 5.0)
    gs = torch.randn(B, B) * 0.1
    (sim_ref * gs).sum().backward()
    dctx_ref = img_l.grad.reshape(B, D, HW).transpose(1, 2)
    HWp, Tp, GW = ops.local_geometry(HW, T)
    dev = "cuda"
    c16 = ctx.to(dev).to(torch.bfloat16).reshape(B * HW, D).contiguous(); w16 = words.to(dev).to(torch.bfloat16).contiguous()
    wn = torch.empty(B, T, device=dev); wT = torch.empty(D, B * Tp, device=dev, dtype=torch.bfloat16)
    ops.call("words_prep", w16, wn, wT, B, T, Tp, D)
    gmp = torch.zeros(B * HWp, GW, device=dev, dtype=torch.bfloat16)
    tl = torch.tensor([[b, m, (b + 1) * HW, 0] for b in range(B) for m in range(b * HW, (b + 1) * HW, 128)], device=dev, dtype=torch.int32)
    cnt = torch.tensor([tl.shape[0]], device=dev, dtype=torch.int32)
    ar = torch.arange(B * HW, device=dev)
    ops.gemm_nt(c16, c16, gmp, c_rowmap=(ar // HW * HWp + ar % HW).int(), tiles=tl, tile_count=cnt, max_tiles=tl.shape[0],
                stride_b=HW * D, M=B * HW, N=HW, col_perm=True)
    sim = torch.empty(B, B, device=dev); capd = cap.int().to(dev)
    # (a) fused per-pair kernel (streams ctx/words itself)
    ops.call("local_pair", c16, w16, gmp, wn, capd, None, sim, None, None, None, None, None, None, B, B, HW, T, D, 4.0, 5.0, 1e-8, 0)
    torch.cuda.synchronize()
    assert torch.allclose(sim.cpu(), sim_ref.detach(), atol=3e-2, rtol=1e-2), (sim.cpu() - sim_ref.detach()).abs().max()
    # (b) scores GEMM with fused word-softmax (A1, LSE) + pair kernel on the precomputed tiles
    a1 = torch.full((B * HWp, B * Tp), float("nan"), device=dev, dtype=torch.bfloat16); lse = torch.empty(B * HWp, B, device=dev)
    ops.call("local_scores", c16, w16, capd, a1, lse, B, B, HW, T, D)
    s_all = torch.einsum("bhd,itd->biht", ctx, words)            # [B,B,HW,T]
    for bb in range(B):
        for ii in range(B):
            # the tile holds the word-softmax as fp16 log-probabilities S - lse (masked words: -60000)
            ref_lp = torch.log_softmax(s_all[bb, ii, :, :caps[ii]], dim=-1)
            tile = a1.view(torch.float16).view(B, HWp, B, Tp)[bb, :HW, ii]
            assert torch.allclose(tile[:, :caps[ii]].float().cpu(), ref_lp, atol=3e-3, rtol=2e-3)
            assert bool((tile[:, caps[ii]:].float() < -5e4).all())
    sim2 = torch.empty(B, B, device=dev)
    ops.call("local_pair", None, None, gmp, wn, capd, None, sim2, None, None, None, None, a1, lse, B, B, HW, T, D, 4.0, 5.0, 1e-8, 0)
    torch.cuda.synchronize()
    assert torch.allclose(sim2.cpu(), sim_ref.detach(), atol=3e-2, rtol=1e-2), (sim2.cpu() - sim_ref.detach()).abs().max()
    # (c) the lean pair kernel (what the engine runs): sim + gradients in one pass, A overwrites A1 in place
    dS = torch.empty(B * HWp, B * Tp, device=dev, dtype=torch.bfloat16); A = a1; U = torch.empty_like(dS)
    sim3 = torch.empty(B, B, device=dev)
    ops.call("local_pair2", a1, lse, gmp, wn, capd, gs.to(dev), sim3, dS, U, None, B, B, HW, T, 4.0, 5.0, 1e-8)
    torch.cuda.synchronize()
    assert torch.allclose(sim3.cpu(), sim_ref.detach(), atol=3e-2, rtol=1e-2), (sim3.cpu() - sim_ref.detach()).abs().max()
    dC = torch.zeros(B * HWp, D, device=dev)
    ops.gemm_nt(dS, wT, dC)
    dGm = torch.empty(B * HWp, HWp, device=dev, dtype=torch.bfloat16)
    tlp = torch.tensor([[b, m, (b + 1) * HWp, 0] for b in range(B) for m in range(b * HWp, (b + 1) * HWp, 128)], device=dev, dtype=torch.int32)
    cntp = torch.tensor([tlp.shape[0]], device=dev, dtype=torch.int32)
    ops.gemm_nt(U, A, dGm, tiles=tlp, tile_count=cntp, max_tiles=tlp.shape[0], stride_b=HWp * B * Tp, M=B * HWp, N=HWp)
    arp = torch.arange(B * HWp, device=dev)
    ops.gemm_tn(dGm, c16, dC.view(B, HWp, D), x_rowmap=(arp // HWp * HW + torch.clamp(arp % HWp, max=HW - 1)).int(),
                row_off=(torch.arange(B + 1, device=dev) * HWp).int(), n_groups=B, stride_w=HWp * D, nsplit=1, M=B * HWp)
    torch.cuda.synchronize()
    got = dC.view(B, HWp, D)[:, :HW].cpu()
    assert rel(got, dctx_ref) < 5e-2, rel(got, dctx_ref)


@pytest.mark.parametrize("caps", [[77, 8, 40, 23, 50, 64, 16, 33, 1], [30, 30, 31], [70]])
def test_local_loss_ragged_classes_vs_oracle(caps):
    """The RAGGED layout the engine runs (captions grouped into length classes 16..80, one launch per class, tables from
    engine.ragged_layout): sim matrix and ctx gradient against the oracle.  First case: all five classes, classes with
    one and with an odd number of members; then a single class; then a single caption."""
    from medmoe_amd import ops
    from medmoe_amd.engine import ragged_layout
    torch.manual_seed(0)
    B, HW, T, D = len(caps), 196, 77, 768
    ctx = bf_round(torch.randn(B, HW, D) * 0.2); words = bf_round(torch.randn(B, T, D) * 0.2)
    hh = int(HW ** 0.5)
    img_l = ctx.transpose(1, 2).reshape(B, D, hh, hh).clone().requires_grad_(True)
    sim_ref, _ = O.gloria_local_sim(img_l, words.transpose(1, 2), list(caps), 4.0, 5.0)
    gs = torch.randn(B, B) * 0.1
    (sim_ref * gs).sum().backward()
    dctx_ref = img_l.grad.reshape(B, D, HW).transpose(1, 2)
    HWp, Tp, GW = ops.local_geometry(HW, T)
    dev = "cuda"
    I32 = torch.int32
    perm, col, ntts, chunk, classes, Kc, Kp = ragged_layout(np.array(caps), T, Tp)
    d = lambda a: torch.from_numpy(np.asarray(a).astype(np.int32)).to(dev)
    d_perm, d_col, d_tp, d_chunk = d(perm), d(col), d(16 * ntts), d(chunk)
    c16 = ctx.to(dev).to(torch.bfloat16).reshape(B * HW, D).contiguous(); w16 = words.to(dev).to(torch.bfloat16).contiguous()
    wn = torch.empty(B, T, device=dev); wT = torch.zeros(D, Kp, device=dev, dtype=torch.bfloat16)
    ops.call("words_prep_ragged", w16, wn, wT, B, T, Tp, D, d_col, d_tp, Kp)
    gmp = torch.zeros(B * HWp, GW, device=dev, dtype=torch.bfloat16)
    tl = torch.tensor([[b, m, (b + 1) * HW, 0] for b in range(B) for m in range(b * HW, (b + 1) * HW, 128)], device=dev, dtype=I32)
    ar = torch.arange(B * HW, device=dev)
    ops.gemm_nt(c16, c16, gmp, c_rowmap=(ar // HW * HWp + ar % HW).int(), tiles=tl, tile_count=torch.tensor([tl.shape[0]], device=dev, dtype=I32),
                max_tiles=tl.shape[0], stride_b=HW * D, M=B * HW, N=HW, col_perm=True)
    capd = torch.tensor(caps, dtype=I32, device=dev)
    lA = torch.zeros(B * HWp, Kp, device=dev, dtype=torch.bfloat16); ldS = torch.zeros_like(lA); lU = torch.zeros_like(lA)
    lse = torch.empty(B * HWp, B, device=dev); sim = torch.empty(B, B, device=dev)
    for ntt, start, n_c, cbase in classes:
        members = d_perm[start:start + n_c]
        ops.call("local_scores_ragged", c16, w16, capd, lA, lse, B, B, HW, T, D, members, n_c, ntt, cbase, Kp)
        ops.call("local_pair2_ragged", lA, lse, gmp, wn, capd, None, sim, ldS, lU, B, B, HW, T, 4.0, 5.0, 1e-8, members, n_c, ntt, cbase, Kp)
    torch.cuda.synchronize()
    assert torch.allclose(sim.cpu(), sim_ref.detach(), atol=3e-2, rtol=1e-2), (sim.cpu() - sim_ref.detach()).abs().max()
    ops.call("scale_blocks_ragged", ldS, lU, gs.to(dev).contiguous(), B, B, HWp, d_chunk, Kp)
    dC = torch.zeros(B * HWp, D, device=dev)
    ops.gemm_nt(ldS, wT, dC)
    dGm = torch.empty(B * HWp, HWp, device=dev, dtype=torch.bfloat16)
    tlp = torch.tensor([[b, b * HWp, (b + 1) * HWp, 0] for b in range(B)], device=dev, dtype=I32)
    cntp = torch.tensor([B], device=dev, dtype=I32)
    if Kp >= 128:
        ops.gemm_nt(lU, lA, dGm, tiles=tlp, tile_count=cntp, max_tiles=B, stride_b=HWp * Kp, M=B * HWp, N=HWp, tile_rows=256)
    else:
        t128 = torch.tensor([[b, m, (b + 1) * HWp, 0] for b in range(B) for m in range(b * HWp, (b + 1) * HWp, 128)], device=dev, dtype=I32)
        ops.gemm_nt(lU, lA, dGm, tiles=t128, tile_count=torch.tensor([t128.shape[0]], device=dev, dtype=I32), max_tiles=t128.shape[0],
                    stride_b=HWp * Kp, M=B * HWp, N=HWp)
    arp = torch.arange(B * HWp, device=dev)
    ops.gemm_tn(dGm, c16, dC.view(B, HWp, D), x_rowmap=(arp // HWp * HW + torch.clamp(arp % HWp, max=HW - 1)).int(),
                row_off=(torch.arange(B + 1, device=dev) * HWp).int(), n_groups=B, stride_w=HWp * D, nsplit=1, M=B * HWp)
    torch.cuda.synchronize()
    got = dC.view(B, HWp, D)[:, :HW].cpu()
    assert rel(got, dctx_ref) < 5e-2, rel(got, dctx_ref)


def test_src_mirror_model_step_matches_oracle():
    """The reference-named API (src.models..., src.losses) through torch autograd: same loss and the
    same flat gradient as the engine's fused train_step / the oracle."""
    from src.losses import GLORIAGlobalContrastiveLoss, GLORIALocalContrastiveLoss
    from src.models.components.med_moe import MedMoE
    from src.models.medmoe_module import MedMoEPretrainingLightningModule
    B = 8
    ocfg = O.config_by_name("tiny")
    p = O.init_params(ocfg, seed=5, std=0.05)
    for k in p:
        if k.endswith(".weight") and p[k].dim() >= 2 and not k.startswith("moe.router") and "embeddings" not in k:
            p[k] = bf_round(p[k])
    p["moe.router.0.weight"] *= 8.0; p["moe.router.2.weight"] *= 8.0
    batch = O.synthetic_batch(ocfg, B, min_len=4)
    batch["image"] = bf_round(batch["image"])
    pr = {k: v.clone().requires_grad_(not k.startswith("text.")) for k, v in p.items()}
    ref = O.model_step(batch, pr, ocfg, O.Vocab.synthetic(ocfg.vocab))
    ref["loss"].backward()
    model = MedMoE({"config_name": "tiny"}, {})
    model.engine.params.load_named(p)
    loss_cfg = {"global_loss": GLORIAGlobalContrastiveLoss(), "local_loss": GLORIALocalContrastiveLoss(),
                "global_loss_weight": 0.5, "local_loss_weight": 0.5, "classifier_loss_weight": 2.0,
                "temp1": 4.0, "temp2": 5.0, "temp3": 10.0, "soft_label": False}
    lit = MedMoEPretrainingLightningModule(model, loss_cfg)
    dev = {"image": batch["image"].cuda(), "label": batch["label"].cuda(),
           "caption": {"ids": batch["ids"].cuda(), "attn_mask": batch["attn_mask"].cuda(), "token_type": batch["token_type"].cuda()}}
    out = lit.model_step(dev)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert abs(out["loss"].item() - ref["loss"].item()) < 3e-2 * max(1.0, abs(ref["loss"].item()))
    assert abs(out["classifier_loss"].item() - ref["classifier_loss"].item()) < 1e-2
    g_mirror = model.weights.grad.clone()
    # the same step through the engine's fused path must give the same flat gradient (same kernels,
    # different orchestration): tight tolerance; against the fp32 oracle the bf16 bar applies
    eng = model.engine
    eng.train_step({"image": dev["image"], "ids": dev["caption"]["ids"], "attn_mask": dev["caption"]["attn_mask"],
                    "token_type": dev["caption"]["token_type"], "label": dev["label"]}, optimizer=False)
    torch.cuda.synchronize()
    assert rel(g_mirror, eng.params.g32) < 2e-2, rel(g_mirror, eng.params.g32)
    if torch.equal(eng.ws["idx"].cpu().long(), ref["idx"]):
        got = eng.params.export_named(g_mirror)
        errs = []
        for k, v in pr.items():
            if k.startswith("text.") or v.grad is None or v.grad.norm() < 1e-7:
                continue
            errs.append(rel(got[k].reshape(v.grad.shape), v.grad))
        assert max(errs) < 0.15, (np.median(errs), max(errs))


def test_cfg0_reference_config_losses():
    """BASELINE.json configs[0] (the reference's CPU-runnable case: ViT-Ti/16 + 2-layer text tower, 2 experts
    top-1, 32 pairs, T=25): loss terms of the HIP engine vs the oracle."""
    B = 32
    ocfg, cfg, p, batch, eng = make("cfg0", B, seed=11)
    with torch.no_grad():
        ref = O.model_step(batch, p, ocfg, O.Vocab.synthetic(ocfg.vocab))
    out_l = eng.train_step(to_dev(batch), optimizer=False)
    torch.cuda.synchronize()
    out = eng.outputs()
    assert np.array_equal(out["cap_lens"].cpu().numpy(), np.asarray(ref["cap_lens"]))
    assert rel(out["txt_g"], ref["txt_g"]) < 2e-2
    same = torch.equal(out["idx"].cpu().long(), ref["idx"])
    if same:
        assert rel(out["img_g"], ref["img_g"]) < 3e-2
        for k in ("g_loss", "l_loss"):
            assert abs(out_l[k].item() - ref[k].item()) < 3e-2 * max(1.0, abs(ref[k].item())), (k, out_l[k].item(), ref[k].item())
    assert abs(out_l["classifier_loss"].item() - ref["classifier_loss"].item()) < 2e-2
    assert torch.isfinite(eng.params.g32).all()


def test_hipgraph_replay_of_forward_and_backward_matches_the_eager_step(monkeypatch):
    """MEDMOE_GRAPH=1: [zero the gradient + both towers' forward + MoE forward] and [the backward] replayed from hipGraphs (the second stream
    forked and joined inside the capture), losses and optimiser eager.  Same losses (1e-4, 4e-4 from the fifth step on) and the same parameters after eight steps over
    three alternating batches (2e-3 of their norm; the wgrads meet in fp32 atomics) as the eager engine; the first two steps of the graphed
    engine run eagerly, the third captures."""
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    import bench
    cfg = config_by_name("tiny2")
    batches = [bench.synthetic_batch(cfg, 8, 50 + i, "cuda:0") for i in range(3)]
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MEDMOE_GRAPH", mode)
        eng = Engine(config_by_name("tiny2"), "cuda:0", seed=0)
        eng.cfg.lr = 1e-3
        losses = [float(eng.train_step(batches[i % 3])["loss"]) for i in range(8)]
        torch.cuda.synchronize()
        out[mode] = (losses, eng.params.p32.clone(), eng._graph)
    assert out["0"][2] is None and out["1"][2] is not None and out["1"][2]["fwd"] is not None and out["1"][2]["bwd"] is not None
    # two trajectories of eight Adam steps at lr 1e-3 whose weight gradients meet in fp32 atomics: the first steps agree to 1e-5, by the
    # eighth the two runs have drifted apart by up to ~1.5e-4 of the loss (seen once in a dozen runs) - the bar grows with the step
    for i, (a, b) in enumerate(zip(out["0"][0], out["1"][0])):
        assert abs(a - b) < (1e-4 if i < 4 else 4e-4) * max(1.0, abs(a)), (i, out["0"][0], out["1"][0])
    assert rel(out["1"][1], out["0"][1]) < 2e-3
