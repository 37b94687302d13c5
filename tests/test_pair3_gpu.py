"""The TRANSPOSED local-loss pair stage (medmoe_amd/csrc/pair3.hip): one wave per (image, caption, 16-word tile) on
[caption word rows][image region columns] matrices, against the oracle's GLoRIA local loss (losses.py:961-1026)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import medmoe_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
BF, I32 = torch.bfloat16, torch.int32


def bf_round(x):
    return x.to(BF).float()


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))


def plain_gram(ctx_bf16, B, HW, D, GR):
    """[B][GR][GR] bf16 Gram matrices of the images' region vectors, zero outside [HW][HW] (fp32 accumulation, one rounding)."""
    c = ctx_bf16.view(B, HW, D).float()
    gm = torch.zeros(B, GR, GR, device=ctx_bf16.device, dtype=BF)
    gm[:, :HW, :HW] = torch.bmm(c, c.transpose(1, 2)).to(BF)
    return gm


def run_pair3(caps, B, HW, T, D, scale, use_transposed_scores, image_major=True):
    from medmoe_amd import ops
    from medmoe_amd.engine import ragged_layout
    torch.manual_seed(0)
    ctx = bf_round(torch.randn(B, HW, D) * scale); words = bf_round(torch.randn(B, T, D) * scale)
    hh = int(HW ** 0.5)
    img_l = ctx.transpose(1, 2).reshape(B, D, hh, hh).clone().requires_grad_(True)
    sim_ref, att_ref = O.gloria_local_sim(img_l, words.transpose(1, 2), list(caps), 4.0, 5.0)
    gs = torch.randn(B, B) * 0.1
    (sim_ref * gs).sum().backward()
    dctx_ref = img_l.grad.reshape(B, D, HW).transpose(1, 2)
    HWp, Tp, _ = ops.local_geometry(HW, T)
    GR = (HW + 31) // 32 * 32
    dev = "cuda"
    perm, col, ntts, chunk, classes, Kc, Kp = ragged_layout(np.array(caps), T, Tp)
    d = lambda a: torch.from_numpy(np.asarray(a).astype(np.int32)).to(dev)
    d_perm, d_col, d_tp = d(perm), d(col), d(16 * ntts)
    c16 = ctx.to(dev).to(BF).reshape(B * HW, D).contiguous(); w16 = words.to(dev).to(BF).contiguous()
    wn = torch.empty(B, T, device=dev); wT = torch.zeros(D, Kp, device=dev, dtype=BF)
    ops.call("words_prep_ragged", w16, wn, wT, B, T, Tp, D, d_col, d_tp, Kp)
    gm = plain_gram(c16, B, HW, D, GR)
    capd = torch.tensor(caps, dtype=I32, device=dev)
    # element (row, image, region) at row*ld + image*bs + region: image-major (what the engine runs) or [word rows][image region columns]
    PW = GR if image_major else HWp          # region columns stored per (row, image): 224 (64-byte aligned segments) / 208
    ld, bs = (PW, Kp * PW) if image_major else (B * PW, PW)
    rbh = (lambda m: m.view(B, Kp, PW).permute(1, 0, 2)) if image_major else (lambda m: m.view(Kp, B, PW))     # -> [row][image][region]
    new = lambda: torch.full((Kp * B * PW,), float("nan"), device=dev, dtype=BF)
    lse = torch.full((B, B, HWp), float("nan"), device=dev)
    if use_transposed_scores:
        lpT = new()
        for ntt, start, n_c, cbase in classes:
            ops.call("local_scores_t", c16, w16, capd, lpT, lse, B, B, HW, T, D, d_perm[start:start + n_c], n_c, ntt, cbase, ld, bs)
    else:
        lA = torch.zeros(B * HWp, Kp, device=dev, dtype=BF)
        for ntt, start, n_c, cbase in classes:
            ops.call("local_scores_ragged", c16, w16, capd, lA, lse, B, B, HW, T, D, d_perm[start:start + n_c], n_c, ntt, cbase, Kp)
        l2 = (lA.view(torch.float16).t().float() * 1.4426950408889634).clamp_min(-60000.0).to(torch.float16).view(Kp, B, HWp)     # log2 domain
        lpT = new()
        rbh(lpT.view(torch.float16))[:, :, :HWp].copy_(l2)
        rbh(lpT)[:, :, HW:] = float("nan")             # regions >= HW are never written by the score kernel: poison them
    if use_transposed_scores:                             # the tiles themselves: log2-softmax over the caption's words, -60000 beyond
        s_all = torch.einsum("bhd,itd->biht", ctx, words)
        tiles = rbh(lpT.view(torch.float16)).float().cpu()
        for ntt, start, n_c, cbase in classes:
            for jj in range(n_c):
                ii = int(perm[start + jj])
                blk = tiles[cbase + jj * 16 * ntt: cbase + (jj + 1) * 16 * ntt][:, :, :HW]      # [t][b][hw]
                ref = torch.log_softmax(s_all[:, ii, :, :caps[ii]], dim=-1) * 1.4426950408889634                        # [b][hw][t]
                assert torch.allclose(blk[:caps[ii]].permute(1, 2, 0), ref, atol=4e-3, rtol=2e-3)
                assert bool((blk[caps[ii]:] < -5e4).all())
                lref = torch.logsumexp(s_all[:, ii, :, :caps[ii]], dim=-1)
                assert torch.allclose(lse[:, ii, :HW].cpu(), lref, atol=1e-3, rtol=1e-4)
    sim = torch.full((B, B), float("nan"), device=dev)
    att = torch.zeros(B, T, HW, device=dev)
    AT, UT = new(), new()
    stats = torch.full((B, Kp, 2), float("nan"), device=dev)
    for ntt, start, n_c, cbase in classes:                   # forward launch: sim, A, per-word sums, attention maps
        ops.call("local_pair3", lpT, None, AT, None, lse, gm, wn, capd, None, sim, att, stats, Kp, B, B, HW, T, 4.0, 5.0, 1e-8,
                 d_perm[start:start + n_c], n_c, ntt, cbase, ld, bs, PW, None)
    torch.cuda.synchronize()
    assert torch.allclose(sim.cpu(), sim_ref.detach(), atol=3e-2, rtol=1e-2), (sim.cpu() - sim_ref.detach()).abs().max()
    for i in range(B):                                   # attention maps of the matching pairs (losses.py:993-995)
        ref = att_ref[i, i, :caps[i]].detach()                                  # [T_i][HW]
        assert torch.allclose(att[i, :caps[i]].cpu(), ref, atol=2e-3, rtol=2e-2), (att[i, :caps[i]].cpu() - ref).abs().max()
    for m in (AT, UT, lpT):
        rbh(m)[Kc:] = 0
    gsd = gs.to(dev).contiguous()
    d2 = torch.full((B, Kp), float("nan"), device=dev)
    for ntt, start, n_c, cbase in classes:
        ops.call("local_pair3", lpT, lpT, AT, UT, lse, gm, wn, capd, gsd, sim, None, stats, Kp, B, B, HW, T, 4.0, 5.0, 1e-8,
                 d_perm[start:start + n_c], n_c, ntt, cbase, ld, bs, PW, d2)
    torch.cuda.synchronize()
    # the row weights alone (what medmoe_gemm_tn_gram consumes) reproduce the stored U = d2 * A bit for bit
    assert bool(torch.isfinite(d2[:, :Kc]).all())
    u_from_d2 = (rbh(AT)[:Kc].float() * d2[:, :Kc].t()[:, :, None]).to(BF)
    assert torch.equal(u_from_d2, rbh(UT)[:Kc])
    dST = lpT
    for m in (dST, AT, UT):
        assert bool(torch.isfinite(m.float()).all())
        if PW > HW:
            assert float(rbh(m)[:, :, HW:].float().abs().max()) == 0.0     # padding regions are written as zeros
    return dict(dST=dST, AT=AT, UT=UT, wT=wT, c16=c16, dctx_ref=dctx_ref, HWp=PW, Kp=Kp, ld=ld, bs=bs, rbh=rbh, image_major=image_major)


def grads_with_torch(r, B, HW, D):
    """ctx gradient from the pair stage's outputs with fp32 torch matmuls (what the two TN GEMMs of the engine compute)."""
    HWp, rbh = r["HWp"], r["rbh"]
    Wr = r["wT"].float().t()                                                    # [Kp][D]
    dC = torch.einsum("rbh,rd->bhd", rbh(r["dST"]).float(), Wr).contiguous()
    Ub = rbh(r["UT"]).float().permute(1, 2, 0)                                  # [B][HWp][Kp]
    Ab = rbh(r["AT"]).float().permute(1, 0, 2)                                  # [B][Kp][HWp]
    dGm = torch.bmm(Ub, Ab)                                                     # [B][hw][hw']
    c = r["c16"].float().view(B, HW, D)
    dC[:, :HW] += torch.bmm(dGm.transpose(1, 2)[:, :HW, :HW], c)
    return dC[:, :HW].cpu()


@pytest.mark.parametrize("tscores", [True, False])
@pytest.mark.parametrize("caps", [[77, 8, 40, 23, 50, 64, 16, 33, 1], [30, 30, 31], [70], [5, 16, 9, 12, 1, 7, 3, 16, 2, 11, 16, 4, 8, 6, 10, 13, 15, 14]])
def test_pair3_vs_oracle(caps, tscores):
    """All five length classes (one member, odd counts, more captions than one workgroup iteration takes), a single class, a single
    caption: sim, attention maps, and the ctx gradient rebuilt from dS / A / U."""
    B, HW, T, D = len(caps), 196, 77, 768
    r = run_pair3(caps, B, HW, T, D, 0.2, use_transposed_scores=tscores)
    got = grads_with_torch(r, B, HW, D)
    assert rel(got, r["dctx_ref"]) < 5e-2, rel(got, r["dctx_ref"])


def test_pair3_many_iterations_per_workgroup():
    """One caption chunk per image: every workgroup loops over the whole caption list (as at batch 1024), so the flag / mailbox exchanges of
    captions longer than 16 words run back to back for many epochs; twice, bit-identical."""
    from medmoe_amd import ops
    caps = [17 + (7 * i) % 60 for i in range(44)]            # classes 2..5, 8-13 captions each
    B, HW, T, D = len(caps), 196, 77, 768
    ops.local_pair3_chunks(1)
    try:
        r1 = run_pair3(caps, B, HW, T, D, 0.2, use_transposed_scores=True)
        got = grads_with_torch(r1, B, HW, D)
        assert rel(got, r1["dctx_ref"]) < 5e-2, rel(got, r1["dctx_ref"])
        r2 = run_pair3(caps, B, HW, T, D, 0.2, use_transposed_scores=True)
        for k in ("dST", "AT", "UT"):
            assert torch.equal(r1[k], r2[k]), k
    finally:
        ops.local_pair3_chunks(0)


@pytest.mark.parametrize("tscores", [True, False])
def test_pair3_small_geometry(tscores):
    """The 64-region / 16-word instantiation the tiny test models run."""
    caps = [16, 3, 9, 1, 12, 16, 7, 5]
    B, HW, T, D = len(caps), 64, 16, 128
    r = run_pair3(caps, B, HW, T, D, 0.5, use_transposed_scores=tscores)
    got = grads_with_torch(r, B, HW, D)
    assert rel(got, r["dctx_ref"]) < 5e-2, rel(got, r["dctx_ref"])


def grads_with_kernels(r, B, HW, D):
    """The same ctx gradient with the engine's two column-group TN GEMMs (gemm_tn_cols) + the grouped dGm . ctx GEMM."""
    from medmoe_amd import ops
    HWp, Kp, ld, bs, rbh = r["HWp"], r["Kp"], r["ld"], r["bs"], r["rbh"]
    dev = r["dST"].device
    Wr = r["wT"].t().contiguous()                                               # [Kp][D] bf16
    dC = torch.zeros(B * HWp, D, device=dev)
    if r["image_major"]:       # the B image blocks as one [Kp][B*HWp] operand (chunks of HWp columns, bs apart)
        ops.call("gemm_tn_cols", r["dST"], ld, Wr, D, dC, D, Kp, B * HWp, D, 1, 0, 0, 0, HWp, bs)       # as the engine: chunked columns
    else:
        ops.call("gemm_tn_cols", r["dST"], ld, Wr, D, dC, D, Kp, ld, D, 1, 0, 0, 0, 0, 0)
    dGm32 = torch.zeros(B, HWp, HWp, device=dev)
    ops.call("gemm_tn_cols", r["UT"], ld, r["AT"], ld, dGm32, HWp, Kp, HWp, HWp, B, bs, bs, HWp * HWp, 0, 0)
    dGm = dGm32.to(BF).view(B * HWp, HWp)
    arp = torch.arange(B * HWp, device=dev)
    ops.gemm_tn(dGm, r["c16"], dC.view(B, HWp, D), x_rowmap=(arp // HWp * HW + torch.clamp(arp % HWp, max=HW - 1)).int(),
                row_off=(torch.arange(B + 1, device=dev) * HWp).int(), n_groups=B, stride_w=HWp * D, nsplit=1, M=B * HWp)
    torch.cuda.synchronize()
    ref32 = torch.bmm(rbh(r["UT"]).float().permute(1, 2, 0), rbh(r["AT"]).float().permute(1, 0, 2))
    assert rel(dGm32, ref32) < 1e-5, rel(dGm32, ref32)
    return dC.view(B, HWp, D)[:, :HW].cpu()


@pytest.mark.parametrize("image_major", [True, False])
@pytest.mark.parametrize("caps", [[77, 8, 40, 23, 50, 64, 16, 33, 1], [70], [5, 16, 9, 12, 1, 7, 3, 16, 2, 11, 16, 4, 8, 6, 10, 13, 15, 14]])
def test_pair3_gradient_gemms(caps, image_major):
    B, HW, T, D = len(caps), 196, 77, 768
    r = run_pair3(caps, B, HW, T, D, 0.2, use_transposed_scores=True, image_major=image_major)
    ref = grads_with_torch(r, B, HW, D)
    got = grads_with_kernels(r, B, HW, D)
    assert rel(got, ref) < 2e-3, rel(got, ref)                 # same inputs, fp32 accumulation both ways (dGm passes through bf16 in the kernels' path)
    assert rel(got, r["dctx_ref"]) < 5e-2


def test_gemm_tn_cols_wide_rows_exact():
    """Integer-valued operands (exact in bf16 and in the fp32 accumulation): rows 4.3 GB apart from the base, two column groups of
    208 / 104 columns, partial 256-tiles."""
    from medmoe_amd import ops
    dev = "cuda"
    M, ld, Nn, Kk, G = 5184, 425984, 208, 104, 2           # M * ld * 2 bytes = 4.4 GB
    g = torch.Generator(device=dev); g.manual_seed(1)
    big = torch.zeros(M, ld, device=dev, dtype=BF)
    cols = [1000, 1000 + 212992]                            # group stride 212992 columns
    for c in cols:
        big[:, c:c + 256] = torch.randint(-3, 4, (M, 256), device=dev, generator=g).to(BF)
    X = torch.randint(-3, 4, (M, 2 * 128), device=dev, generator=g).to(BF)
    out = torch.zeros(G, Nn, Kk, device=dev)
    ops.call("gemm_tn_cols", big[:, 1000:], ld, X, 256, out, Kk, M, Nn, Kk, G, 212992, 128, Nn * Kk, 0, 0)
    torch.cuda.synchronize()
    for q, c in enumerate(cols):
        ref = big[:, c:c + Nn].float().t() @ X[:, q * 128: q * 128 + Kk].float()
        assert torch.equal(out[q], ref)


def test_caption_shorter_than_its_class():
    """A 5-word caption stored with 48 rows (class 3): the score GEMM masks every padding tile, the pair kernel every padding word."""
    from medmoe_amd import ops
    torch.manual_seed(2)
    B, HW, T, D, ntt = 2, 196, 77, 768, 3
    caps = [5, 40]
    ctx = bf_round(torch.randn(B, HW, D) * 0.2); words = bf_round(torch.randn(B, T, D) * 0.2)
    img_l = ctx.transpose(1, 2).reshape(B, D, 14, 14)
    sim_ref, _ = O.gloria_local_sim(img_l, words.transpose(1, 2), caps, 4.0, 5.0)
    dev = "cuda"
    HWp, Tp, _ = ops.local_geometry(HW, T)
    c16 = ctx.to(dev).to(BF).reshape(B * HW, D).contiguous(); w16 = words.to(dev).to(BF).contiguous()
    capd = torch.tensor(caps, dtype=I32, device=dev)
    rows = B * 16 * ntt
    ld, bs = HWp, rows * HWp
    lpT = torch.full((B * rows * HWp,), float("nan"), device=dev, dtype=BF)
    lse = torch.full((B, B, HWp), float("nan"), device=dev)
    members = torch.arange(B, device=dev, dtype=I32)
    ops.call("local_scores_t", c16, w16, capd, lpT, lse, B, B, HW, T, D, members, B, ntt, 0, ld, bs)
    tiles = lpT.view(torch.float16).view(B, rows, HWp).float().cpu()             # [image][row][region]
    s_all = torch.einsum("bhd,itd->biht", ctx, words)
    for i, cap in enumerate(caps):
        blk = tiles[:, i * 16 * ntt:(i + 1) * 16 * ntt, :HW]                   # [b][t][hw]
        ref = torch.log_softmax(s_all[:, i, :, :cap], dim=-1) * 1.4426950408889634
        assert torch.allclose(blk[:, :cap].permute(0, 2, 1), ref, atol=4e-3, rtol=2e-3)
        assert bool((blk[:, cap:] < -5e4).all())
    wn = torch.empty(B, T, device=dev); wT = torch.zeros(D, 64 * ((B * 16 * ntt + 63) // 64), device=dev, dtype=BF)
    ops.call("words_prep", w16, wn, torch.empty(D, B * Tp, device=dev, dtype=BF), B, T, Tp, D)
    gm = plain_gram(c16, B, HW, D, 224)
    sim = torch.full((B, B), float("nan"), device=dev); AT = torch.empty_like(lpT); stats = torch.empty(B, rows, 2, device=dev)
    ops.call("local_pair3", lpT, None, AT, None, lse, gm, wn, capd, None, sim, None, stats, rows, B, B, HW, T, 4.0, 5.0, 1e-8, members, B, ntt, 0, ld, bs, HWp, None)
    torch.cuda.synchronize()
    assert torch.allclose(sim.cpu(), sim_ref, atol=3e-2, rtol=1e-2), (sim.cpu() - sim_ref).abs().max()


@pytest.mark.parametrize("B, Kp, HWq, cap", [(3, 96, 208, 96), (5, 1024, 208, 1100), (4, 2048, 64, 2048), (2, 32, 208, 40)])
def test_weighted_gram_gemm_matches_two_operand_form(B, Kp, HWq, cap):
    """medmoe_gemm_tn_gram (dGm_b = A_b^T diag(w_b) A_b, the weights applied to the fragments inside the GEMM) against medmoe_gemm_tn_cols
    on the explicitly stored U = bf16(w * A): the same bf16 products in the same order -> bit-identical fp32 sums; and against fp32 torch."""
    from medmoe_amd import ops
    dev = "cuda"
    g = torch.Generator().manual_seed(B * 1000 + Kp)
    A = (torch.randn(B, Kp, HWq, generator=g) * 0.5).to(BF).to(dev)
    w = torch.full((B, cap), float("nan"), device=dev)
    w[:, :Kp] = (torch.randn(B, Kp, generator=g) * 2.0).to(dev)
    U = (A.float() * w[:, :Kp, None]).to(BF).contiguous()
    ld, bs = HWq, Kp * HWq
    out1 = torch.zeros(B, HWq, HWq, device=dev); out2 = torch.zeros_like(out1)
    ops.call("gemm_tn_gram", A, ld, w, cap, 1, out1, HWq, Kp, HWq, B, bs, HWq * HWq)
    ops.call("gemm_tn_cols", U, ld, A, ld, out2, HWq, Kp, HWq, HWq, B, bs, bs, HWq * HWq, 0, 0)
    torch.cuda.synchronize()
    ref = torch.bmm(U.float().transpose(1, 2), A.float())
    assert rel(out2, ref) < 1e-5
    assert torch.equal(out1, out2), float((out1 - out2).abs().max())


def test_src_losses_takes_the_transposed_path_at_196_regions():
    """src.losses.GLORIALocalContrastiveLoss (the reference-named API a LightningModule calls) at the ViT-B/16 geometry - 196 regions, 77
    word slots, ragged captions over all five length classes - runs TransposedLocalLoss.standalone: both losses, the attention maps of
    the matching pairs and the image-side gradient against the oracle; a second call reuses the cached buffers and reproduces itself."""
    import src.losses as L
    caps = [77, 8, 40, 23, 50, 64, 16, 33, 1]
    B, D, Hh, T = len(caps), 768, 14, 77
    g = torch.Generator().manual_seed(3)
    img = (torch.randn(B, D, Hh, Hh, generator=g) * 0.2).to(BF).float()
    words = (torch.randn(B, D, T, generator=g) * 0.2).to(BF).float()
    xr = img.clone().requires_grad_(True)
    l0r, l1r, maps_r = O.gloria_local(xr, words, caps, 4.0, 5.0, 10.0)
    (l0r + 2.0 * l1r).backward()
    outs = []
    for _ in range(2):
        x = img.cuda().requires_grad_(True)
        o = L.GLORIALocalContrastiveLoss()(x, words.cuda(), caps, temp1=4.0, temp2=5.0, temp3=10.0)
        (o.loss0 + 2.0 * o.loss1).backward()
        outs.append((o, x.grad))
    assert len(L._TL_CACHE) == 1
    o, gx = outs[0]
    assert abs(o.loss0.item() - l0r.item()) < 3e-2 * max(1.0, abs(l0r.item())) and abs(o.loss1.item() - l1r.item()) < 3e-2 * max(1.0, abs(l1r.item()))
    assert rel(gx.cpu(), xr.grad) < 5e-2, rel(gx.cpu(), xr.grad)
    for i in range(B):
        assert tuple(o.att_maps[i].shape) == (1, caps[i], Hh, Hh)
        assert rel(o.att_maps[i].cpu(), maps_r[i].detach()) < 4e-2, i
    assert abs(outs[0][0].loss0.item() - outs[1][0].loss0.item()) < 1e-5 * abs(outs[0][0].loss0.item())      # row losses meet in fp32 atomics
    assert rel(outs[1][1], outs[0][1]) < 2e-3          # fp32 atomics feed bf16 casts
    # two forwards before a backward: the first one's pair matrices are gone - refused, not silently wrong
    xa = img.cuda().requires_grad_(True)
    oa = L.GLORIALocalContrastiveLoss()(xa, words.cuda(), caps)
    L.GLORIALocalContrastiveLoss()(img.cuda().requires_grad_(True), words.cuda(), caps)
    with pytest.raises(RuntimeError, match="earlier forward"):
        oa.loss0.backward()


def test_word_embedding_gradient_all_length_classes_against_the_oracle():
    """d loss / d words of src.losses.GLORIALocalContrastiveLoss (the reference differentiates the word embeddings, losses.py:985-1012; needed
    once the text tower trains) at the real geometry - 196 regions, 77 words, width 768, one caption in every length class, an odd batch
    (the row-major pair matrices' pitch is then padded to the GEMM k-step): against the oracle's autograd, rel-L2 5e-2 (the bar of the
    region-feature gradient of this path); the region-feature gradient itself is unchanged by the word-gradient mode (2e-3: fp32 atomics)."""
    import src.losses as L
    caps = [77, 8, 40, 23, 50, 64, 16, 33, 1]
    B, D, Hh, T = len(caps), 768, 14, 77
    g = torch.Generator().manual_seed(3)
    img = (torch.randn(B, D, Hh, Hh, generator=g) * 0.2).to(BF).float()
    words = (torch.randn(B, D, T, generator=g) * 0.2).to(BF).float()
    xr, wr = img.clone().requires_grad_(True), words.clone().requires_grad_(True)
    l0r, l1r, _ = O.gloria_local(xr, wr, caps, 4.0, 5.0, 10.0)
    (l0r + 2.0 * l1r).backward()
    x, w = img.cuda().requires_grad_(True), words.cuda().requires_grad_(True)
    o = L.GLORIALocalContrastiveLoss()(x, w, caps, temp1=4.0, temp2=5.0, temp3=10.0)
    (o.loss0 + 2.0 * o.loss1).backward()
    torch.cuda.synchronize()
    e_x, e_w = rel(x.grad.cpu(), xr.grad), rel(w.grad.cpu(), wr.grad)
    print(f"word-gradient mode: d img_l {e_x:.4f}  d words {e_w:.4f}")
    assert abs(o.loss0.item() - l0r.item()) < 3e-2 * max(1.0, abs(l0r.item()))
    assert e_x < 5e-2 and e_w < 5e-2
    for i, n in enumerate(caps):                       # words at or beyond a caption's length: exactly no gradient, as in the reference
        if n < T:
            assert float(w.grad[i, :, n:].abs().max()) == 0.0 and float(wr.grad[i, :, n:].abs().max()) == 0.0
    x2 = img.cuda().requires_grad_(True)
    o2 = L.GLORIALocalContrastiveLoss()(x2, words.cuda(), caps, temp1=4.0, temp2=5.0, temp3=10.0)
    (o2.loss0 + 2.0 * o2.loss1).backward()
    assert rel(x.grad, x2.grad) < 2e-3
    # geometries without the transposed kernels refuse instead of returning no gradient
    with pytest.raises(NotImplementedError):
        L.GLORIALocalContrastiveLoss()(torch.randn(2, 128, 3, 3, device="cuda"), torch.randn(2, 128, 16, device="cuda").requires_grad_(True), [3, 5])


@pytest.mark.parametrize("geom", [(8, 16), (14, 77), (3, 16)])
def test_local_loss_agg_mean_shifts_every_caption_column_by_log_of_its_length(geom):
    """agg = 'mean' (losses.py:1006-1009: row_sim.mean over the caption's words before the log): the similarity of caption j drops by
    log(cap_len_j) against agg = 'sum' - through all three code paths of src.losses (64 / 196 regions: transposed kernels; 9 regions: the
    uniform-layout pair kernel) against the oracle's formula, losses 1e-2 and the region-feature gradient 5e-2."""
    import src.losses as L
    side, T = geom
    caps = [T, 3, 9, 1, max(1, T - 4), 7][: 6]
    B, D = len(caps), 128
    g = torch.Generator().manual_seed(5)
    img = (torch.randn(B, D, side, side, generator=g) * 0.3).to(BF).float()
    words = (torch.randn(B, D, T, generator=g) * 0.3).to(BF).float()
    xr = img.clone().requires_grad_(True)
    sim, _ = O.gloria_local_sim(xr, words, caps, 4.0, 5.0)
    sim = (sim - torch.log(torch.tensor(caps, dtype=torch.float32))[None, :]) * 10.0
    lab = torch.arange(B)
    l0r, l1r = torch.nn.functional.cross_entropy(sim, lab), torch.nn.functional.cross_entropy(sim.t(), lab)
    (l0r + l1r).backward()
    x = img.cuda().requires_grad_(True)
    o = L.GLORIALocalContrastiveLoss()(x, words.cuda(), caps, temp1=4.0, temp2=5.0, temp3=10.0, agg="mean")
    (o.loss0 + o.loss1).backward()
    assert abs(o.loss0.item() - l0r.item()) < 1e-2 * max(1.0, abs(l0r.item())) and abs(o.loss1.item() - l1r.item()) < 1e-2 * max(1.0, abs(l1r.item()))
    assert rel(x.grad.cpu(), xr.grad) < 5e-2
    o_sum = L.GLORIALocalContrastiveLoss()(img.cuda(), words.cuda(), caps, temp1=4.0, temp2=5.0, temp3=10.0)
    assert abs(o_sum.loss0.item() - o.loss0.item()) > 1e-3                       # the two aggregations differ on ragged captions
    with pytest.raises(ValueError):
        L.GLORIALocalContrastiveLoss()(img.cuda(), words.cuda(), caps, agg="max")
