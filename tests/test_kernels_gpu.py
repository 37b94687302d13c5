"""GPU parity tests for the individual HIP kernels, each against a plain torch fp32
reference of the same op on the same (bf16-rounded) inputs.  Tolerances are stated per test."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from medmoe_amd import ops as o
    return o


def bf(t):
    return t.to(torch.bfloat16)


def rel_err(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_gemm_nt_identity_asymmetric(ops):
    """A = I with an ASYMMETRIC B catches swapped row/col maps (cdna guide section 3). Exact."""
    dev = "cuda"
    K = 128
    a = torch.eye(K, device=dev)
    b = torch.arange(K * K, device=dev, dtype=torch.float32).reshape(K, K) % 251 - 125   # exact in bf16
    out = torch.empty(K, K, device=dev, dtype=torch.float32)
    ops.gemm_nt(bf(a), bf(b), out)
    assert torch.equal(out, b.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (197 * 3, 768, 768), (100, 192, 128), (50432 // 8, 2304, 768),
                                   (1100, 192, 128), (2048, 128, 64), (1283, 132, 320),    # M >= 1024 takes the 256x128 3-stage kernel
                                   (1300, 1536, 128), (2049, 1032, 192), (4096, 3072, 768)])   # ... and N >= 1024 the 256x256 one
def test_gemm_nt_plain(ops, M, N, K):
    torch.manual_seed(0)
    a = bf(torch.randn(M, K, device="cuda")); b = bf(torch.randn(N, K, device="cuda"))
    ref = a.float() @ b.float().t()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(a, b, out)
    assert rel_err(out, ref) < 1e-5           # fp32 accumulate of exact bf16 products
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt(a, b, out16)
    assert rel_err(out16, ref) < 4e-3         # one bf16 rounding


@pytest.mark.parametrize("M,N", [(300, 256), (1300, 256), (1300, 1280)])       # 128x128 / 256x128 / 256x256 kernel
def test_gemm_nt_epilogues(ops, M, N):
    torch.manual_seed(1)
    K = 128
    a = bf(torch.randn(M, K, device="cuda")); b = bf(torch.randn(N, K, device="cuda") * 0.1)
    bias = torch.randn(N, device="cuda"); res = bf(torch.randn(M, N, device="cuda"))
    z = a.float() @ b.float().t() + bias
    # gelu + aux
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); aux = torch.empty_like(out)
    ops.gemm_nt(a, b, out, bias=bias, aux=aux, epi=ops.EPI_GELU)
    assert rel_err(aux, z) < 4e-3 and rel_err(out, torch.nn.functional.gelu(z)) < 5e-3
    # relu + residual
    ops.gemm_nt(a, b, out, bias=bias, residual=res, epi=ops.EPI_RELU)
    assert rel_err(out, torch.relu(z) + res.float()) < 4e-3
    # (acc + residual) * gelu'(aux)
    zz = bf(torch.randn(M, N, device="cuda"))
    zf = zz.float()
    dg = 0.5 * (1 + torch.erf(zf / math.sqrt(2))) + zf * torch.exp(-0.5 * zf * zf) / math.sqrt(2 * math.pi)
    ops.gemm_nt(a, b, out, residual=res, aux=zz, epi=ops.EPI_MUL_DGELU, alpha=0.5)
    assert rel_err(out, (0.5 * (a.float() @ b.float().t()) + res.float()) * dg) < 5e-3
    ops.gemm_nt(a, b, out, aux=zz, epi=ops.EPI_MUL_DRELU)
    assert rel_err(out, (a.float() @ b.float().t()) * (zf > 0)) < 4e-3
    # the engine's pair: forward stores GELU'(z), backward multiplies by it
    ops.gemm_nt(a, b, out, bias=bias, aux=aux, epi=ops.EPI_GELU_DAUX)
    dgz = 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)
    assert rel_err(out, torch.nn.functional.gelu(z)) < 5e-3 and rel_err(aux, dgz) < 4e-3
    ops.gemm_nt(a, b, out, aux=zz, epi=ops.EPI_MUL_AUX)
    assert rel_err(out, (a.float() @ b.float().t()) * zf) < 4e-3


def test_gemm_nt_rowmaps_and_groups(ops):
    torch.manual_seed(2)
    K, N = 64, 128
    src = bf(torch.randn(500, K, device="cuda"))
    w = bf(torch.randn(3, N, K, device="cuda"))
    bias = torch.randn(3, N, device="cuda")
    # three groups with ragged row counts 130, 0, 77 -> tiles built on the host for the test
    counts = [130, 0, 77]
    rows = sum(counts)
    amap = torch.randperm(500, device="cuda")[:rows].int()
    cmap = torch.randperm(400, device="cuda")[:rows].int()
    tiles, start = [], 0
    for g, c in enumerate(counts):
        m = start
        while m < start + c:
            tiles.append([g, m, start + c, 0]); m += 128
        start += c
    tl = torch.tensor(tiles, device="cuda", dtype=torch.int32)
    cnt = torch.tensor([len(tiles)], device="cuda", dtype=torch.int32)
    out = torch.zeros(400, N, device="cuda", dtype=torch.float32)
    ops.gemm_nt(src, w, out, bias=bias, a_rowmap=amap, c_rowmap=cmap, tiles=tl, tile_count=cnt,
                max_tiles=len(tiles) + 3, stride_b=N * K, stride_bias=N, M=rows)
    ref = torch.zeros_like(out)
    start = 0
    for g, c in enumerate(counts):
        r = slice(start, start + c)
        ref[cmap[r].long()] = src[amap[r].long()].float() @ w[g].float().t() + bias[g]
        start += c
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("epi", ["none", "relu_bias", "drelu_res_aux"])
def test_gemm_nt_tiles256_groups(ops, epi):
    """Grouped + gathered + scattered GEMM on the 256x256 kernel (256-row tile table), ragged groups."""
    torch.manual_seed(7)
    K, N = 192, 320
    src = bf(torch.randn(2000, K, device="cuda"))
    w = bf(torch.randn(3, N, K, device="cuda") * 0.1)
    bias = torch.randn(3, N, device="cuda")
    counts = [700, 0, 390]
    rows = sum(counts)
    amap = torch.randperm(2000, device="cuda")[:rows].int()
    cmap = torch.randperm(1500, device="cuda")[:rows].int()
    tiles, start = [], 0
    for g, c in enumerate(counts):
        m = start
        while m < start + c:
            tiles.append([g, m, start + c, 0]); m += 256
        start += c
    tl = torch.tensor(tiles, device="cuda", dtype=torch.int32)
    cnt = torch.tensor([len(tiles)], device="cuda", dtype=torch.int32)
    out = torch.zeros(1500, N, device="cuda", dtype=torch.bfloat16)
    res = bf(torch.randn(1500, N, device="cuda")); aux = bf(torch.randn(1500, N, device="cuda"))
    kw = dict(a_rowmap=amap, c_rowmap=cmap, tiles=tl, tile_count=cnt, max_tiles=len(tiles) + 2, stride_b=N * K, M=rows, tile_rows=256)
    if epi == "none":
        ops.gemm_nt(src, w, out, **kw)
    elif epi == "relu_bias":
        ops.gemm_nt(src, w, out, bias=bias, stride_bias=N, epi=ops.EPI_RELU, **kw)
    else:
        ops.gemm_nt(src, w, out, residual=res, aux=aux, epi=ops.EPI_MUL_DRELU, **kw)
    ref = torch.zeros(1500, N, device="cuda")
    start = 0
    for g, c in enumerate(counts):
        r = slice(start, start + c)
        z = src[amap[r].long()].float() @ w[g].float().t()
        cr = cmap[r].long()
        if epi == "relu_bias":
            z = torch.relu(z + bias[g])
        elif epi == "drelu_res_aux":
            z = (z + res[cr].float()) * (aux[cr].float() > 0)
        ref[cr] = z
        start += c
    assert rel_err(out, ref) < 4e-3


@pytest.mark.parametrize("M,Nn,Kk,nsplit", [(64, 128, 128, 1), (1000, 192, 320, 4), (197 * 16, 768, 768, 8),
                                            (8192, 512, 256, 8), (6400, 768, 1024, 8), (197 * 64, 256, 768, 8)])   # the last three: 256x256 kernel
def test_gemm_tn(ops, M, Nn, Kk, nsplit):
    torch.manual_seed(3)
    g = bf(torch.randn(M, Nn, device="cuda")); x = bf(torch.randn(M, Kk, device="cuda"))
    dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
    ops.gemm_tn(g, x, dw, db=db, nsplit=nsplit)
    assert rel_err(dw, g.float().t() @ x.float()) < 1e-5
    assert rel_err(db, g.float().sum(0)) < 1e-5
    ops.gemm_tn(g, x, dw, db=db, nsplit=nsplit)      # accumulates
    assert rel_err(dw, 2 * (g.float().t() @ x.float())) < 1e-5


def test_gemm_tn_exact_integer(ops):
    """Exact small-integer data: any k-order / transposed-read mix-up shows as a wrong integer."""
    M, Nn, Kk = 192, 128, 128
    g = (torch.arange(M * Nn, device="cuda").reshape(M, Nn) % 7 - 3).float()
    x = (torch.arange(M * Kk, device="cuda").reshape(M, Kk) % 5 - 2).float()
    dw = torch.zeros(Nn, Kk, device="cuda")
    ops.gemm_tn(bf(g), bf(x), dw, nsplit=2)
    assert torch.equal(dw, g.t() @ x)


def test_gemm_tn512_exact_integer(ops):
    """The 256x256 wgrad kernel (LDS-DMA ring + inline-asm transposed reads) on exact small-integer data: a fragment
    read before its data arrived, or a k-order mix-up, shows as a wrong integer.  Two launches (the second accumulates)."""
    M, Nn, Kk = 8192, 256, 512
    g = (torch.arange(M * Nn, device="cuda").reshape(M, Nn) % 7 - 3).float()
    x = (torch.arange(M * Kk, device="cuda").reshape(M, Kk) % 5 - 2).float()
    dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
    ops.gemm_tn(bf(g), bf(x), dw, db=db)
    ref = g.double().t() @ x.double()
    assert torch.equal(dw.double(), ref) and torch.equal(db.double(), g.double().sum(0))
    ops.gemm_tn(bf(g), bf(x), dw, db=db)
    assert torch.equal(dw.double(), 2 * ref)


@pytest.mark.parametrize("four_wave", [1, 0])
def test_gemm_tn_256_tile_generations_exact(ops, four_wave):
    """gemm_tn4w (four waves, default) and gemm_tn512 (eight waves, option 8 = 0) on exact small-integer data, odd and even
    numbers of 32-row sub-steps per range, with the bias gradient."""
    ops.set_option(8, four_wave)
    try:
        for M in (4096, 8192 + 32 * 5, 16384 + 32):
            Nn, Kk = 512, 256
            g = (torch.arange(M * Nn, device="cuda").reshape(M, Nn) % 7 - 3).float()
            x = (torch.arange(M * Kk, device="cuda").reshape(M, Kk) % 5 - 2).float()
            dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
            ops.gemm_tn(bf(g), bf(x), dw, db=db)
            assert torch.equal(dw.double(), g.double().t() @ x.double()), M
            assert torch.equal(db.double(), g.double().sum(0)), M
    finally:
        ops.set_option(8, 1)


@pytest.mark.parametrize("epi", ["none", "bias_res", "gelu_daux", "mul_aux", "f32"])
def test_gemm_nt_256_tile_generations_bit_equal(ops, epi):
    """gemm_nt4w (four waves, default) and gemm_nt512 (eight waves, option 7 = 0) share tile order, accumulation order and
    epilogue code: their results must be BIT-equal, ragged M / N edges included; and match torch fp32 within one bf16 rounding."""
    torch.manual_seed(5)
    M, N, K = 1283, 1048, 320
    a = bf(torch.randn(M, K, device="cuda")); b = bf(torch.randn(N, K, device="cuda") * 0.1)
    bias = torch.randn(N, device="cuda"); res = bf(torch.randn(M, N, device="cuda")); auxin = bf(torch.randn(M, N, device="cuda"))
    outs = []
    for four_wave in (1, 0):
        ops.set_option(7, four_wave)
        try:
            c = torch.empty(M, N, device="cuda", dtype=torch.float32 if epi == "f32" else torch.bfloat16)
            aux = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            if epi == "none" or epi == "f32":
                ops.gemm_nt(a, b, c)
            elif epi == "bias_res":
                ops.gemm_nt(a, b, c, bias=bias, residual=res)
            elif epi == "gelu_daux":
                ops.gemm_nt(a, b, c, bias=bias, aux=aux, epi=ops.EPI_GELU_DAUX)
            else:
                aux = auxin.clone()
                ops.gemm_nt(a, b, c, aux=aux, epi=ops.EPI_MUL_AUX)
            torch.cuda.synchronize()
            outs.append((c.clone(), aux.clone()))
        finally:
            ops.set_option(7, 1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    z = a.float() @ b.float().t()
    ref = {"none": z, "f32": z, "bias_res": z + bias + res.float(), "gelu_daux": torch.nn.functional.gelu(z + bias),
           "mul_aux": z * auxin.float()}[epi]
    assert rel_err(outs[0][0], ref) < (1e-5 if epi == "f32" else 5e-3)


def test_gemm_tn_groups_rowmap(ops):
    torch.manual_seed(4)
    Nn, Kk = 128, 64
    g = bf(torch.randn(300, Nn, device="cuda")); xs = bf(torch.randn(700, Kk, device="cuda"))
    xmap = torch.randperm(700, device="cuda")[:300].int()
    off = torch.tensor([0, 100, 100, 300], device="cuda", dtype=torch.int32)
    dw = torch.zeros(3, Nn, Kk, device="cuda"); db = torch.zeros(3, Nn, device="cuda")
    ops.gemm_tn(g, xs, dw, db=db, x_rowmap=xmap, row_off=off, n_groups=3, stride_w=Nn * Kk, stride_db=Nn, nsplit=3)
    for i, (a, b) in enumerate([(0, 100), (100, 100), (100, 300)]):
        ref = g[a:b].float().t() @ xs[xmap[a:b].long()].float()
        assert rel_err(dw[i], ref) < 1e-5 if b > a else dw[i].abs().max() == 0
        assert torch.allclose(db[i], g[a:b].float().sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mapped", ["x", "g", "none"])
def test_gemm_tn512_groups_rowmap(ops, mapped):
    """Grouped / row-mapped wgrad on the 256x256 kernel: ragged group sizes (not multiples of 32), Nn = 1.5 tiles."""
    torch.manual_seed(8)
    Nn, Kk = 384, 512
    counts = [30000, 4099, 0, 6213]        # the first group is longer than the 8192-entry row-map ring of one workgroup
    M = sum(counts)
    bounds = [0]
    for c in counts: bounds.append(bounds[-1] + c)
    gs = bf(torch.randn(M + 900, Nn, device="cuda")); xs = bf(torch.randn(M + 500, Kk, device="cuda"))
    gmap = torch.randperm(M + 900, device="cuda")[:M].int() if mapped == "g" else None
    xmap = torch.randperm(M + 500, device="cuda")[:M].int() if mapped == "x" else None
    off = torch.tensor(bounds, device="cuda", dtype=torch.int32)
    G = len(counts)
    dw = torch.zeros(G, Nn, Kk, device="cuda"); db = torch.zeros(G, Nn, device="cuda")
    ops.gemm_tn(gs, xs, dw, db=db, x_rowmap=xmap, g_rowmap=gmap, row_off=off, n_groups=G, stride_w=Nn * Kk, stride_db=Nn, nsplit=3, M=M)
    for i in range(G):
        a, b = bounds[i], bounds[i + 1]
        gi = gs[gmap[a:b].long()] if gmap is not None else gs[a:b]
        xi = xs[xmap[a:b].long()] if xmap is not None else xs[a:b]
        if b > a:
            assert rel_err(dw[i], gi.float().t() @ xi.float()) < 1e-5
            assert rel_err(db[i], gi.float().sum(0)) < 1e-5
        else:
            assert dw[i].abs().max() == 0 and db[i].abs().max() == 0


@pytest.mark.parametrize("rows,D", [(37, 64), (1000, 768), (513, 1024), (64, 192), (100353, 96), (25087, 192), (3, 96), (4099, 128), (6273, 384)])
def test_layernorm(ops, rows, D):
    torch.manual_seed(5)
    x = bf(torch.randn(rows, D, device="cuda") * 2 + 0.5)
    gam = torch.rand(D, device="cuda") + 0.5; bet = torch.randn(D, device="cuda") * 0.1
    y = torch.empty_like(x); mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    ops.layernorm_fwd(x, gam, bet, y, mean, rstd, 1e-6)
    xr = x.float().requires_grad_(True); gr = gam.clone().requires_grad_(True); br = bet.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    assert rel_err(y, ref) < 4e-3
    assert torch.allclose(mean, x.float().mean(1), atol=1e-5, rtol=1e-5)
    y32 = torch.empty(rows, D, device="cuda")
    ops.layernorm_fwd(x, gam, bet, y32, mean, rstd, 1e-6)
    assert rel_err(y32, ref) < 1e-5
    dy = bf(torch.randn(rows, D, device="cuda")); add = bf(torch.randn(rows, D, device="cuda"))
    ref.backward(dy.float())
    dx = torch.empty_like(x); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dy, x, mean, rstd, gam, dx, dg, db, add=add)
    assert rel_err(dx, xr.grad + add.float()) < 5e-3
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4


def _attn_ref(qkv, B, N, H, mask):
    D = H * 64
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) / 8.0
    if mask is not None:
        s = s.masked_fill(~mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    return (p @ v).transpose(1, 2).reshape(B, N, D), torch.logsumexp(s, -1)


@pytest.mark.parametrize("resident", [0, 1])
@pytest.mark.parametrize("B,N,H,masked", [(2, 197, 3, False), (3, 77, 2, True), (2, 16, 1, True), (1, 257, 2, False), (2, 65, 1, False),
                                          (2, 577, 2, False), (2, 300, 1, True), (1, 64, 1, False), (1, 129, 2, True)])
def test_attention_fwd_bwd(ops, B, N, H, masked, resident):
    """resident = 1: the round-1 kernels (whole key range in LDS, N <= 272); 0: the streaming kernels (any N, the default)."""
    if resident and N > 592:
        pytest.skip("resident kernels hold at most 592 keys")
    ops.set_option(11, resident)
    try:
        _attention_case(ops, B, N, H, masked)
    finally:
        ops.set_option(11, 1)


def _attention_case(ops, B, N, H, masked):
    torch.manual_seed(6)
    D = H * 64
    qkv = bf(torch.randn(B, N, 3 * D, device="cuda"))
    mask = None
    if masked:
        lens = torch.randint(1, N + 1, (B,), device="cuda")
        mask = (torch.arange(N, device="cuda")[None] < lens[:, None]).to(torch.uint8)
    out = torch.empty(B, N, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device="cuda")
    ops.attn_fwd(qkv, out, lse, mask, B, N, H)
    qr = qkv.float().requires_grad_(True)
    ref, lse_ref = _attn_ref(qr, B, N, H, mask)
    assert rel_err(out, ref) < 1e-2            # P rounded to bf16 before P.V
    assert torch.allclose(lse, lse_ref, atol=2e-3, rtol=1e-4)
    dout = bf(torch.randn(B, N, D, device="cuda"))
    ref.backward(dout.float())
    dqkv = torch.zeros_like(qkv); delta = torch.empty(B, H, N, device="cuda")
    ops.attn_bwd(qkv, out, dout, lse, mask, dqkv, delta, B, N, H)
    g = qr.grad
    if masked:   # gradients of padded key/value rows are exactly zero in both
        pass
    assert rel_err(dqkv, g) < 2e-2


def test_preprocess_bicubic_matches_pillow_bit_for_bit(ops):
    """The reference's preprocessing = HF image processor = PIL.Image.resize(BICUBIC) on uint8, x 1/255, (x - mean) / std
    (swin.py:131).  Pillow is importable here, so the resized uint8 image must match it BIT FOR BIT (down- and up-scaling, odd
    sizes), and the normalised bf16 output must be the bf16 rounding of the fp32 formula on those bytes."""
    from PIL import Image
    from medmoe_amd.data import IMAGENET_MEAN, IMAGENET_STD, preprocess_images_bicubic
    g = torch.Generator().manual_seed(5)
    sizes = [(300, 420), (224, 224), (97, 131), (512, 160), (33, 700), (225, 223)]
    imgs = []
    for (h, w) in sizes:                                     # smooth + noise: exercises the negative lobes and the clipping
        base = torch.rand(h // 8 + 2, w // 8 + 2, 3, generator=g)
        up = torch.nn.functional.interpolate(base.permute(2, 0, 1)[None], size=(h, w), mode="bilinear")[0].permute(1, 2, 0)
        imgs.append(((up * 300 - 20 + torch.randn(h, w, 3, generator=g) * 25).clamp(0, 255)).to(torch.uint8).contiguous())
    for S in (224, 336):
        out, u8 = preprocess_images_bicubic([im.cuda() for im in imgs], size=S, return_uint8=True)
        torch.cuda.synchronize()
        for b, im in enumerate(imgs):
            ref = torch.from_numpy(np.asarray(Image.fromarray(im.numpy()).resize((S, S), resample=Image.BICUBIC)))
            assert torch.equal(u8[b].cpu(), ref), (sizes[b], S, int((u8[b].cpu() != ref).sum()))
            want = ((ref.float() * (1.0 / 255.0) - torch.tensor(IMAGENET_MEAN)) / torch.tensor(IMAGENET_STD)).permute(2, 0, 1).to(torch.bfloat16)
            assert torch.equal(out[b].cpu(), want), (sizes[b], S)


def test_preprocess_matches_torch_interpolate(ops):
    """Device image preprocessing (uint8 HWC, any size -> bf16 [B,3,224,224]): pinned to torch's bilinear interpolate
    (align_corners=False) + rescale + normalise in fp32; the HF processor the reference calls is absent offline
    (parity with its PIL bicubic filter: unpinned)."""
    from medmoe_amd.data import preprocess_images, IMAGENET_MEAN, IMAGENET_STD
    torch.manual_seed(9)
    sizes = [(160, 320), (224, 224), (301, 187), (97, 512)]
    imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda") for h, w in sizes]
    out = preprocess_images(imgs, size=224, resample="bilinear")
    assert out.shape == (4, 3, 224, 224) and out.dtype == torch.bfloat16
    mean = torch.tensor(IMAGENET_MEAN, device="cuda")[:, None, None]; std = torch.tensor(IMAGENET_STD, device="cuda")[:, None, None]
    for b, im in enumerate(imgs):
        x = im.permute(2, 0, 1)[None].float()
        ref = torch.nn.functional.interpolate(x, size=(224, 224), mode="bilinear", align_corners=False)[0] / 255.0
        ref = (ref - mean) / std
        assert (out[b].float() - ref).abs().max() < 2e-2 and rel_err(out[b], ref) < 4e-3


def test_gemm_nt_random_shapes_all_kernels(ops):
    """Seeded random shapes over the three forward/dgrad kernels' dispatch ranges (128x128, 256x128, 256x256 tiles), ragged M and
    N, every epilogue family: bias / residual / GELU (+aux) / activation-derivative factors, bf16 and fp32 C."""
    import random
    rng = random.Random(1234)
    torch.manual_seed(11)
    for _ in range(24):
        M = rng.choice([rng.randint(64, 900), rng.randint(1024, 3000), rng.randint(3000, 9000)])
        N = rng.choice([64, 136, 256, 520, 768, 1032, 1536, 2304]) if rng.random() < 0.8 else 8 * rng.randint(8, 300)
        K = 64 * rng.randint(1, 12)
        a = bf(torch.randn(M, K, device="cuda")); b = bf(torch.randn(N, K, device="cuda") * 0.2)
        z = a.float() @ b.float().t()
        kind = rng.choice(["plain", "f32", "bias", "bias_res", "gelu_aux", "gelu_daux", "mul_aux", "relu_bias"])
        bias = torch.randn(N, device="cuda"); res = bf(torch.randn(M, N, device="cuda")); aux = bf(torch.randn(M, N, device="cuda"))
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if kind == "f32" else torch.bfloat16)
        if kind in ("plain", "f32"):
            ops.gemm_nt(a, b, out); ref = z
        elif kind == "bias":
            ops.gemm_nt(a, b, out, bias=bias); ref = z + bias
        elif kind == "bias_res":
            ops.gemm_nt(a, b, out, bias=bias, residual=res); ref = z + bias + res.float()
        elif kind == "gelu_aux":
            ax = torch.empty_like(aux); ops.gemm_nt(a, b, out, bias=bias, aux=ax, epi=ops.EPI_GELU)
            ref = torch.nn.functional.gelu(z + bias); assert rel_err(ax, z + bias) < 4e-3, (M, N, K, kind)
        elif kind == "gelu_daux":
            ax = torch.empty_like(aux); ops.gemm_nt(a, b, out, bias=bias, aux=ax, epi=ops.EPI_GELU_DAUX)
            zz = z + bias; ref = torch.nn.functional.gelu(zz)
            dg = 0.5 * (1 + torch.erf(zz / math.sqrt(2))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2 * math.pi)
            assert rel_err(ax, dg) < 5e-3, (M, N, K, kind)
        elif kind == "mul_aux":
            ops.gemm_nt(a, b, out, aux=aux, epi=ops.EPI_MUL_AUX); ref = z * aux.float()
        else:
            ops.gemm_nt(a, b, out, bias=bias, epi=ops.EPI_RELU); ref = torch.relu(z + bias)
        tol = 1e-5 if kind == "f32" else 5e-3
        assert rel_err(out, ref) < tol, (M, N, K, kind, rel_err(out, ref))


@pytest.mark.gpu
def test_nt4w_row_major_epilogues_soak_against_nt512():
    """tools/soak_nt_epilogues.py: 160 random full-tile cases of the gemm_nt4w builds with the row-major epilogue (bias + residual, residual,
    x stored GELU', bias + GELU + GELU'; K from 128 = two stages), launched back to back, bit-identical to gemm_nt512."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_nt_epilogues.py")], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("M,N,K", [(4100, 96, 96), (8192, 384, 96), (5003, 96, 288), (4096, 768, 96), (100352, 288, 96)])
def test_gemm_nt_direct_short_k(ops, M, N, K):
    """K % 64 == 32 (Swin-T stage 1: 96 / 288 channels) runs gemm_nt_direct_kernel (fragments straight from global memory, no LDS): every
    epilogue the tower uses against fp32 and against the 128x128 DMA kernel (medmoe_set_option(17, 0))."""
    torch.manual_seed(M + N + K)
    a = bf(torch.randn(M, K, device="cuda") * 0.5); w = bf(torch.randn(N, K, device="cuda") * 0.2)
    bias = torch.randn(N, device="cuda"); res = bf(torch.randn(M, N, device="cuda")); auxin = bf(torch.rand(M, N, device="cuda") + 0.5)
    z = a.float() @ w.float().t()

    def run(**kw):
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm_nt(a, w, out, **kw)
        return out
    cases = [(dict(), z), (dict(bias=bias), z + bias), (dict(bias=bias, residual=res), z + bias + res.float()),
             (dict(bias=bias, epi=ops.EPI_RELU), torch.relu(z + bias)), (dict(aux=auxin, epi=ops.EPI_MUL_AUX), z * auxin.float())]
    for kw, want in cases:
        got = run(**kw)
        assert ops.load_library().medmoe_last_gemm_nt_kernel() == 6, kw
        ops.set_option(17, 0)
        try:
            old = run(**kw)
            assert ops.load_library().medmoe_last_gemm_nt_kernel() == 0
        finally:
            ops.set_option(17, 1)
        assert rel_err(got, want) < 5e-3, kw
        assert rel_err(got, old.float()) < 3e-3, kw
    dg = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)                     # GELU with its derivative stored (FC1 forward)
    h = run(bias=bias, aux=dg, epi=ops.EPI_GELU_DAUX)
    zz = (z + bias).requires_grad_(True)
    g = torch.nn.functional.gelu(zz); g.sum().backward()
    assert rel_err(h, g.detach()) < 5e-3 and rel_err(dg, zz.grad) < 8e-3
