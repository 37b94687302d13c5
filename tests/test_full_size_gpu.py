"""Full-size (BASELINE.json configs[2]: ViT-B/16, 8 experts top-2, batch 1024) checks of the hot path through properties that
do not need the CPU oracle (which would take hours at this size):
  * repeatability: the same batch twice gives bit-identical routing (indices and probabilities); the loss sums and the wgrads meet
    in fp32 atomics, so the losses agree to 1e-6 and the gradients (bf16 casts downstream of atomic sums) to 2e-3 of their norm;
  * linearity: loss_scale = 2 doubles every gradient (within the same run-to-run noise) and the reported losses;
  * permutation invariance: the contrastive losses do not depend on the order of the pairs in the batch, and every sample keeps
    its own router decision (bit-exact indices) wherever it sits;
  * a 32 x 32 sample of the 1024 x 1024 local-loss similarities against the oracle's GLoRIA attention on the same features;
  * simplex checks: router probabilities sum to one, indices are distinct and in range.
Tolerances are stated at each assert."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


def test_cfg2_global_batch_1024_properties():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    import bench
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    cfg = config_by_name("cfg2")
    B = 1024
    eng = Engine(cfg, "cuda:0", seed=0)
    batch = bench.synthetic_batch(cfg, B, 4321, eng.device)

    def run(bt, scale=1.0):
        out = eng.train_step(bt, optimizer=False, loss_scale=scale)
        torch.cuda.synchronize()
        o = eng.outputs()
        return ({k: float(v) for k, v in out.items()}, eng.params.g32.clone(), o["idx"].clone(), o["probs"].clone())

    l1, g1, idx1, pr1 = run(batch)
    l2, g2, idx2, pr2 = run(batch)
    # repeatability: forward has no atomics -> identical losses / routing; the wgrads meet in fp32 atomics (order-dependent last bits)
    for k in l1:            # every loss is a sum over 2048 rows / columns that meets in one fp32 atomic: the order of arrival moves the last
        assert abs(l1[k] - l2[k]) <= 5e-6 * max(1.0, abs(l1[k])), k      # bits (seen: 11 ulp of 19.0 once in ~ten runs, 0-3 ulp otherwise)
    assert torch.equal(idx1, idx2) and torch.equal(pr1, pr2)
    # fp32 atomics (wgrad partial sums, the local-loss context gradient) feed bf16 casts: a last-bit difference can flip a bf16
    # rounding upstream of the rest of the backward, so two runs agree to ~3e-4 of the gradient norm, not to fp32 precision
    assert _rel(g2, g1) < 2e-3
    for v in l1.values():
        assert v == v and abs(v) < 1e4          # finite

    # the local-loss similarity of 32 x 32 randomly chosen (image, caption) pairs of the 1024 x 1024 the step computed, against
    # the oracle's GLoRIA attention (CPU, fp32) on the engine's own bf16 expert features / word embeddings: exercises the score
    # GEMM, the ragged class layout and the pair kernel at full size (same tolerance as the small-batch oracle test)
    import oracle.medmoe_oracle as O
    gi = torch.Generator().manual_seed(3)
    rows = torch.randperm(B, generator=gi)[:32]; cols = torch.randperm(B, generator=gi)[:32]
    P, Hh = cfg.n_patch, int(cfg.n_patch ** 0.5)
    img_l = eng.ws["img_l"][rows.to(eng.device)].float().cpu().transpose(1, 2).reshape(32, cfg.d_out, Hh, Hh)
    words = eng.ws["words"][cols.to(eng.device)].float().cpu().transpose(1, 2)               # [32, D, T]
    caps = eng.outputs()["cap_lens"].cpu()[cols].tolist()
    sim_ref, _ = O.gloria_local_sim(img_l, words, caps, cfg.temp1, cfg.temp2)
    sim_got = eng.ws["sim"].cpu()[rows][:, cols]
    assert torch.allclose(sim_got, sim_ref, atol=3e-2, rtol=1e-2), float((sim_got - sim_ref).abs().max())

    # the global contrastive loss and the router CE of the full batch from the oracle's formulas on the engine's own embeddings /
    # probabilities (fp32 on both sides: 1e-4 relative)
    g_ref = float(O.gloria_global(eng.ws["img_g"].float().cpu(), eng.ws["txt_g"].float().cpu(), cfg.temp3))
    c_ref = float(O.router_ce(pr1.float().cpu(), batch["label"].cpu()))
    assert abs(l1["g_loss"] - g_ref) <= 1e-4 * max(1.0, abs(g_ref)), (l1["g_loss"], g_ref)
    assert abs(l1["classifier_loss"] - c_ref) <= 1e-4 * max(1.0, abs(c_ref)), (l1["classifier_loss"], c_ref)
    # ... and the local loss from the engine's full 1024 x 1024 similarity matrix
    simf = eng.ws["sim"].float().cpu() * cfg.temp3
    lab = torch.arange(B)
    l_ref = float(torch.nn.functional.cross_entropy(simf, lab) + torch.nn.functional.cross_entropy(simf.t(), lab))
    assert abs(l1["l_loss"] - l_ref) <= 1e-4 * max(1.0, abs(l_ref)), (l1["l_loss"], l_ref)

    # simplex checks
    assert float((pr1.sum(1) - 1).abs().max()) < 1e-5
    srt = idx1.long().sort(1).values
    assert int(idx1.min()) >= 0 and int(idx1.max()) < cfg.n_expert and bool((srt[:, 1:] != srt[:, :-1]).all())

    # linearity in the loss scale (x2 is exact in bf16 / fp32 up to the atomic ordering above)
    l3, g3, _, _ = run(batch, 2.0)
    assert _rel(g3, 2 * g1) < 2e-3
    for k in ("loss", "g_loss", "l_loss", "classifier_loss"):
        assert abs(l3[k] - 2 * l1[k]) <= 1e-5 * max(1.0, abs(2 * l1[k])), k

    # permutation invariance of the batch
    perm = torch.randperm(B, device=eng.device, generator=torch.Generator(device=eng.device).manual_seed(7))
    pb = {k: v[perm].contiguous() for k, v in batch.items()}
    l4, g4, idx4, pr4 = run(pb)
    assert torch.equal(idx4, idx1[perm])                      # each sample's router decision is its own: bit-exact
    assert torch.equal(pr4, pr1[perm])
    # sums over pairs in a different order (and a different ragged class layout): fp32 reassociation only
    for k in ("g_loss", "l_loss", "classifier_loss", "loss"):
        assert abs(l4[k] - l1[k]) <= 2e-3 * max(1.0, abs(l1[k])), (k, l4[k], l1[k])
    assert abs(l4["classifier_acc"] - l1["classifier_acc"]) < 1e-6
    assert _rel(g4, g1) < 2e-2                                # bf16 activations summed in another order


def test_full_size_gemms_against_fp32_matmul():
    """The step's largest Linear shapes (201728 tokens = 1024 x 197) on the four-wave kernels against torch's fp32 matmul of the
    same bf16 operands: forward with bias + GELU (FC1), dgrad-shaped plain product, and the wgrad with its bias gradient.
    Tolerances: one bf16 rounding of the output for the NT products (4e-3 of the norm), fp32 accumulation for the wgrad (1e-5)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from medmoe_amd import ops
    torch.manual_seed(11)
    bf = torch.bfloat16
    M, K, N = 201728, 768, 3072
    a = (torch.randn(M, K, device="cuda") * 0.5).to(bf)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    bias = torch.randn(N, device="cuda") * 0.1
    h = torch.empty(M, N, device="cuda", dtype=bf); d = torch.empty(M, N, device="cuda", dtype=bf)
    ops.gemm_nt(a, w, h, bias=bias, aux=d, epi=ops.EPI_GELU_DAUX)
    # fp32 reference in row blocks (the whole fp32 product would be 2.5 GB per tensor: fine, but blocks keep the peak low)
    errs_h, errs_d = [], []
    for r0 in range(0, M, 50432):
        z = a[r0:r0 + 50432].float() @ w.float().t() + bias
        zz = z.detach().clone().requires_grad_(True)
        g = torch.nn.functional.gelu(zz)
        g.sum().backward()
        errs_h.append((_rel(h[r0:r0 + 50432], g.detach()), g.detach().norm().item()))
        errs_d.append((_rel(d[r0:r0 + 50432], zz.grad), zz.grad.norm().item()))
    assert max(e for e, _ in errs_h) < 4e-3 and max(e for e, _ in errs_d) < 4e-3
    # plain product with the transposed weight (the dgrad shape: N = 768, K = 3072)
    wt = (torch.randn(K, N, device="cuda") * 0.05).to(bf)
    dx = torch.empty(M, K, device="cuda", dtype=bf)
    ops.gemm_nt(h, wt, dx)
    for r0 in (0, 100864, M - 50432):
        ref = h[r0:r0 + 50432].float() @ wt.float().t()
        assert _rel(dx[r0:r0 + 50432], ref) < 4e-3
    # wgrad + bias grad
    dw = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    ops.gemm_tn(d, a, dw, db=db)
    ref_w = torch.zeros(N, K, device="cuda"); ref_b = torch.zeros(N, device="cuda")
    for r0 in range(0, M, 50432):
        ref_w += d[r0:r0 + 50432].float().t() @ a[r0:r0 + 50432].float()
        ref_b += d[r0:r0 + 50432].float().sum(0)
    assert _rel(dw, ref_w) < 1e-5 and _rel(db, ref_b) < 1e-5
