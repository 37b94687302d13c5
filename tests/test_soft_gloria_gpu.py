"""Soft-GLoRIA (reference losses.py:814-883 and 1111-1214, medmoe_module.py:258-296) on the GPU: the head kernel
against the oracle's literal loop, src.losses' two classes against the REFERENCE fixture tests/golden/soft_gloria.npz
(oracle/gen_golden_soft.py), Engine.train_step and the Lightning module with `soft_label: true` against the oracle's model_step."""
import os
import sys

import numpy as np
import pytest
import torch

import medmoe_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.mark.parametrize("B", [8, 41, 300])
def test_soft_head_kernel_rows_and_columns(B):
    """medmoe_soft_xent_strided over the rows and (accumulating) over the columns = the reference's double loop: loss and d loss / dx
    in fp32, rows with one .. many positives, a row whose positive is also a negative (t1 < t2 is legal), a row without negatives."""
    from medmoe_amd import ops
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, B, generator=g) * 1.5
    soft = torch.rand(B, B, generator=g)
    soft.fill_diagonal_(1.0)
    soft[1] = 0.9                                  # every column positive, none negative
    soft[1, 1] = 1.0
    # (B = 300: few positives per row, the oracle's double loop is Python)
    for t1, t2 in (((0.7, 0.4), (0.5, 0.6)) if B < 100 else ((0.97, 0.5), (0.93, 0.95))):
        xr = x.clone().requires_grad_(True)
        l0, l1 = O.soft_gloria_head(xr * 2.5, soft, t1, t2)
        (l0 + l1).backward()
        xd, sd = x.cuda(), soft.cuda()
        dx = torch.empty_like(xd); loss = torch.zeros(2, device="cuda")
        ops.call("soft_xent_strided", xd, dx, sd, B, B, B, 1, 2.5, t1, t2, 1.0 / B, 0, loss)
        ops.call("soft_xent_strided", xd, dx, sd, B, B, 1, B, 2.5, t1, t2, 1.0 / B, 1, loss[1:])
        torch.cuda.synchronize()
        assert abs(loss[0].item() - l0.item()) < 2e-5 * max(1.0, abs(l0.item())), (loss, l0)
        assert abs(loss[1].item() - l1.item()) < 2e-5 * max(1.0, abs(l1.item())), (loss, l1)
        assert rel(dx, xr.grad) < 2e-5


def test_soft_losses_reference_fixture(golden_dir):
    """src.losses.SoftGLORIAGlobalContrastiveLoss / SoftGLORIALocalContrastiveLoss on the inputs of the reference fixture: the global loss
    (fp32 kernels) to 1e-4 with both gradients; the local loss (bf16 MFMA path) at the bf16 bar with the image-side gradient and the maps."""
    from src.losses import SoftGLORIAGlobalContrastiveLoss, SoftGLORIALocalContrastiveLoss
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "soft_gloria.npz")).items()}
    thr = tuple(float(v) for v in z["thresholds"])
    a = z["a"].cuda().requires_grad_(True); t = z["t"].cuda().requires_grad_(True)
    g = SoftGLORIAGlobalContrastiveLoss()(a, t, temp3=10.0, idx=z["soft"].cuda(), probs=thr)
    g.backward()
    assert abs(g.item() - z["g_loss"].item()) < 1e-4 * max(1.0, abs(z["g_loss"].item()))
    assert rel(a.grad, z["grad_a"]) < 1e-4 and rel(t.grad, z["grad_t"]) < 1e-4
    il = z["img_l"].cuda().requires_grad_(True)
    caps = z["cap_lens"].tolist()
    o = SoftGLORIALocalContrastiveLoss()(il, z["words"].cuda(), caps, temp1=4.0, temp2=5.0, temp3=10.0, idx=z["soft"].cuda(), probs=thr)
    (o.loss0 + 2.0 * o.loss1).backward()
    assert abs(o.loss0.item() - z["loss0"].item()) < 3e-2 * max(1.0, abs(z["loss0"].item())), (o.loss0, z["loss0"])
    assert abs(o.loss1.item() - z["loss1"].item()) < 3e-2 * max(1.0, abs(z["loss1"].item())), (o.loss1, z["loss1"])
    assert rel(il.grad, z["grad_img_l"]) < 5e-2, rel(il.grad, z["grad_img_l"])
    for i in range(len(caps)):
        assert rel(o.att_maps[i], z[f"att{i}"]) < 4e-2, i
    # without idx the Soft classes refuse, the plain classes ignore idx as the reference's do
    with pytest.raises(ValueError):
        SoftGLORIAGlobalContrastiveLoss()(a, t)


def _thresholds(soft):
    """(t_pos, t_neg, margin): thresholds in the widest gaps of the sorted off-diagonal scores between their 55th..90th / 15th..50th percentiles, and
    half the narrower gap (no score within `margin` of a threshold)."""
    B = soft.shape[0]
    v = soft[~torch.eye(B, dtype=torch.bool)].sort().values
    gaps = v[1:] - v[:-1]
    n = len(gaps)

    def widest(q0, q1):
        lo, hi = int(q0 * n), int(q1 * n)
        return lo + int(gaps[lo:hi].argmax())
    ihi, ilo = widest(0.55, 0.9), widest(0.15, 0.5)
    return float((v[ihi] + v[ihi + 1]) / 2), float((v[ilo] + v[ilo + 1]) / 2), float(min(gaps[ilo], gaps[ihi]) / 2)


@pytest.mark.parametrize("cfg_name, B, seed", [("tiny2", 8, 5), ("cfg0", 16, 11)])
def test_engine_step_with_soft_labels_matches_oracle(cfg_name, B, seed):
    """Engine.train_step with cfg.soft_label: caption scores from the text tower's own [CLS] (the frozen tool model), positives /
    negatives by two thresholds placed in gaps of the oracle's scores (so bf16 noise cannot move a caption across), Soft-GLoRIA global +
    local + router CE: scores, losses and every gradient against the oracle's model_step."""
    from test_parity2_gpu import make, to_dev
    ocfg, cfg, p, batch, eng, vocab = make(cfg_name, B, seed=seed, images="struct")
    with torch.no_grad():
        soft_ref = O.text_soft_target(O.text_hidden_states(batch["ids"], batch["attn_mask"], batch["token_type"], p, ocfg)[-1])
    t0, t1, margin = _thresholds(soft_ref)
    assert margin > 3e-4, margin
    for c in (ocfg, cfg):
        c.soft_label, c.threshold0, c.threshold1 = True, t0, t1
    npos = (soft_ref > t0).sum(1); nneg = (soft_ref <= t1).sum(1)
    assert int(npos.max()) >= 3 and int(nneg.max()) >= 3 and int(npos.min()) >= 1
    po = {k: v.clone().requires_grad_(not k.startswith("text.")) for k, v in p.items()}
    ref = O.model_step(batch, po, ocfg, vocab)
    ref["loss"].backward()
    out = eng.train_step(to_dev(batch), optimizer=False)
    torch.cuda.synchronize()
    if not torch.equal(eng.outputs()["idx"].cpu().long(), ref["idx"]):
        pytest.skip("near-tie routing differs between bf16 and fp32 towers on this seed")
    soft = eng._soft.cpu()
    assert float((soft - soft_ref).abs().max()) < 2e-3, float((soft - soft_ref).abs().max())       # bf16 tower against fp32: 5e-4 seen
    assert torch.equal(soft > t0, soft_ref > t0) and torch.equal(soft <= t1, soft_ref <= t1)
    for k in ("g_loss", "l_loss", "loss"):
        assert abs(float(out[k]) - float(ref[k])) < 3e-2 * max(1.0, abs(float(ref[k]))), (k, float(out[k]), float(ref[k]))
    # the soft head is not the hard one: the same batch under the diagonal-label losses gives different values
    ocfg.soft_label = False
    with torch.no_grad():
        hard = O.model_step(batch, p, ocfg, vocab)
    assert abs(float(hard["g_loss"]) - float(ref["g_loss"])) > 0.1
    # the two soft losses in isolation: differentiated by the oracle at the engine's own tower outputs and the engine's own scores
    P, Do = cfg.n_patch, cfg.d_out
    Hh = int(P ** 0.5)
    x = eng.ws["img_l"].float().cpu().transpose(1, 2).reshape(B, Do, Hh, Hh).requires_grad_(True)
    xg = eng.ws["img_g"].float().cpu().requires_grad_(True)
    l0, l1, _ = O.soft_gloria_local(x, eng.ws["words"].float().cpu().transpose(1, 2), ref["cap_lens"], soft, (t0, t1),
                                    ocfg.temp1, ocfg.temp2, ocfg.temp3)
    (ocfg.w_local * (l0 + l1) + ocfg.w_global * O.soft_gloria_global(xg, eng.ws["txt_g"].float().cpu(), soft, (t0, t1), ocfg.temp3)).backward()
    e_l = rel(eng.ws["d_img_l"].float(), x.grad.reshape(B, Do, P).transpose(1, 2))
    e_g = rel(eng.ws["d_img_g"], xg.grad)
    assert e_l < 2e-2 and e_g < 1e-3, (e_l, e_g)
    if cfg_name == "tiny2":                       # the whole fp32 chain where the loss is well conditioned (tests/test_engine_gpu.py)
        got = eng.params.export_named(eng.params.g32)
        errs = {}
        for k, v in po.items():
            if k.startswith("text.") or v.grad is None or v.grad.norm() < 1e-7:
                continue
            errs[k] = rel(got[k].reshape(v.grad.shape), v.grad)
        bad = {k: e for k, e in errs.items() if e > 0.15}
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
        assert float(np.median(list(errs.values()))) < 6e-2, float(np.median(list(errs.values())))


def test_lightning_module_soft_label_tracks_engine():
    """MedMoEPretrainingLightningModule with `soft_label: true` and the Soft-GLoRIA classes (the reference's commented-in configuration,
    med-moe_pretraining.yaml:25-37): get_text_soft_target + both losses through torch autograd = the fused engine step."""
    from src.losses import SoftGLORIAGlobalContrastiveLoss, SoftGLORIALocalContrastiveLoss
    from src.models.components.med_moe import MedMoE
    from src.models.medmoe_module import MedMoEPretrainingLightningModule
    from test_parity2_gpu import make, to_dev
    B = 8
    ocfg, cfg, p, batch, eng, vocab = make("tiny2", B, seed=5, images="struct")
    with torch.no_grad():
        soft_ref = O.text_soft_target(O.text_hidden_states(batch["ids"], batch["attn_mask"], batch["token_type"], p, ocfg)[-1])
    t0, t1, _ = _thresholds(soft_ref)
    cfg.soft_label, cfg.threshold0, cfg.threshold1 = True, t0, t1
    model = MedMoE({"config_name": "tiny2"}, {})
    model.engine.params.load_named(p)
    loss_cfg = {"global_loss": SoftGLORIAGlobalContrastiveLoss(), "local_loss": SoftGLORIALocalContrastiveLoss(),
                "global_loss_weight": 0.5, "local_loss_weight": 0.5, "classifier_loss_weight": 2.0,
                "temp1": 4.0, "temp2": 5.0, "temp3": 10.0, "soft_label": True, "topk": 5, "threshold0": t0, "threshold1": t1}
    lit = MedMoEPretrainingLightningModule(model, loss_cfg)
    dev = {"image": batch["image"].cuda(), "label": batch["label"].cuda(),
           "caption": {"ids": batch["ids"].cuda(), "attn_mask": batch["attn_mask"].cuda(), "token_type": batch["token_type"].cuda()}}
    out = lit.model_step(dev)
    out["loss"].backward()
    e = eng.train_step(to_dev(batch), optimizer=False)
    torch.cuda.synchronize()
    for k in ("g_loss", "l_loss", "loss"):
        assert abs(float(out[k]) - float(e[k])) < 3e-3 * max(1.0, abs(float(e[k]))), (k, float(out[k]), float(e[k]))
    ge = eng.params.g32
    cos = float((model.weights.grad * ge).sum() / (model.weights.grad.norm() * ge.norm()))
    assert cos > 0.995, cos


def test_hard_negative_and_zero_losses_reference_fixture(golden_dir):
    """src.losses.HardNegativeContrastiveLoss against the reference fixture (loss and both input gradients, two margins) and on a batch
    larger than one workgroup pass (B = 300) against the oracle; the ZERO switches return zeros of the reference's container shape."""
    from src.losses import HardNegativeContrastiveLoss, ZEROGlobalContrastiveLoss, ZEROLocalContrastiveLoss
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, "hard_negative.npz")).items()}
    a = z["imgs"].cuda().requires_grad_(True); t = z["caps"].cuda().requires_grad_(True)
    l = HardNegativeContrastiveLoss()(a, t)
    l.backward()
    assert abs(l.item() - z["loss"].item()) < 1e-5 * max(1.0, abs(z["loss"].item()))
    assert rel(a.grad, z["grad_imgs"]) < 1e-5 and rel(t.grad, z["grad_caps"]) < 1e-5
    l9 = HardNegativeContrastiveLoss(margin=0.9)(z["imgs"].cuda(), z["caps"].cuda())
    assert abs(l9.item() - z["loss_margin09"].item()) < 1e-5 * z["loss_margin09"].item()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(300, 96, generator=g); y = (0.3 * x + torch.randn(300, 96, generator=g))
    xr = x.clone().requires_grad_(True); yr = y.clone().requires_grad_(True)
    lr = O.hard_negative(xr, yr); lr.backward()
    xd = x.cuda().requires_grad_(True); yd = y.cuda().requires_grad_(True)
    ld = HardNegativeContrastiveLoss()(xd, yd); ld.backward()
    assert abs(ld.item() - lr.item()) < 1e-4 * abs(lr.item())
    assert rel(xd.grad, xr.grad) < 1e-4 and rel(yd.grad, yr.grad) < 1e-4
    # nmax > 1 (losses.py:909-911: the nmax hardest per row and per column): the reference's formula written out in fp32 torch on the CPU
    def ref_nmax(imgs, caps, nmax, margin=0.2):
        c, i = torch.nn.functional.normalize(caps, dim=-1), torch.nn.functional.normalize(imgs, dim=-1)
        sc = i @ c.t()
        dg = sc.diag()
        sc = sc - 2 * torch.diag(sc.diag())
        mc = torch.sort(sc, 0, descending=True)[0][:nmax, :]
        mi = torch.sort(sc, 1, descending=True)[0][:, :nmax]
        return torch.clamp(mc + (margin - dg).view(1, -1).expand_as(mc), min=0).sum() + torch.clamp(mi + (margin - dg).view(-1, 1).expand_as(mi), min=0).sum()
    for nmax in (2, 5):
        xr = x[:64].clone().requires_grad_(True); yr = y[:64].clone().requires_grad_(True)
        lr = ref_nmax(xr, yr, nmax); lr.backward()
        xd = x[:64].cuda().requires_grad_(True); yd = y[:64].cuda().requires_grad_(True)
        ld = HardNegativeContrastiveLoss(nmax=nmax)(xd, yd); ld.backward()
        assert abs(ld.item() - lr.item()) < 1e-4 * abs(lr.item()), (nmax, ld.item(), lr.item())
        assert rel(xd.grad, xr.grad) < 1e-4 and rel(yd.grad, yr.grad) < 1e-4
    assert float(ZEROGlobalContrastiveLoss()(a, t)) == 0.0
    o = ZEROLocalContrastiveLoss()(torch.zeros(2, 8, 2, 2, device="cuda"), torch.zeros(2, 8, 4, device="cuda"), [4, 4])
    assert float(o.loss0) == 0.0 and float(o.loss1) == 0.0 and o.att_maps == []
