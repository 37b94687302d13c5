"""The reference's OWN model (configs/experiment/pretraining_medmoe.yaml: Swin-T tower + six pyramid experts + 3136-region local loss + frozen
text tower; reference src/models/components/swin.py:119-149, src/models/medmoe_module.py:284-339) in the fused step:
`medmoe_amd.swin_engine.SwinEngine` behind `MedMoEPretrainingLightningModule(fused_step=True)` against the torch-autograd mirror
(`src.models.components.swin.SWIN` + `src.losses` + `clip_grad_norm_` + `torch.optim.Adam`) from the same initial state."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "configs")
SWIN = ["experiment=pretraining_medmoe", "model.model.vision.arch=swin_t", "model.optimizer.lr=0.0002"]


@pytest.fixture()
def project_root(monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", ROOT)


def _lit(overrides):
    from medmoe_amd.hydra_lite import compose, instantiate
    cfg = compose(CONFIGS, "train.yaml", overrides)
    return cfg, instantiate(cfg.model)


def _batch(lit, B, seed):
    import bench
    b = bench.synthetic_batch(lit.model.cfg, B, seed, lit.model.device)
    b["label"] = b["label"] % lit.model.cfg.n_expert
    return {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"], "token_type": b["token_type"]}}


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20))


@pytest.mark.gpu
def test_fused_swin_steps_track_the_autograd_mirror(project_root):
    """Two optimiser steps each way from the same seed, stochastic depth off (the two paths draw their masks from different generators):
    every reported loss agrees, the parameter updates point the same way, and afterwards both models score a fresh batch alike."""
    cfg, ref = _lit(SWIN)
    _, fus = _lit(SWIN + ["model.fused_step=true"])
    assert fus.fused_step and not ref.fused_step
    ref.model.swin.drop_path_rate = fus.model.swin.drop_path_rate = 0.0
    ref.train(); fus.train()
    clip = float(cfg.trainer.gradient_clip_val)
    fus.configure_optimizers()
    fus.configure_fused(1, clip)
    params = [p for p in ref.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=2e-4, weight_decay=float(cfg.model.optimizer.weight_decay))
    names = ref.model.swin._names
    p0 = {n: p.detach().clone() for n, p in zip(names, ref.model.swin.params)}
    for n, p in zip(names, fus.model.swin.params):
        assert torch.equal(p.detach(), p0[n]), n
    B = 8
    for it in range(2):
        mb = _batch(ref, B, 40 + it)
        opt.zero_grad()
        out_r = ref.model_step(mb)
        out_r["loss"].backward()
        torch.nn.utils.clip_grad_norm_(params, clip)
        opt.step()
        out_f = fus.fused_training_step(mb)
        for k in ("loss", "l_loss", "g_loss", "classifier_loss", "classifier_acc"):
            a, b = float(out_f[k]), float(out_r[k])
            assert abs(a - b) < 2e-3 * max(1.0, abs(b)), (it, k, a, b)
    torch.cuda.synchronize()
    # the nn.Parameters of the fused module alias the arena the fused Adam kernel updates
    num = den = dot = 0.0
    for n, pr, pf in zip(names, ref.model.swin.params, fus.model.swin.params):
        ur, uf = (pr.detach() - p0[n]).double(), (pf.detach() - p0[n]).double()
        dot += float((ur * uf).sum()); num += float(uf.pow(2).sum()); den += float(ur.pow(2).sum())
    cos = dot / (num ** 0.5 * den ** 0.5)
    assert den > 0 and cos > 0.98 and abs(num ** 0.5 / den ** 0.5 - 1.0) < 0.05, (cos, num, den)
    ref.eval(); fus.eval()
    mb = _batch(ref, B, 50)
    with torch.no_grad():
        a, b = float(ref.model_step(mb)["loss"]), float(fus.model_step(mb)["loss"])
    assert abs(a - b) < 5e-3 * abs(a), (a, b)


@pytest.mark.gpu
def test_fused_swin_accumulation_and_trainer_keys(project_root):
    """trainer.accumulate_grad_batches / gradient_clip_val reach the Swin engine: the first micro-batch of a window of two leaves the
    parameters alone, the second updates them; one window of two half-scaled micro-batches of the SAME batch equals one full step."""
    _, a = _lit(SWIN + ["model.fused_step=true"])
    _, b = _lit(SWIN + ["model.fused_step=true"])
    for m in (a, b):
        m.model.swin.drop_path_rate = 0.0
        m.train(); m.configure_optimizers()
    a.configure_fused(2, 0.25); b.configure_fused(1, 0.25)
    assert a.model.engine.cfg.clip == 0.25
    mb = _batch(a, 8, 60)
    a.training_step(mb, 0)                                            # builds the arenas; no update yet
    pa = [p.detach().clone() for p in a.model.swin.params]
    assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(a.model.swin.params, b.model.swin.params))
    a.training_step(mb, 1)
    b.training_step(mb, 0)
    torch.cuda.synchronize()
    dot = na = nb = 0.0
    for p0, p, q in zip(pa, a.model.swin.params, b.model.swin.params):
        ua, ub = (p.detach() - p0).double(), (q.detach() - p0).double()
        dot += float((ua * ub).sum()); na += float(ua.pow(2).sum()); nb += float(ub.pow(2).sum())
    assert na > 0 and dot / (na * nb) ** 0.5 > 0.98, (dot, na, nb)


@pytest.mark.gpu
def test_fused_swin_stochastic_depth_and_eval(project_root):
    """Train mode draws per-block keep masks on the device (SwinConfig.drop_path_rate 0.1): two steps on one batch give different losses only
    through the masks and the update; eval_step draws no masks and equals the autograd mirror's evaluation."""
    from medmoe_amd.swin_engine import SwinEngine
    _, lit = _lit(SWIN + ["model.fused_step=true"])
    lit.train(); lit.configure_optimizers(); lit.configure_fused(1, 0.25)
    mb = _batch(lit, 8, 70)
    out = lit.fused_training_step(mb)
    assert all(torch.isfinite(v).all() for v in out.values())
    se = lit._swin_engine
    assert isinstance(se, SwinEngine) and se.training
    masks = se._drop_path_masks(8)
    assert masks[0] is None and all(m is not None and m.shape == (8,) and set(m.unique().tolist()) <= {0.0, 1.0} for m in masks[1:])
    cap = mb["caption"]
    eb = {"image": mb["image"], "label": mb["label"], "ids": cap["ids"], "attn_mask": cap["attn_mask"], "token_type": cap["token_type"]}
    e1, e2 = se.eval_step(eb), se.eval_step(eb)
    assert abs(float(e1["loss"]) - float(e2["loss"])) < 1e-5 * abs(float(e1["loss"]))      # fp32 atomics in the grouped wgrad-shaped GEMMs: order-dependent last bits
    lit.eval()
    with torch.no_grad():
        ev = lit.model_step(mb)
    for k in ("loss", "l_loss", "g_loss", "classifier_loss", "classifier_acc"):
        assert abs(float(ev[k]) - float(e1[k])) < 2e-3 * max(1.0, abs(float(ev[k]))), (k, float(ev[k]), float(e1[k]))


@pytest.mark.gpu
def test_train_py_runs_the_reference_model_through_the_fused_step(tmp_path):
    """`python src/train.py experiment=pretraining_medmoe model.model.vision.arch=swin_t model.fused_step=true ...`: the Hydra entry point the
    reference names, its own model, the fused step - one epoch over synthetic shards with an accumulation window of two; the checkpoint holds
    the UPDATED weights under the reference's names (the parameters alias the arena the fused Adam kernel writes)."""
    import re
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("PROJECT_ROOT", None)
    env["MEDMOE_LOG_PARAM_HASH"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "src", "train.py"), "experiment=pretraining_medmoe", "model.model.vision.arch=swin_t",
           "model.fused_step=true", "model.model.vision.num_experts=3", "model.model.text.n_layer=2",
           "data.synthetic_size=32", "data.synthetic_classes=3", "data.batch_size=8", "data.num_workers=0", "trainer.max_epochs=1",
           "trainer.accumulate_grad_batches=2", "extras.print_config=false", f"callbacks.model_checkpoint.dirpath={tmp_path}/ckpt",
           "+optimized_metric=train/loss"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    m = re.search(r"metrics: train/loss=([0-9.]+), val/loss=([0-9.]+)", r.stdout + r.stderr)
    assert m and 0 < float(m.group(1)) < 100 and 0 < float(m.group(2)) < 100, (r.stdout + r.stderr)[-1500:]
    sd = torch.load(os.path.join(str(tmp_path), "ckpt", "last.ckpt"), map_location="cpu", weights_only=True)["state_dict"]
    w = sd["model.image_encoder.model.encoder.layers.0.blocks.0.mlp.fc1.weight"]
    assert tuple(w.shape) == (384, 96) and tuple(sd["model.image_encoder.moe.experts.2.proj_convs.0.0.weight"].shape) == (768, 96, 1)
    # a fresh module of the same seed has the initial weights: the checkpoint's differ (the fused steps reached the nn.Parameters)
    os.environ["PROJECT_ROOT"] = ROOT
    try:
        _, fresh = _lit(SWIN + ["model.model.vision.num_experts=3", "model.model.text.n_layer=2"])
    finally:
        os.environ.pop("PROJECT_ROOT", None)
    w0 = fresh.state_dict()["model.image_encoder.model.encoder.layers.0.blocks.0.mlp.fc1.weight"].cpu()
    assert w0.shape == w.shape and float((w0 - w).abs().max()) > 0


@pytest.mark.gpu
def test_two_ranks_step_the_reference_model_as_replicas():
    """tools/two_rank_swin.py: two gloo ranks on the one GPU through the Hydra-built module (Swin-T, fused step): bit-identical replicas after
    the step, the optimiser's gradient = the mean of the ranks' arenas, the gathered global loss = the one-process loss on the whole batch."""
    import subprocess
    import sys
    env = dict(os.environ, PROJECT_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_rank_swin.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TWO_RANK_SWIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_fused_swin_soft_gloria_matches_the_mirror(project_root):
    """loss.soft_label=true with the two Soft-GLoRIA classes (reference medmoe_module.py:291-296, losses.py:826-883) through the fused Swin step:
    the same loss values as the autograd mirror on the same batch and weights."""
    soft = ["model.loss.soft_label=true", "model.loss.global_loss._target_=src.losses.SoftGLORIAGlobalContrastiveLoss",
            "model.loss.local_loss._target_=src.losses.SoftGLORIALocalContrastiveLoss", "model.loss.threshold0=0.995", "model.loss.threshold1=0.99"]
    _, ref = _lit(SWIN + soft)
    _, fus = _lit(SWIN + soft + ["model.fused_step=true"])
    ref.model.swin.drop_path_rate = fus.model.swin.drop_path_rate = 0.0
    ref.train(); fus.train()
    fus.configure_optimizers(); fus.configure_fused(1, 0.25)
    assert fus.model.engine.cfg.soft_label
    mb = _batch(ref, 8, 81)
    with torch.no_grad():
        out_r = ref.model_step(mb)
    out_f = fus.fused_training_step(mb)
    for k in ("loss", "l_loss", "g_loss", "classifier_loss"):
        a, b = float(out_f[k]), float(out_r[k])
        assert abs(a - b) < 3e-3 * max(1.0, abs(b)), (k, a, b)
