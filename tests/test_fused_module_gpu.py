"""The drop-in path as the fast path: `MedMoEPretrainingLightningModule(fused_step=True)` (`model.fused_step: true`, switched on by
the BASELINE experiments pretraining_medmoe_cfg1..4) runs `Engine.train_step` - embedding all-gather, key-gradient reduce-scatter,
per-layer gradient all-reduce under the backward, fused clip + Adam - behind the reference's module / trainer interface
(reference src/models/medmoe_module.py:284-339, src/train.py:42-108)."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "configs")


@pytest.fixture()
def project_root(monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", ROOT)


def _lit(overrides):
    from medmoe_amd.hydra_lite import compose, instantiate
    cfg = compose(CONFIGS, "train.yaml", overrides)
    return cfg, instantiate(cfg.model)


def _mb(b):
    return {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"]}}


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20))


def test_fused_experiments_compose_and_refuse_what_they_cannot_fuse(project_root):
    """CPU: the four BASELINE experiments switch the fused step on, the reference experiment keeps torch autograd."""
    from medmoe_amd.hydra_lite import compose
    for name in ("cfg1", "cfg2", "cfg3", "cfg4"):
        assert compose(CONFIGS, "train.yaml", [f"experiment=pretraining_medmoe_{name}"]).model.fused_step is True
    sw = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe_swin"])        # the reference's own model on the fused step
    assert sw.model.fused_step is True and sw.model.model.vision.arch == "swin_t" and sw.data.batch_size == 256 and sw.model.model.text.max_length == 25
    base = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe"])
    assert base.model.fused_step is False and base.model.loss.local_loss_global is False


@pytest.mark.gpu
def test_fused_module_steps_equal_engine_steps(project_root):
    """Three optimiser steps through `lit.training_step` against three `Engine.train_step`s of a second engine with the same seed and
    configuration: same losses (1e-5), same parameter update (the wgrads meet in fp32 atomics, so the sign-like first Adam steps agree
    in direction: cosine > 0.999); the trainer's accumulation / clip keys and the scheduler's learning rate reach the engine."""
    import bench
    from medmoe_amd.config import config_by_name
    from medmoe_amd.engine import Engine
    cfg, lit = _lit(["experiment=pretraining_medmoe_cfg2", "model.model.vision.config_name=tiny2", "model.optimizer.lr=0.001"])
    assert lit.fused_step and lit.automatic_optimization is False
    opt = lit.configure_optimizers()["optimizer"]
    lit.configure_fused(cfg.trainer.accumulate_grad_batches, cfg.trainer.gradient_clip_val)
    eng_m = lit.model.engine
    assert (eng_m.cfg.lr, eng_m.cfg.clip, eng_m.cfg.weight_decay) == (1e-3, 0.25, 0.0)
    assert (eng_m.cfg.w_local, eng_m.cfg.w_global, eng_m.cfg.w_cls, eng_m.cfg.temp3) == (0.5, 0.5, 2.0, 10.0)
    c2 = config_by_name("tiny2"); c2.lr = 1e-3
    eng = Engine(c2, "cuda:0", seed=0)
    assert torch.equal(eng.params.p32, lit.model.weights.detach())
    p0 = eng.params.p32.clone()
    b = bench.synthetic_batch(c2, 8, 21, eng.device)
    for it in range(3):
        l_m = lit.training_step(_mb(b), it)
        l_e = eng.train_step(b)["loss"]
        assert abs(float(l_m) - float(l_e)) < 1e-5 * max(1.0, abs(float(l_e))), (it, float(l_m), float(l_e))
    torch.cuda.synchronize()
    u_m, u_e = lit.model.weights.detach() - p0, eng.params.p32 - p0
    cos = float((u_m * u_e).sum() / (u_m.norm() * u_e.norm()))
    assert cos > 0.999 and rel(u_m, u_e) < 5e-2, (cos, rel(u_m, u_e))
    # the bf16 working copies follow the fused update: an evaluation step through the autograd mirror sees the new weights
    with torch.no_grad():
        ev = lit.model_step(_mb(b))
    assert abs(float(ev["loss"]) - float(eng.train_step(b, optimizer=False)["loss"])) < 2e-3 * abs(float(ev["loss"]))
    # ReduceLROnPlateau acts on the (never stepped) torch optimizer; the engine applies its learning rate
    opt.param_groups[0]["lr"] = 2.5e-4
    lit.training_step(_mb(b), 0)
    assert eng_m.cfg.lr == 2.5e-4
    # accumulation: two micro-batches of an accumulation window of 2 == the engine's two half-scaled calls
    lit.configure_fused(2, 0.25)
    eng.cfg.lr = 2.5e-4
    b2 = bench.synthetic_batch(c2, 8, 22, eng.device)
    before_m, before_e = lit.model.weights.detach().clone(), eng.params.p32.clone()
    lit.training_step(_mb(b), 0)
    assert torch.equal(lit.model.weights.detach(), before_m)                       # first micro-batch: no update yet
    lit.training_step(_mb(b2), 1)
    eng.train_step(b, optimizer=False, loss_scale=0.5); eng.train_step(b2, zero_grad=False, loss_scale=0.5)
    torch.cuda.synchronize()
    u_m, u_e = lit.model.weights.detach() - before_m, eng.params.p32 - before_e
    assert float(u_m.norm()) > 0 and float((u_m * u_e).sum() / (u_m.norm() * u_e.norm())) > 0.9


@pytest.mark.gpu
def test_fused_step_refuses_unfusable_configurations(project_root):
    with pytest.raises(NotImplementedError):
        _lit(["experiment=pretraining_medmoe", "model.fused_step=true", "model.model.vision.config_name=tiny2",
              "model.optimizer._target_=torch.optim.SGD"])
    with pytest.raises(NotImplementedError):
        _lit(["experiment=pretraining_medmoe", "model.fused_step=true", "model.model.vision.config_name=tiny2",
              "model.loss.global_loss._target_=src.losses.HardNegativeContrastiveLoss"])


@pytest.mark.gpu
def test_checkpoint_keys_follow_the_reference_layout_and_round_trip(project_root, tmp_path):
    """state_dict keys are the reference's (`model.image_encoder.moe.experts.{e}.proj_convs.{s}.0.weight`, `model.image_encoder.moe.router.*`,
    swin.py:83-92 under med_moe.py:32), not the flat buffer's private layout; a checkpoint written by one module loads into a fresh
    one (other seed) and reproduces its outputs; the legacy flat key still loads."""
    import bench
    cfg, lit = _lit(["experiment=pretraining_medmoe_cfg2", "model.model.vision.config_name=tiny2"])
    sd = lit.state_dict()
    E = lit.model.cfg.n_expert
    for k in ("model.image_encoder.moe.router.0.weight", "model.image_encoder.moe.router.2.bias", "model.image_encoder.vit.layer.0.attention.input_proj.weight",
              "model.text_encoder.layer.0.feedforward.model.0.weight", "model.text_encoder.word_embeddings"):
        assert k in sd, k
    for e in range(E):
        assert tuple(sd[f"model.image_encoder.moe.experts.{e}.proj_convs.3.0.weight"].shape) == (lit.model.cfg.d_out, lit.model.cfg.d_v, 1)
        assert tuple(sd[f"model.image_encoder.moe.experts.{e}.attn_proj.2.weight"].shape) == (1, lit.model.cfg.d_out // 2)
    assert "model.weights" not in sd
    torch.save({"state_dict": sd}, tmp_path / "a.ckpt")
    b = bench.synthetic_batch(lit.model.cfg, 8, 5, lit.model.device)
    with torch.no_grad():
        ref = lit.model_step(_mb(b))
    # a second module with different weights (perturbed in place), then the checkpoint through load_state_dict
    _, lit2 = _lit(["experiment=pretraining_medmoe_cfg2", "model.model.vision.config_name=tiny2"])
    with torch.no_grad():
        lit2.model.weights.mul_(1.5)
        t = lit2.model.engine.params.text
        t["layer.0.feedforward.model.0.weight"] = t["layer.0.feedforward.model.0.weight"] * 0.5
        assert abs(float(lit2.model_step(_mb(b))["loss"]) - float(ref["loss"])) > 1e-3
    missing = lit2.load_state_dict(torch.load(tmp_path / "a.ckpt", map_location="cpu", weights_only=True)["state_dict"])
    assert not missing.missing_keys and not missing.unexpected_keys
    with torch.no_grad():
        got = lit2.model_step(_mb(b))
    assert abs(float(got["loss"]) - float(ref["loss"])) < 1e-5 * abs(float(ref["loss"]))
    assert torch.equal(lit2.model.weights.detach(), lit.model.weights.detach())
    # legacy layout: the flat buffer under `model.weights`
    legacy = {"model.weights": lit.model.weights.detach().clone() * 1.0}
    lit2.model.weights.data.mul_(0.5)
    lit2.load_state_dict(legacy, strict=False)
    assert torch.equal(lit2.model.weights.detach(), lit.model.weights.detach())


@pytest.mark.gpu
def test_two_rank_fused_module_step_equals_one_process():
    """tools/two_rank_module.py: two gloo ranks on the one GPU through the Hydra-built module (fused step, globalised local loss) against
    one process on the concatenated batch - identical replicas, same loss (5e-3), same averaged gradient (2e-2)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_rank_module.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "two-rank fused module path OK" in r.stdout


@pytest.mark.gpu
def test_train_py_under_torch_distributed_run_two_ranks(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 src/train.py experiment=pretraining_medmoe_cfg2 trainer=ddp ...` (gloo backend,
    both ranks on the one test GPU): the fused step's collectives, rank-0 checkpointing and the validation all-reduce run end to end,
    and both replicas finish with the SAME weights (sha256 of the parameters printed by every rank)."""
    env = dict(os.environ, MEDMOE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MEDMOE_LOG_PARAM_HASH="1")
    env.pop("PROJECT_ROOT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "src", "train.py"), "experiment=pretraining_medmoe_cfg2", "trainer=ddp", "trainer.devices=2",
           "model.model.vision.config_name=tiny2", "data.synthetic_size=48", "data.synthetic_vocab=97", "data.synthetic_classes=3",
           "data.batch_size=16", "data.max_len=16", "data.num_workers=0", "trainer.max_epochs=2", "model.optimizer.lr=0.001",
           "extras.print_config=false", f"callbacks.model_checkpoint.dirpath={tmp_path}/ckpt", "+optimized_metric=train/loss"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    hashes = dict(re.findall(r"\[rank (\d)\] param sha256 ([0-9a-f]{16})", out))
    assert set(hashes) == {"0", "1"} and hashes["0"] == hashes["1"], hashes
    steps = set(re.findall(r"global_step (\d+)", out))
    assert steps == {"6"}, steps                      # 48 samples / global batch 16 = 3 steps per epoch, 2 epochs
    assert os.path.exists(os.path.join(str(tmp_path), "ckpt", "last.ckpt"))
    m = re.search(r"metrics: train/loss=([0-9.]+), val/loss=([0-9.]+)", out)
    assert m and 0 < float(m.group(1)) < 100 and 0 < float(m.group(2)) < 100, out[-1500:]


def test_data_parallel_ranks_read_disjoint_samples():
    """CPU: under a two-rank trainer the datamodule hands rank r the samples r, r + 2, ... - disjoint, together the whole set -
    at the per-device batch size (global batch / world, unimed_datamodule.py:74-79)."""
    from src.data.unimed_datamodule import UnimedDataModule

    class T:
        def __init__(self, rank):
            self.world_size, self.global_rank = 2, rank

    seen = []
    for r in range(2):
        dm = UnimedDataModule(batch_size=8, synthetic_size=24, max_len=16, synthetic_vocab=97, synthetic_classes=3)
        dm.trainer = T(r)
        dm.setup("fit")
        assert dm.batch_size_per_device == 4
        caps = []
        for batch in dm.train_dataloader():
            assert len(batch["caption"]) == 4
            caps += [tuple(c.tolist()) for c in batch["caption"]]
        seen.append(caps)
    assert len(seen[0]) == len(seen[1]) == 12 and not set(seen[0]) & set(seen[1])
    ref = UnimedDataModule(batch_size=8, synthetic_size=24, max_len=16, synthetic_vocab=97, synthetic_classes=3)
    everything = [tuple(ref.data_train[i][1].tolist()) for i in range(24)]
    assert seen[0] == everything[0::2] and seen[1] == everything[1::2]


@pytest.mark.gpu
def test_freeze_bert_false_trains_the_text_tower_through_the_fused_module(project_root):
    """`model.model.text.freeze_bert=false` (reference text_encoder.py:27-30) with the fused step: the Hydra-built module's engine carries a
    text store, training steps move text parameters, the checkpoint's `model.text_encoder.*` entries follow and load back into a fresh
    module; the torch-autograd mirror (fused_step off) refuses to train with an unfrozen tower instead of silently freezing it."""
    import bench
    ov = ["experiment=pretraining_medmoe_cfg2", "model.model.vision.config_name=tiny2", "model.model.text.freeze_bert=false", "model.optimizer.lr=0.001"]
    cfg, lit = _lit(ov)
    eng = lit.model.engine
    assert eng.cfg.freeze_text is False and eng.tstore is not None
    lit.configure_optimizers(); lit.configure_fused(1, 0.25)
    b = bench.synthetic_batch(eng.cfg, 8, 3, eng.device)
    before = eng.tstore.p32.clone()
    l0 = float(lit.training_step(_mb(b), 0))
    for i in range(5):
        l1 = float(lit.training_step(_mb(b), i + 1))
    assert l1 < l0 and float((eng.tstore.p32 - before).abs().max()) > 0
    sd = lit.state_dict()
    k = "model.text_encoder.layer.0.feedforward.model.0.weight"
    assert torch.equal(sd[k], eng.tstore.w16("layer.0.feedforward.model.0.weight"))
    _, lit2 = _lit(ov)
    lit2.load_state_dict({kk: v.clone() for kk, v in sd.items()})
    t2 = lit2.model.engine.tstore
    assert torch.equal(t2.w16("layer.0.feedforward.model.0.weight"), eng.tstore.w16("layer.0.feedforward.model.0.weight"))
    assert torch.equal(t2.f32("position_embeddings"), eng.tstore.f32("position_embeddings"))
    # the autograd mirror keeps the text tower frozen: training it there is refused
    _, lit3 = _lit(["experiment=pretraining_medmoe", "model.model.vision.config_name=tiny2", "model.model.text.freeze_bert=false"])
    with pytest.raises(NotImplementedError):
        lit3.training_step(_mb(b), 0)


@pytest.mark.gpu
@pytest.mark.parametrize("overrides", [["experiment=pretraining_medmoe_cfg2", "model.model.vision.config_name=tiny2"],
                                       ["experiment=pretraining_medmoe", "model.model.vision.arch=swin_t", "model.fused_step=true",
                                        "model.model.vision.num_experts=3", "model.model.text.n_layer=2"]], ids=["vit", "swin"])
def test_fused_adam_state_travels_with_the_checkpoint(project_root, tmp_path, overrides):
    """`on_save_checkpoint` / `on_load_checkpoint` (the Lightning hooks; the stand-in trainer calls them): a module restored from a checkpoint
    continues the SAME optimisation - its third step equals the original's third step; restored without the moments it does not."""
    import bench
    ov = overrides + ["model.optimizer.lr=0.001"]

    def build():
        _, lit = _lit(ov)
        lit.train(); lit.configure_optimizers(); lit.configure_fused(1, 0.25)
        if getattr(lit.model, "swin", None) is not None:
            lit.model.swin.drop_path_rate = 0.0
        return lit

    def batch(lit, seed):
        b = bench.synthetic_batch(lit.model.cfg, 8, seed, lit.model.device)
        b["label"] = b["label"] % lit.model.cfg.n_expert
        return {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"], "token_type": b["token_type"]}}

    def flat(lit):
        return torch.cat([p.detach().float().reshape(-1) for p in lit.parameters() if p.requires_grad])

    a = build()
    for it in range(2):
        a.training_step(batch(a, 90 + it), it)
    ck = {"state_dict": a.state_dict()}
    a.on_save_checkpoint(ck)
    path = os.path.join(str(tmp_path), "c.ckpt")
    torch.save(ck, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert ck["fused_adam"] and all(v["step"] == 2 for v in ck["fused_adam"].values())
    r, n = build(), build()
    r.load_state_dict(ck["state_dict"]); r.on_load_checkpoint(ck)
    n.load_state_dict(ck["state_dict"])
    p2 = flat(a).clone()
    assert rel(flat(r), p2) < 1e-7 and rel(flat(n), p2) < 1e-7
    b3 = batch(a, 93)
    for m in (a, r, n):
        m.training_step(b3, 2)
    torch.cuda.synchronize()
    ua, ur, un = flat(a) - p2, flat(r) - p2, flat(n) - p2
    assert float(ua.norm()) > 0 and rel(ur, ua) < 2e-2, rel(ur, ua)          # same moments, same step count: the same update (fp32 atomics aside)
    assert rel(un, ua) > 0.2, rel(un, ua)                                     # fresh moments: Adam's first step is a different update
