"""Row b' of the coverage table: the Hydra entry point `src/train.py experiment=pretraining_medmoe` and its config chain
(reference src/train.py:42-131, configs/experiment/pretraining_medmoe.yaml:6-11, configs/model/med-moe_pretraining.yaml:5,31,36,
configs/model/med-moe.yaml:16-44).  hydra / lightning are not installed here: the YAML tree is composed by
medmoe_amd.hydra_lite; every `_target_` must resolve by importlib."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "configs")


@pytest.fixture()
def project_root(monkeypatch):
    monkeypatch.setenv("PROJECT_ROOT", ROOT)


def test_config_chain_composes_with_the_reference_values(project_root):
    from medmoe_amd.hydra_lite import compose
    cfg = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe"])
    used = cfg["_composed_from_"]
    for f in ("train.yaml", "experiment/pretraining_medmoe.yaml", "data/unimed.yaml", "model/med-moe_pretraining.yaml",
              "model/med-moe.yaml", "trainer/default.yaml", "callbacks/default.yaml", "paths/default.yaml"):
        assert f in used, f
    # values of the reference experiment (pretraining_medmoe.yaml:14-34, med-moe_pretraining.yaml:5-41, med-moe.yaml:16-44)
    assert cfg.seed == 12345 and cfg.tags == ["unimed", "pretraining"]
    assert cfg.trainer.gradient_clip_val == 0.25 and cfg.trainer.accumulate_grad_batches == 10 and cfg.trainer.max_epochs == 10
    assert cfg.data.batch_size == 256 and cfg.data._target_ == "src.data.unimed_datamodule.UnimedDataModule"
    assert cfg.model._target_ == "src.models.medmoe_module.MedMoEPretrainingLightningModule"
    assert cfg.model.optimizer.lr == 5e-5 and cfg.model.optimizer.weight_decay == 0.0 and cfg.model.optimizer._partial_ is True
    assert cfg.model.scheduler._target_ == "torch.optim.lr_scheduler.ReduceLROnPlateau" and cfg.model.scheduler.patience == 10
    lc = cfg.model.loss
    assert (lc.global_loss_weight, lc.local_loss_weight, lc.classifier_loss_weight) == (0.5, 0.5, 2.0)
    assert (lc.temp1, lc.temp2, lc.temp3) == (4.0, 5.0, 10.0) and lc.soft_label is False
    assert lc.global_loss._target_ == "src.losses.GLORIAGlobalContrastiveLoss"
    assert lc.local_loss._target_ == "src.losses.GLORIALocalContrastiveLoss"
    m = cfg.model.model
    assert m._target_ == "src.models.components.med_moe.MedMoE"
    assert m.text.max_length == 25 and m.text.last_n_layers == 4 and m.text.freeze_bert is True and m.vision.embed_dim == 768
    assert (m.vision.arch, m.vision.num_experts, m.vision.top_k) == ("vit_b16", 6, 1)       # the build's extension keys
    # interpolations: ${paths.*}, ${oc.env:PROJECT_ROOT}, ${hydra:runtime.output_dir}
    assert cfg.paths.root_dir == ROOT and cfg.data.data_dir == os.path.join(ROOT, "datasets") + "/"
    assert cfg.trainer.default_root_dir == cfg.paths.output_dir


def test_overrides_group_choice_and_baseline_experiments(project_root):
    from medmoe_amd.hydra_lite import compose
    cfg = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe_cfg2", "trainer=ddp", "model.optimizer.lr=1e-4", "+extra.flag=true"])
    v = cfg.model.model.vision
    assert (v.arch, v.num_experts, v.top_k) == ("vit_b16", 8, 2) and cfg.model.model.text.max_length == 77
    assert cfg.data.batch_size == 1024 and cfg.trainer.accumulate_grad_batches == 1
    assert cfg.trainer.devices == 8 and cfg.trainer.strategy == "ddp_find_unused_parameters_true" and cfg.trainer.max_epochs == 10
    assert cfg.model.optimizer.lr == 1e-4 and cfg.extra.flag is True
    for name, (arch, ne, tk) in {"cfg1": ("vit_b16", 4, 1), "cfg4": ("vit_l14", 16, 2)}.items():
        c = compose(CONFIGS, "train.yaml", [f"experiment=pretraining_medmoe_{name}"]).model.model.vision
        assert (c.arch, c.num_experts, c.top_k) == (arch, ne, tk)
        # the key the fp8 experiment exists for reaches the engine configuration (expert_dtype: fp8 -> MedMoEConfig.expert_fp8)
        from src.models.components.med_moe import config_from_hydra
        mm = compose(CONFIGS, "train.yaml", [f"experiment=pretraining_medmoe_{name}"]).model.model
        assert c.expert_dtype == ("fp8" if name == "cfg4" else "bf16")
        assert config_from_hydra(mm.vision, mm.text).expert_fp8 == (name == "cfg4")
    # the experiment pins `trainer.accelerator: gpu` after the trainer group is merged (pretraining_medmoe.yaml:25), as in the reference
    sim = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe", "trainer=ddp_sim"]).trainer
    assert (sim.devices, sim.strategy, sim.accelerator) == (2, "ddp_spawn", "gpu")
    assert compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe", "trainer=cpu", "trainer.accelerator=cpu"]).trainer.accelerator == "cpu"


def test_soft_label_configuration_composes(project_root):
    """The reference's commented-in alternative (med-moe_pretraining.yaml:25-37): soft_label with the two Soft-GLoRIA classes."""
    from medmoe_amd.hydra_lite import compose, instantiate, locate
    cfg = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe", "model.loss.soft_label=true",
                                          "model.loss.global_loss._target_=src.losses.SoftGLORIAGlobalContrastiveLoss",
                                          "model.loss.local_loss._target_=src.losses.SoftGLORIALocalContrastiveLoss"])
    lc = cfg.model.loss
    assert lc.soft_label is True and (lc.topk, lc.threshold0, lc.threshold1) == (5, 0.98, 0.97)
    import src.losses as L
    assert locate(lc.global_loss._target_) is L.SoftGLORIAGlobalContrastiveLoss
    assert isinstance(instantiate(lc.local_loss), L.SoftGLORIALocalContrastiveLoss)
    assert issubclass(L.SoftGLORIALocalContrastiveLoss, L.GLORIALocalContrastiveLoss)


def test_every_target_resolves_and_host_objects_construct(project_root):
    """Every `_target_` of the composed tree imports; the objects that need no GPU are built from their configs."""
    import torch
    from medmoe_amd.hydra_lite import compose, instantiate, locate, targets_of
    cfg = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe", "data.num_workers=0", "data.synthetic_size=64"])
    targets = targets_of(cfg)
    assert len(targets) == 10
    for t in targets:
        assert callable(locate(t)), t
    dm = instantiate(cfg.data)
    dm.setup("fit", world_size=8)
    assert dm.batch_size_per_device == 32                              # 256 // 8 (unimed_datamodule.py:74-79)
    batch = next(iter(dm.train_dataloader()))
    assert set(batch) == {"image", "caption", "label"} and len(batch["image"]) == 32
    opt = instantiate(cfg.model.optimizer)(params=[torch.nn.Parameter(torch.zeros(3))])
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 5e-5 and opt.defaults["weight_decay"] == 0.0
    sch = instantiate(cfg.model.scheduler)(optimizer=opt)
    assert isinstance(sch, torch.optim.lr_scheduler.ReduceLROnPlateau)
    from src.utils import instantiate_callbacks
    cbs = instantiate_callbacks(cfg.callbacks)
    assert len(cbs) == 2
    tr = instantiate(cfg.trainer, callbacks=cbs, logger=[])
    assert tr.accumulate_grad_batches == 10 and tr.gradient_clip_val == 0.25 and tr.max_epochs == 10
    with pytest.raises(RuntimeError):                                    # no CPU path in this build
        instantiate(compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe", "trainer=cpu", "trainer.accelerator=cpu"]).trainer)
    # keys that would select code outside the hot path are rejected at model construction, not silently ignored
    from src.models.components.med_moe import config_from_hydra
    c = config_from_hydra(cfg.model.model.vision, cfg.model.model.text)
    assert (c.n_expert, c.top_k, c.max_len, c.d_v, c.n_layer_t) == (6, 1, 25, 768, 12)
    unfrozen = dict(cfg.model.model.text); unfrozen["freeze_bert"] = False          # the text tower trains too: reaches the engine configuration
    assert config_from_hydra(cfg.model.model.vision, unfrozen).freeze_text is False and c.freeze_text is True
    bad = dict(cfg.model.model.text); bad["norm"] = True
    with pytest.raises(NotImplementedError):
        config_from_hydra(cfg.model.model.vision, bad)


@pytest.mark.gpu
def test_module_constructs_from_the_config_chain_and_steps(project_root):
    """hydra-style instantiation of the whole model node on the GPU box, one model_step + backward."""
    import torch
    from medmoe_amd.hydra_lite import compose, instantiate
    cfg = compose(CONFIGS, "train.yaml", ["experiment=pretraining_medmoe", "model.model.vision.config_name=tiny2"])
    lit = instantiate(cfg.model)
    from src.losses import GLORIAGlobalContrastiveLoss, GLORIALocalContrastiveLoss
    from src.models.medmoe_module import MedMoEPretrainingLightningModule
    assert isinstance(lit, MedMoEPretrainingLightningModule)
    assert isinstance(lit.global_loss, GLORIAGlobalContrastiveLoss) and isinstance(lit.local_loss, GLORIALocalContrastiveLoss)
    assert (lit.local_loss_weight, lit.global_loss_weight, lit.classifier_loss_weight) == (0.5, 0.5, 2.0)
    import bench
    c = lit.model.cfg
    b = bench.synthetic_batch(c, 8, 5, lit.model.device)
    out = lit.model_step({"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"]}})
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and float(lit.model.weights.grad.abs().max()) > 0
    opt = lit.configure_optimizers()["optimizer"]
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 5e-5


@pytest.mark.gpu
def test_train_py_runs_the_experiment_end_to_end(tmp_path):
    """`python src/train.py experiment=pretraining_medmoe ...` at unit-test scale: two epochs over the synthetic shards with
    gradient accumulation, clipping, validation, checkpoint callback; finite train / val losses are reported."""
    env = dict(os.environ)
    env.pop("PROJECT_ROOT", None)
    cmd = [sys.executable, os.path.join(ROOT, "src", "train.py"), "experiment=pretraining_medmoe", "model.model.vision.config_name=tiny2",
           "data.synthetic_size=32", "data.synthetic_vocab=97", "data.synthetic_classes=3", "data.batch_size=8", "data.max_len=16",
           "data.num_workers=0", "trainer.max_epochs=3", "trainer.accumulate_grad_batches=2", "model.optimizer.lr=0.001",
           "extras.print_config=false", f"callbacks.model_checkpoint.dirpath={tmp_path}/ckpt", "+optimized_metric=train/loss"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert os.path.exists(os.path.join(str(tmp_path), "ckpt", "last.ckpt"))
    import re
    m = re.search(r"metrics: train/loss=([0-9.]+), val/loss=([0-9.]+)", r.stdout + r.stderr)
    assert m and 0 < float(m.group(1)) < 100 and 0 < float(m.group(2)) < 100, (r.stdout + r.stderr)[-1500:]
    import torch
    sd = torch.load(os.path.join(str(tmp_path), "ckpt", "last.ckpt"), map_location="cpu", weights_only=True)
    # reference-style keys (src/models/components/med_moe.py state-dict hooks), not the flat buffer's private layout
    assert "model.image_encoder.moe.router.0.weight" in sd["state_dict"] and "model.weights" not in sd["state_dict"] and sd["epoch"] == 2


@pytest.mark.gpu
def test_train_py_with_the_swin_tower_and_soft_labels(tmp_path):
    """`python src/train.py experiment=pretraining_medmoe model.model.vision.arch=swin_t model.loss.soft_label=true ...`: the reference's own
    image encoder and its Soft-GLoRIA configuration through the Hydra entry point - one epoch over synthetic shards, finite losses."""
    env = dict(os.environ)
    env.pop("PROJECT_ROOT", None)
    cmd = [sys.executable, os.path.join(ROOT, "src", "train.py"), "experiment=pretraining_medmoe", "model.model.vision.arch=swin_t",
           "model.model.vision.num_experts=3", "model.model.text.n_layer=2", "model.loss.soft_label=true",
           "model.loss.global_loss._target_=src.losses.SoftGLORIAGlobalContrastiveLoss",
           "model.loss.local_loss._target_=src.losses.SoftGLORIALocalContrastiveLoss", "model.loss.threshold0=0.995", "model.loss.threshold1=0.99",
           "data.synthetic_size=16", "data.synthetic_classes=3", "data.batch_size=8", "data.num_workers=0", "trainer.max_epochs=1",
           "trainer.accumulate_grad_batches=1", "extras.print_config=false", f"callbacks.model_checkpoint.dirpath={tmp_path}/ckpt",
           "+optimized_metric=train/loss"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    import re
    m = re.search(r"metrics: train/loss=([0-9.]+), val/loss=([0-9.]+)", r.stdout + r.stderr)
    assert m and 0 < float(m.group(1)) < 100 and 0 < float(m.group(2)) < 100, (r.stdout + r.stderr)[-1500:]
    # the checkpoint holds the reference's key layout for its own model: HF SwinModel names under image_encoder.model, the MoE under
    # image_encoder.moe (swin.py:119-128, med_moe.py:32) - not `swin.params.<i>`
    import torch
    sd = torch.load(os.path.join(str(tmp_path), "ckpt", "last.ckpt"), map_location="cpu", weights_only=True)["state_dict"]
    assert "model.image_encoder.model.embeddings.patch_embeddings.projection.weight" in sd
    assert tuple(sd["model.image_encoder.moe.experts.2.proj_convs.0.0.weight"].shape) == (768, 96, 1)
    assert not [k for k in sd if ".params." in k or k.endswith("model.weights")]
