"""Entry point: `python src/train.py experiment=pretraining_medmoe [key=value ...]` (reference src/train.py:42-131).

Same flow as the reference: compose configs/train.yaml, seed, instantiate datamodule / model / callbacks / loggers / trainer
from their `_target_`s, trainer.fit, optional trainer.test, return the metric dict.  With hydra + lightning installed the
real `@hydra.main` / Lightning Trainer run; in images without them (this one) the YAML tree is composed by
medmoe_amd.hydra_lite and `lightning.pytorch.trainer.Trainer` resolves to medmoe_amd.trainer.Trainer.
Multi-GPU: `python -m torch.distributed.run --nproc-per-node 8 src/train.py experiment=... trainer=ddp` (one process per GPU).
"""
import os
import sys
from typing import Any, Dict, Optional, Tuple


def setup_root(start: str, indicator: str = ".project-root") -> str:
    """rootutils.setup_root equivalent (train.py:11): find the directory holding `indicator`, put it on sys.path, export
    PROJECT_ROOT (configs/paths/default.yaml reads it)."""
    d = os.path.dirname(os.path.abspath(start))
    root = None
    while True:
        if os.path.exists(os.path.join(d, indicator)):
            root = d
            break
        parent = os.path.dirname(d)
        if parent == d:
            break
        d = parent
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(start)))
    if root not in sys.path:
        sys.path.insert(0, root)
    os.environ.setdefault("PROJECT_ROOT", root)
    return root


ROOT = setup_root(__file__)

try:                                                     # pragma: no cover - not installed in this image
    import hydra
    from hydra.utils import instantiate
    HAVE_HYDRA = True
except ImportError:
    from medmoe_amd.hydra_lite import compose, instantiate
    HAVE_HYDRA = False

from src.utils import (RankedLogger, extras, get_metric_value, instantiate_callbacks, instantiate_loggers,  # noqa: E402
                       log_hyperparameters, task_wrapper)

log = RankedLogger(__name__, rank_zero_only=True)


def seed_everything(seed: int) -> None:
    try:                                                 # pragma: no cover
        import lightning as L
        L.seed_everything(seed, workers=True)
    except ImportError:
        import random

        import numpy as np
        import torch
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)


def init_distributed() -> None:
    """One process per GPU (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE); backend nccl = RCCL."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not torch.distributed.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MEDMOE_DIST_BACKEND", "nccl")
        dev = local % torch.cuda.device_count()                   # one index for set_device AND the process group's device
        torch.cuda.set_device(dev)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{dev}"))
        else:
            torch.distributed.init_process_group(backend)


@task_wrapper
def train(cfg) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """train.py:42-108."""
    if cfg.get("seed"):
        seed_everything(cfg.seed)
    init_distributed()
    log.info(f"Instantiating datamodule <{cfg.data._target_}>")
    datamodule = instantiate(cfg.data)
    log.info(f"Instantiating model <{cfg.model._target_}>")
    model = instantiate(cfg.model)
    callbacks = instantiate_callbacks(cfg.get("callbacks"))
    logger = instantiate_loggers(cfg.get("logger"))
    log.info(f"Instantiating trainer <{cfg.trainer._target_}>")
    trainer = instantiate(cfg.trainer, callbacks=callbacks, logger=logger)
    object_dict = {"cfg": cfg, "datamodule": datamodule, "model": model, "callbacks": callbacks, "logger": logger, "trainer": trainer}
    if logger:
        log_hyperparameters(object_dict)
    if cfg.get("train"):
        trainer.fit(model=model, datamodule=datamodule, ckpt_path=cfg.get("ckpt_path"))
    train_metrics = dict(trainer.callback_metrics)
    if cfg.get("test"):
        ckpt_path = trainer.checkpoint_callback.best_model_path if trainer.checkpoint_callback else ""
        trainer.test(model=model, datamodule=datamodule, ckpt_path=ckpt_path or None)
    return {**train_metrics, **dict(trainer.callback_metrics)}, object_dict


def _run(cfg) -> Optional[float]:
    extras(cfg)
    metric_dict, objects = train(cfg)
    log.info("metrics: " + ", ".join(f"{k}={float(v):.5f}" for k, v in sorted(metric_dict.items())))
    if os.environ.get("MEDMOE_LOG_PARAM_HASH") == "1":      # replica-drift check under data parallelism: every rank prints its own digest
        import hashlib
        model = objects.get("model")
        if model is not None:
            h = hashlib.sha256()
            for p_ in model.parameters():
                h.update(p_.detach().float().cpu().numpy().tobytes())
            print(f"[rank {int(os.environ.get('RANK', '0'))}] param sha256 {h.hexdigest()[:16]} global_step {objects['trainer'].global_step}", flush=True)
    return get_metric_value(metric_dict=metric_dict, metric_name=cfg.get("optimized_metric"))


if HAVE_HYDRA:                                           # pragma: no cover
    main = hydra.main(version_base="1.3", config_path="../configs", config_name="train.yaml")(_run)
else:
    def main(argv=None) -> Optional[float]:
        argv = sys.argv[1:] if argv is None else argv
        out_dir = os.path.join(ROOT, "logs", "train", "runs", "latest")
        cfg = compose(os.path.join(ROOT, "configs"), "train.yaml", overrides=argv, output_dir=out_dir)
        import logging
        logging.basicConfig(level=logging.INFO, format="%(message)s")
        return _run(cfg)


if __name__ == "__main__":
    main()
