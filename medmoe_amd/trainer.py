"""Minimal stand-in for `lightning.pytorch.trainer.Trainer` so that `src/train.py experiment=pretraining_medmoe` runs in
images without Lightning (this one).  It implements what the reference's trainer config asks for
(configs/trainer/default.yaml, configs/experiment/pretraining_medmoe.yaml:20-25): epochs, gradient accumulation,
gradient-norm clipping, a validation pass per epoch feeding ReduceLROnPlateau (`monitor: val/loss`,
medmoe_module.py:148-169), checkpoint / early-stopping callbacks on `val/loss` (configs/callbacks/default.yaml).
With Lightning installed the config's `_target_` resolves to the real Trainer and this file is unused.

Data-parallel: one process per GPU started by torch.distributed.run; every rank reads its own shard of the data (the datamodule gets
`trainer.world_size` / `trainer.global_rank`).  Two training modes:
  * `model.fused_step: true` (experiments pretraining_medmoe_cfg1..4): `training_step` IS `Engine.train_step` - forward, losses, backward,
    embedding all-gather + reduce-scatter, per-layer gradient buckets all-reduced under the backward, fused clip + Adam; the trainer only
    tells the module its accumulation / clip settings and counts steps;
  * otherwise torch autograd + the configured torch optimizer: gradients of the module's parameters are flattened into one buffer and
    averaged with ONE all-reduce per optimiser step.
"""
import math
import os
from typing import Any, Dict, List, Optional

import torch


class Callback:
    def on_validation_end(self, trainer, module, metrics: Dict[str, float]):
        pass


class ModelCheckpoint(Callback):
    def __init__(self, dirpath: Optional[str] = None, filename: str = "epoch_{epoch:03d}", monitor: str = "val/loss", mode: str = "min",
                 save_last: bool = True, **_):
        self.dirpath, self.filename, self.monitor, self.mode, self.save_last = dirpath, filename, monitor, mode, save_last
        self.best_model_path, self.best_score = "", None

    def on_validation_end(self, trainer, module, metrics):
        if not self.dirpath or trainer.global_rank != 0:
            return
        os.makedirs(self.dirpath, exist_ok=True)
        state = {"state_dict": module.state_dict(), "epoch": trainer.current_epoch, "global_step": trainer.global_step}
        if hasattr(module, "on_save_checkpoint"):
            module.on_save_checkpoint(state)                          # fused step: Adam's moments (Lightning calls the same hook)
        if self.save_last:
            torch.save(state, os.path.join(self.dirpath, "last.ckpt"))
        score = metrics.get(self.monitor)
        if score is None:
            return
        better = self.best_score is None or (score < self.best_score if self.mode == "min" else score > self.best_score)
        if better:
            self.best_score = score
            self.best_model_path = os.path.join(self.dirpath, self.filename.format(epoch=trainer.current_epoch) + ".ckpt")
            torch.save(state, self.best_model_path)


class EarlyStopping(Callback):
    def __init__(self, monitor: str = "val/loss", patience: int = 3, mode: str = "min", min_delta: float = 0.0, **_):
        self.monitor, self.patience, self.mode, self.min_delta = monitor, patience, mode, min_delta
        self.best, self.bad = None, 0

    def on_validation_end(self, trainer, module, metrics):
        score = metrics.get(self.monitor)
        if score is None:
            return
        better = self.best is None or (score < self.best - self.min_delta if self.mode == "min" else score > self.best + self.min_delta)
        if better:
            self.best, self.bad = score, 0
        else:
            self.bad += 1
            if self.bad >= self.patience and trainer.current_epoch + 1 >= trainer.min_epochs:
                trainer.should_stop = True


class Trainer:
    def __init__(self, default_root_dir: Optional[str] = None, min_epochs: int = 1, max_epochs: int = 1, accelerator: str = "gpu",
                 devices: Any = 1, num_nodes: int = 1, strategy: Any = "auto", check_val_every_n_epoch: int = 1,
                 deterministic: bool = False, accumulate_grad_batches: int = 1, gradient_clip_val: Optional[float] = None,
                 sync_batchnorm: bool = False, limit_train_batches: Optional[int] = None, limit_val_batches: Optional[int] = None,
                 callbacks: Optional[List[Any]] = None, logger: Any = None, **unused):
        if accelerator not in ("gpu", "cuda", "auto"):
            raise RuntimeError(f"accelerator={accelerator}: the MI355X build has no CPU path (trainer=cpu / ddp_sim are for the reference)")
        self.default_root_dir, self.min_epochs, self.max_epochs = default_root_dir, min_epochs, max_epochs
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.gradient_clip_val = gradient_clip_val
        self.check_val_every_n_epoch = max(1, int(check_val_every_n_epoch))
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        self.callbacks = [c for c in (callbacks or []) if isinstance(c, Callback)]
        self.logger = logger
        self.callback_metrics: Dict[str, float] = {}
        self.current_epoch, self.global_step, self.should_stop = 0, 0, False
        dist = torch.distributed
        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.global_rank = dist.get_rank() if self.world_size > 1 else 0
        want = devices if isinstance(devices, int) else (len(devices) if isinstance(devices, (list, tuple)) else 1)
        if want * num_nodes != self.world_size and self.world_size == 1 and want > 1:
            raise RuntimeError(f"trainer.devices={want}: start one process per GPU with `python -m torch.distributed.run --nproc-per-node "
                               f"{want} src/train.py ...` (the stand-in trainer does not spawn ranks)")

    @property
    def checkpoint_callback(self):
        for c in self.callbacks:
            if isinstance(c, ModelCheckpoint):
                return c
        return None

    # ------------------------------------------------------------------------------------------------
    def _to_device(self, datamodule, module, batch):
        if hasattr(datamodule, "to_device_batch") and isinstance(batch.get("image"), list):
            dev = next(module.parameters()).device
            size = getattr(getattr(module, "model", None), "cfg", None)
            return datamodule.to_device_batch(batch, dev, size.img_size if size is not None else 224)
        return batch

    def _allreduce_grads(self, params):
        """Average the gradients over the ranks with ONE collective: flatten, all-reduce, scatter back (a parameter without a gradient
        on this rank contributes zeros, so every rank sends the same layout)."""
        if self.world_size == 1:
            return
        params = [p for p in params if p.requires_grad]
        if not params:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
        torch.distributed.all_reduce(flat)
        flat.div_(self.world_size)
        off = 0
        for p in params:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def fit(self, model, datamodule=None, ckpt_path: Optional[str] = None):
        if ckpt_path:
            ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
            model.load_state_dict(ckpt["state_dict"])
            if hasattr(model, "on_load_checkpoint"):
                model.on_load_checkpoint(ckpt)
        datamodule.trainer = self
        datamodule.setup("fit")
        opt_cfg = model.configure_optimizers()
        opt = opt_cfg["optimizer"] if isinstance(opt_cfg, dict) else opt_cfg
        sched = opt_cfg.get("lr_scheduler", {}).get("scheduler") if isinstance(opt_cfg, dict) else None
        monitor = opt_cfg.get("lr_scheduler", {}).get("monitor", "val/loss") if isinstance(opt_cfg, dict) else "val/loss"
        params = [p for g in opt.param_groups for p in g["params"]]
        acc = self.accumulate_grad_batches
        fused = bool(getattr(model, "fused_step", False))
        if fused:
            model.configure_fused(acc, self.gradient_clip_val)

        def optimizer_step():
            self._allreduce_grads(params)
            if self.gradient_clip_val:
                torch.nn.utils.clip_grad_norm_(params, self.gradient_clip_val)
            opt.step()
            opt.zero_grad()
            self.global_step += 1

        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            model.train()
            opt.zero_grad()
            run, n, pending = 0.0, 0, 0
            loader = datamodule.train_dataloader()
            n_batches = len(loader) if hasattr(loader, "__len__") else None
            if self.limit_train_batches is not None:
                n_batches = min(n_batches, self.limit_train_batches) if n_batches is not None else self.limit_train_batches
            for i, batch in enumerate(loader):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                dbatch = self._to_device(datamodule, model, batch)
                last_of_epoch = n_batches is not None and i + 1 == n_batches
                if fused:
                    # micro-batches of an accumulation window: zero the gradient on the first, step on the last (Lightning steps on
                    # the last batch of the epoch too when micro-batches remain)
                    first, step = pending == 0, (pending + 1 == acc) or last_of_epoch
                    out = model.fused_training_step(dbatch, optimizer_step=step, zero_grad=first, loss_scale=1.0 / acc)
                    loss = out["loss"] * acc                          # the engine reports the scaled loss
                    pending = 0 if step else pending + 1
                    if step:
                        self.global_step += 1
                else:
                    loss = model.training_step(dbatch, i)
                    (loss / acc).backward()
                    pending += 1
                    if pending == acc or last_of_epoch:
                        optimizer_step()
                        pending = 0
                run += float(loss.detach()); n += 1
            if pending and not fused:                                # a loader without a length: step on what is left
                optimizer_step()
            self.callback_metrics["train/loss"] = run / max(1, n)
            if (epoch + 1) % self.check_val_every_n_epoch == 0:
                metrics = self.validate(model, datamodule)
                if sched is not None and monitor in metrics:
                    sched.step(metrics[monitor])
                for c in self.callbacks:
                    c.on_validation_end(self, model, metrics)
            if self.should_stop and epoch + 1 >= self.min_epochs:
                break
        return self.callback_metrics

    @torch.no_grad()
    def validate(self, model, datamodule):
        model.eval()
        tot, n = 0.0, 0
        for i, batch in enumerate(datamodule.val_dataloader()):
            if self.limit_val_batches is not None and i >= self.limit_val_batches:
                break
            out = model.model_step(self._to_device(datamodule, model, batch))
            tot += float(out["loss"]); n += 1
        val = tot / n if n else math.nan
        if self.world_size > 1:
            t = torch.tensor([val], device=next(model.parameters()).device)
            torch.distributed.all_reduce(t)
            val = float(t) / self.world_size
        self.callback_metrics["val/loss"] = val
        return dict(self.callback_metrics)

    def test(self, model, datamodule=None, ckpt_path: Optional[str] = None):
        """Lightning's `trainer.test(ckpt_path=best)` evaluates THAT checkpoint, not the last weights."""
        if ckpt_path:
            model.load_state_dict(torch.load(ckpt_path, map_location="cpu", weights_only=True)["state_dict"])
        return self.validate(model, datamodule)
