#!/usr/bin/env python3
"""The DROP-IN path under data parallelism: two ranks sharing ONE GPU (gloo backend, CUDA tensors) build the reference-named
`MedMoEPretrainingLightningModule` from the Hydra tree (`experiment=pretraining_medmoe_cfg2`, `model.fused_step=true`) and run one
fused training step each on their half of a batch.  Checked against ONE process running the same module on the concatenated batch:
  * both ranks end the step with bit-identical parameters;
  * the gathered global loss (mean over ranks) is the one-process global loss;
  * with `model.loss.local_loss_global=true` every loss term and the averaged gradient the optimiser sees are the one-process ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

OVERRIDES = ["experiment=pretraining_medmoe_cfg2", "model.model.vision.config_name=tiny2", "model.loss.local_loss_global=true",
             "model.optimizer.lr=0.001"]


def build():
    from medmoe_amd.hydra_lite import compose, instantiate
    cfg = compose(os.path.join(ROOT, "configs"), "train.yaml", OVERRIDES)
    lit = instantiate(cfg.model)
    assert lit.fused_step and lit.model.engine.cfg.local_loss_global and lit.model.engine.cfg.lr == 1e-3
    lit.configure_optimizers()
    lit.configure_fused(cfg.trainer.accumulate_grad_batches, cfg.trainer.gradient_clip_val)
    assert lit.model.engine.cfg.clip == 0.25
    return lit


def module_batch(b):
    return {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"]}}


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    lit = build()
    eng = lit.model.engine
    assert eng.dist and eng.world == world
    full = bench.synthetic_batch(eng.cfg, 16, 777, eng.device)
    B = 16 // world
    mine = {k: v[rank * B:(rank + 1) * B].contiguous() for k, v in full.items()}
    cap = {}
    orig = eng.params.adam_step
    def spy(*a, **k):
        torch.cuda.synchronize(); cap["g"] = eng.params.g32.clone(); return orig(*a, **k)
    eng.params.adam_step = spy
    loss = lit.training_step(module_batch(mine), 0)
    out = {k: v.detach().clone() for k, v in lit.fused_training_step(module_batch(mine), optimizer_step=False).items()}   # losses after the update
    torch.cuda.synchronize()
    p = lit.model.weights.detach().clone()
    gl = [torch.zeros_like(p) for _ in range(world)]
    dist.all_gather(gl, p)
    first = torch.stack([loss.detach().float().reshape(())])
    allf = [torch.zeros_like(first) for _ in range(world)]
    dist.all_gather(allf, first)
    if rank == 0:
        ret["same_params"] = all(torch.equal(gl[0], g) for g in gl)
        ret["grad"] = cap["g"].cpu()
        ret["params"] = p.cpu()
        ret["loss"] = [float(v) for v in allf]
    dist.destroy_process_group()


def main():
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(2, 29531, ret), nprocs=2, join=True)
    import bench
    lit = build()
    eng = lit.model.engine
    assert not eng.dist
    full = bench.synthetic_batch(eng.cfg, 16, 777, eng.device)
    cap = {}
    orig = eng.params.adam_step
    def spy(*a, **k):
        torch.cuda.synchronize(); cap["g"] = eng.params.g32.clone(); return orig(*a, **k)
    eng.params.adam_step = spy
    p0 = lit.model.weights.detach().clone().cpu()
    loss1 = float(lit.training_step(module_batch(full), 0))
    torch.cuda.synchronize()
    g1, p1 = cap["g"].cpu(), lit.model.weights.detach().cpu()
    grad, params = ret["grad"], ret["params"]
    e_g = float((grad - g1).norm() / g1.norm())
    upd2, upd1 = params - p0, p1 - p0
    cos = float((upd2 * upd1).sum() / (upd2.norm() * upd1.norm()))
    print("two ranks' loss", ret["loss"], "one process", loss1, "| averaged gradient vs one process: rel", e_g, "| update cosine", cos)
    assert ret["same_params"]
    # router CE and the gathered global loss are per-rank means over the rank's own rows: their mean over ranks is the one-process value
    mean2 = sum(ret["loss"]) / len(ret["loss"])
    assert abs(mean2 - loss1) < 5e-3 * max(1.0, abs(loss1)), (ret["loss"], loss1)
    assert e_g < 2e-2, e_g
    assert cos > 0.98, cos                      # Adam's first step is sign-like: elements with |g| at the noise level may flip
    print("two-rank fused module path OK")


if __name__ == "__main__":
    main()
