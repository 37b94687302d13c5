#!/usr/bin/env python3
"""The reference's own model (Swin-T + pyramid experts) in the fused step under data parallelism: two ranks sharing ONE GPU (gloo backend, CUDA
tensors) build the Hydra module (`vision.arch=swin_t`, `model.fused_step=true`) and step on their halves of a batch.  Checked:
  * both ranks end the step with bit-identical parameters (tower arena and MoE arena);
  * what the optimiser sees is the mean over ranks of the two local gradient arenas (captured around the all-reduces);
  * the gathered global loss, averaged over ranks, is the one-process global loss on the concatenated batch (rows = my images against ALL
    captions; the local loss stays rank-local, as the reference's does, losses.py:961-1026)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

OVERRIDES = ["experiment=pretraining_medmoe", "model.model.vision.arch=swin_t", "model.fused_step=true", "model.model.vision.num_experts=3",
             "model.model.text.n_layer=2", "model.optimizer.lr=0.0005"]
NB = 8


def build():
    from medmoe_amd.hydra_lite import compose, instantiate
    cfg = compose(os.path.join(ROOT, "configs"), "train.yaml", OVERRIDES)
    lit = instantiate(cfg.model)
    lit.train(); lit.configure_optimizers(); lit.configure_fused(1, 0.25)
    lit.model.swin.drop_path_rate = 0.0
    return lit


def batch(lit, lo, hi):
    import bench
    b = bench.synthetic_batch(lit.model.cfg, NB, 4242, lit.model.device)
    b["label"] = b["label"] % 3
    b = {k: v[lo:hi].contiguous() for k, v in b.items()}
    return {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"], "token_type": b["token_type"]}}


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lit = build()
    assert lit.model.engine.dist and lit.model.engine.world == world
    seen = []
    orig = dist.all_reduce

    def spy(t, *a, **k):
        torch.cuda.synchronize()
        if t.numel() > 100000:
            seen.append(t.detach().clone())
        return orig(t, *a, **k)
    dist.all_reduce = spy
    B = NB // world
    out = lit.fused_training_step(batch(lit, rank * B, (rank + 1) * B))
    torch.cuda.synchronize()
    dist.all_reduce = orig
    enc = lit._swin_engine.enc
    assert len(seen) == 2 and seen[0].numel() == enc.store.numel and seen[1].numel() == enc.tower.store.numel
    after = [enc.store.g32.detach().clone(), enc.tower.store.g32.detach().clone()]
    ok_mean = True
    for local, avg in zip(seen, after):
        parts = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(parts, local)
        want = (parts[0] + parts[1]) / world
        ok_mean = ok_mean and bool(torch.allclose(avg, want, rtol=1e-6, atol=1e-9))
    same = True
    for p in (enc.store.p32, enc.tower.store.p32):
        parts = [torch.zeros_like(p) for _ in range(world)]
        dist.all_gather(parts, p)
        same = same and torch.equal(parts[0], parts[1])
    g = torch.stack([out["g_loss"].detach().float().reshape(())])
    gl = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gl, g)
    if rank == 0:
        ret["same_params"], ret["mean_grad"], ret["g_loss"] = same, ok_mean, [float(v) for v in gl]
        ret["moved"] = float(after[1].abs().max()) > 0
    dist.destroy_process_group()


def main():
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(2, 29537, ret), nprocs=2, join=True)
    lit = build()
    assert not lit.model.engine.dist
    one = lit.fused_training_step(batch(lit, 0, NB))
    r = dict(ret)
    g_one, g_two = float(one["g_loss"]), sum(r["g_loss"]) / 2
    print("two ranks:", r, "one process g_loss:", g_one)
    assert r["same_params"] and r["mean_grad"] and r["moved"]
    assert abs(g_one - g_two) < 2e-3 * abs(g_one), (g_one, g_two)
    print("TWO_RANK_SWIN_OK")


if __name__ == "__main__":
    main()
