#!/usr/bin/env python3
"""Two ranks sharing ONE GPU (gloo backend, CUDA tensors): src.losses.contrastive_loss_with_temperature with the
embedding all-gather (reference losses.py:503-524, distributed.py:28-58) for BackpropType GLOBAL / LOCAL / NONE.

Expected values come from the CPU oracle's autograd on the concatenated batch:
  GLOBAL  every rank back-propagates its own loss; the all-gather's backward SUMS what all ranks sent to my slice
          -> d(sum_r loss_r) / d a_mine
  LOCAL   only my own slice of the gathered tensors carries gradient -> d loss_mine / d a_mine, other slices detached
  NONE    the gathered tensors carry no gradient -> d loss_mine / d a_mine through the query side only
Run by tests/test_parity2_gpu.py (RCCL refuses two ranks on one device; on the 8-GPU node the same code runs on "nccl")."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def inputs(world, B, D):
    g = torch.Generator().manual_seed(21)
    a = torch.randn(world * B, D, generator=g) * 0.3
    b = torch.randn(world * B, D, generator=g) * 0.3
    return a, b, torch.tensor(1.3)


def expected(world, B, D, rank, mode):
    import medmoe_oracle as O
    a, b, s = inputs(world, B, D)
    parts_a = [a[r * B:(r + 1) * B].clone().requires_grad_(True) for r in range(world)]
    parts_b = [b[r * B:(r + 1) * B].clone().requires_grad_(True) for r in range(world)]
    s = s.clone().requires_grad_(True)

    def gathered(parts, me):
        if mode == "GLOBAL":
            return torch.cat(parts)
        if mode == "LOCAL":
            return torch.cat([p if r == me else p.detach() for r, p in enumerate(parts)])
        return torch.cat([p.detach() for p in parts])

    outs = [O.contrastive_with_temperature(parts_a[r], parts_b[r], gathered(parts_a, r), gathered(parts_b, r), s, r)
            for r in range(world)]
    mine = outs[rank]
    # GLOBAL: the reduce-scatter adds every rank's contribution to my slice; the scale gradient stays per rank (DDP
    # would average it afterwards, outside this function)
    total = sum(o[0] for o in outs) if mode == "GLOBAL" else mine[0]
    ga, gb = torch.autograd.grad(total, [parts_a[rank], parts_b[rank]], retain_graph=True)
    gs, = torch.autograd.grad(mine[0], [s])
    return mine, ga, gb, gs


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from src.losses import contrastive_loss_with_temperature
    from src.utils.distributed import BackpropType
    B, D = 6, 32
    a, b, s0 = inputs(world, B, D)
    ok = True
    msgs = []
    for mode in ("GLOBAL", "LOCAL", "NONE"):
        al = a[rank * B:(rank + 1) * B].clone().cuda().requires_grad_(True)
        bl = b[rank * B:(rank + 1) * B].clone().cuda().requires_grad_(True)
        s = torch.nn.Parameter(s0.clone().cuda())
        out = contrastive_loss_with_temperature(al, bl, s, backprop_type=getattr(BackpropType, mode))
        out.loss.backward()
        torch.cuda.synchronize()
        (lo, la, lb, loa, lob), ga, gb, gs = expected(world, B, D, rank, mode)

        def rel(x, y):
            return float((x.detach().cpu().float() - y).norm() / y.norm().clamp_min(1e-12))
        errs = {"loss": abs(float(out.loss) - float(lo)), "logits_a": rel(out.logits_a, la.detach()), "logits_b": rel(out.logits_b, lb.detach()),
                "loss_a": abs(float(out.loss_a) - float(loa)), "loss_b": abs(float(out.loss_b) - float(lob)),
                "da": rel(al.grad, ga), "db": rel(bl.grad, gb), "ds": abs(float(s.grad) - float(gs)) / max(1.0, abs(float(gs)))}
        bad = {k: v for k, v in errs.items() if v > 2e-5}
        msgs.append(f"rank {rank} {mode}: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
        if bad:
            ok = False
        if tuple(out.logits_a.shape) != (B, world * B):
            ok = False
    ret[rank] = (ok, msgs)
    dist.barrier()
    dist.destroy_process_group()


def main():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(2, 29531, ret), nprocs=2, join=True)
    good = True
    for r in range(2):
        ok, msgs = ret[r]
        print("\n".join(msgs))
        good = good and ok
    assert good, "two-rank contrastive mismatch"
    print("two-rank contrastive OK")


if __name__ == "__main__":
    main()
