"""world_size-2 gloo test (CPU) of the data-parallel exchange steps in medmoe_amd/dist.py:
the all-gather + local-rows contrastive loss with a reduce-scatter of the key gradients must
equal the single-process GLoRIA global loss/gradient on the concatenated batch (the identity
SURVEY.md 8e asks for), and the flat-gradient all-reduce must average."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _cos_ce(a_loc, b_all, off, temp):
    import torch.nn.functional as F
    s = (a_loc @ b_all.t()) / (a_loc.norm(dim=1, keepdim=True) * b_all.norm(dim=1)[None]).clamp(min=1e-8) * temp
    return F.cross_entropy(s, off + torch.arange(a_loc.shape[0]))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import medmoe_oracle as O
    from medmoe_amd import dist as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    Bg, Dm = 12, 16
    img = torch.randn(Bg, Dm); txt = torch.randn(Bg, Dm)
    B = Bg // world
    sl = slice(rank * B, (rank + 1) * B)
    # single-process reference on the concatenated batch
    img_ref = img.clone().requires_grad_(True)
    loss_ref = O.gloria_global(img_ref, txt, 10.0)
    loss_ref.backward()
    # sharded: what Engine.forward_backward_losses does for world > 1
    assert D.is_dist() and D.label_offset(B) == rank * B
    il = img[sl].clone().requires_grad_(True)
    tl = txt[sl].clone()
    img_all, txt_all = D.gather_embeddings(il.detach(), tl)
    assert torch.equal(img_all, img) and torch.equal(txt_all, txt)
    img_all = img_all.requires_grad_(True)
    loss = _cos_ce(il, txt_all, rank * B, 10.0) + _cos_ce(tl, img_all, rank * B, 10.0)
    loss.backward()
    d_img = il.grad + D.scatter_key_grads(img_all.grad)         # direct rows + every rank's key gradients
    flat = torch.cat([d_img.reshape(-1), loss.detach().reshape(1)])
    # DDP semantics: each rank back-propagates d_img into ITS samples; the averaged objective is mean_r loss_r
    lsum = loss.detach().clone(); dist.all_reduce(lsum)
    ok_loss = torch.allclose(lsum / world, loss_ref.detach(), atol=1e-5)
    ok_grad = torch.allclose(d_img / world, img_ref.grad[sl], atol=1e-5)
    g = torch.full((5,), float(rank + 1)); D.allreduce_mean_(g)
    ok_ar = torch.allclose(g, torch.full((5,), (1 + world) / 2))
    # bucketed / overlapped reducer: out-of-order readiness, exact coverage, mean
    flat = torch.arange(20, dtype=torch.float32) * (rank + 1)
    red = D.BucketedAllReduce(flat, [0, 3, 11, 20])
    for i in (2, 0, 1):
        red.ready(i)
    red.finish()
    ok_ar = ok_ar and torch.allclose(flat, torch.arange(20, dtype=torch.float32) * (1 + world) / 2)
    try:
        D.BucketedAllReduce(flat, [0, 5, 5, 20]); ok_ar = False
    except ValueError:
        pass
    red2 = D.BucketedAllReduce(flat, [0, 10, 20]); red2.ready(0)
    try:
        red2.finish(); ok_ar = False
    except RuntimeError:
        red2.ready(1); red2.finish()
    q.put((rank, bool(ok_loss), bool(ok_grad), bool(ok_ar)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gathered_global_loss_identity_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[2] and r[3], r
