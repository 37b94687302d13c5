"""Data-parallel exchange steps of the hot path (torch.distributed plumbing; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The reference's only collective call sites are src/utils/distributed.py:28-58 (gather_tensor,
used by losses.py:503-524) plus DDP's gradient all-reduce (configs/trainer/ddp.yaml:4).  Here:
  * ONE all-gather of cat([img_g, txt_g]) per step instead of two (losses.py:512-513) - the
    message is latency-bound (<= 0.8 MB/rank), so one hop on the xGMI mesh instead of two;
  * the backward of the gathered keys is ONE reduce-scatter (BackpropType.GLOBAL semantics of
    torch.distributed.nn.functional.all_gather: every rank's gradient for my slice is summed);
  * gradients: one all-reduce over the single flat fp32 gradient buffer (ParamStore.g32).
"""
import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def gather_embeddings(img_g: torch.Tensor, txt_g: torch.Tensor):
    """[B,D],[B,D] -> (img_all [W*B,D], txt_all [W*B,D]) with rank r's rows at [r*B,(r+1)*B)."""
    B, D = img_g.shape
    W = dist.get_world_size()
    packed = torch.cat([img_g, txt_g], dim=1).contiguous()
    out = torch.empty(W * B, 2 * D, device=packed.device, dtype=packed.dtype)
    dist.all_gather_into_tensor(out, packed)
    return out[:, :D].contiguous(), out[:, D:].contiguous()


def scatter_key_grads(d_all: torch.Tensor) -> torch.Tensor:
    """Sum over ranks of d_all [W*B,D]; returns this rank's [B,D] slice (reduce-scatter)."""
    W = dist.get_world_size()
    B = d_all.shape[0] // W
    out = torch.empty(B, d_all.shape[1], device=d_all.device, dtype=d_all.dtype)
    dist.reduce_scatter_tensor(out, d_all.contiguous())
    return out


def allreduce_mean_(flat_grad: torch.Tensor):
    dist.all_reduce(flat_grad)
    flat_grad.div_(dist.get_world_size())
    return flat_grad


def label_offset(local_batch: int) -> int:
    """labels = B_loc*rank + arange(B_loc)  (losses.py:516-518)."""
    return local_batch * dist.get_rank()
