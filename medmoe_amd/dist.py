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


def gather_rows(t: torch.Tensor) -> torch.Tensor:
    """[B, ...] -> [W*B, ...] with rank r's rows at [r*B, (r+1)*B) (no gradient: frozen text features, caption lengths, similarities)."""
    t = t.contiguous()
    out = torch.empty((dist.get_world_size() * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if t.is_cuda and dist.get_backend() != "nccl":
        torch.cuda.synchronize()                                  # gloo reads the buffer from the host right away (tests)
    dist.all_gather_into_tensor(out, t)
    return out


def scatter_key_grads(d_all: torch.Tensor) -> torch.Tensor:
    """Sum over ranks of d_all [W*B,D]; returns this rank's [B,D] slice (reduce-scatter)."""
    W = dist.get_world_size()
    B = d_all.shape[0] // W
    out = torch.empty(B, d_all.shape[1], device=d_all.device, dtype=d_all.dtype)
    dist.reduce_scatter_tensor(out, d_all.contiguous())
    return out


def allreduce_mean_(flat_grad: torch.Tensor):
    if flat_grad.is_cuda and dist.get_backend() != "nccl":
        torch.cuda.synchronize()                                  # gloo reads the buffer from the host right away (tests)
    dist.all_reduce(flat_grad)
    flat_grad.div_(dist.get_world_size())
    return flat_grad


def label_offset(local_batch: int) -> int:
    """labels = B_loc*rank + arange(B_loc)  (losses.py:516-518)."""
    return local_batch * dist.get_rank()


class BucketedAllReduce:
    """Overlaps the gradient all-reduce with the hand-scheduled backward: the flat fp32 gradient buffer is
    cut into contiguous buckets in the order the backward FINISHES them (MoE/router first, ViT layers
    L-1..0, embeddings last); `ready(i)` launches an async all-reduce of bucket i on RCCL's stream (it
    waits for the kernels already enqueued on the compute stream), `finish()` joins them and averages.
    One message per ViT layer (7.1 M fp32 = 28 MB): large enough to run at link rate on the xGMI mesh,
    small enough that only the last bucket is exposed."""

    def __init__(self, flat_grad: torch.Tensor, boundaries):
        # boundaries: ascending element offsets [0, ..., numel]; bucket i = [b[i], b[i+1])
        b = list(boundaries)
        if b[0] != 0 or b[-1] != flat_grad.numel() or any(b[i] >= b[i + 1] for i in range(len(b) - 1)):
            raise ValueError("bucket boundaries must cover the flat gradient exactly once")
        self.flat, self.b, self.work = flat_grad, b, {}

    @property
    def n_buckets(self):
        return len(self.b) - 1

    def ready(self, i: int):
        if i in self.work:
            raise RuntimeError(f"bucket {i} reduced twice")
        if self.flat.is_cuda and dist.get_backend() != "nccl":
            # RCCL enqueues the collective behind the kernels already on the compute stream; gloo reads the buffer
            # from the host right away, so the producers have to be finished first (tests / tools/two_rank_gpu.py)
            torch.cuda.synchronize()
        self.work[i] = dist.all_reduce(self.flat[self.b[i]: self.b[i + 1]], async_op=True)

    def finish(self):
        if len(self.work) != self.n_buckets:
            missing = [i for i in range(self.n_buckets) if i not in self.work]
            raise RuntimeError(f"gradient buckets never reduced: {missing}")
        for w in self.work.values():
            w.wait()
        self.work = {}
        self.flat.div_(dist.get_world_size())
