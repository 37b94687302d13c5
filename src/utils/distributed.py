"""Mirror of reference src/utils/distributed.py:16-90 (same names, argument meaning, defaults)."""
from enum import Enum
from typing import List

import torch
from torch import Tensor


class BackpropType(Enum):
    """distributed.py:16-25."""
    GLOBAL = 0
    LOCAL = 1
    NONE = 2


def get_rank() -> int:
    """distributed.py:86-90."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return torch.distributed.get_rank()
    return 0


def gather_tensor(tensor: Tensor, backprop_type: BackpropType = BackpropType.GLOBAL) -> List[Tensor]:
    """distributed.py:28-58.  GLOBAL uses the differentiable all-gather (backward = sum of every
    rank's gradient for this rank's slice)."""
    world_size = torch.distributed.get_world_size()
    if backprop_type == BackpropType.GLOBAL:
        from torch.distributed.nn.functional import all_gather as all_gather_with_backprop
        return list(all_gather_with_backprop(tensor))
    tensor_all_gpus = [torch.zeros_like(tensor) for _ in range(world_size)]
    torch.distributed.all_gather(tensor_all_gpus, tensor)
    if backprop_type == BackpropType.LOCAL:
        tensor_all_gpus[get_rank()] = tensor
    return tensor_all_gpus


def concat_gather_all_gpu(tensor: Tensor, backprop_type: BackpropType = BackpropType.GLOBAL, dim: int = 0) -> Tensor:
    """distributed.py:61-83."""
    if not torch.distributed.is_available() or not torch.distributed.is_initialized():
        return tensor
    return torch.cat(gather_tensor(tensor, backprop_type), dim=dim)
