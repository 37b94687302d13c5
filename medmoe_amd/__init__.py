"""medmoe_amd: MI355X-native (gfx950) implementation of MedMoE's contrastive fwd/bwd hot path.

All compute goes through the C-ABI HIP library ``medmoe_amd/lib/libmedmoe_hip.so``
(declared in ``include/medmoe_hip.h``); there is no CPU or eager-PyTorch fallback.
"""
from ._lib import lib_path, load_library  # noqa: F401
