"""Drop-in mirror of the reference's `src` package for the hot path only (Hydra `_target_`
strings such as src.losses.GLORIAGlobalContrastiveLoss resolve here).  Everything computes
through medmoe_amd's C-ABI HIP kernels; there is no eager fallback."""
