"""TEST INFRASTRUCTURE ONLY - never imported by the product path.

Loads the reference's own importable components from /root/reference (read-only,
exists only in the build container; never on the GPU box) so that
``oracle/gen_golden.py`` can emit golden vectors.  Recipe follows SURVEY.md section 8c:
  * ``src.utils`` is pre-registered as a stub package so its ``__init__`` (hydra /
    lightning imports) does not run;
  * ``swin.py`` is loaded by file path with stub ``open_clip`` modules (unused imports);
  * ``transformer.py`` needs a stub ``torchvision.ops.stochastic_depth.StochasticDepth``
    (only used when drop_path_rate is set) - installed AFTER transformers is imported.
"""
import importlib.util
import sys
import types

REF = "/root/reference"


def load():
    import torch
    from torch import nn

    if REF not in sys.path:
        sys.path.insert(0, REF)
    # stub package for src.utils (skip template glue)
    if "src.utils" not in sys.modules or not hasattr(sys.modules["src.utils"], "__path__"):
        import src  # noqa: F401  (reference's empty package)
        pkg = types.ModuleType("src.utils")
        pkg.__path__ = [REF + "/src/utils"]
        sys.modules["src.utils"] = pkg
    # swin.py first (pulls transformers) with open_clip stubs
    for name in ("open_clip", "open_clip.transformer"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.VisionTransformer = object
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location(
        "ref_swin", REF + "/src/models/components/swin.py")
    swin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(swin)
    # torchvision stub (after transformers import)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tv.__path__ = []
        ops = types.ModuleType("torchvision.ops")
        ops.__path__ = []
        sd = types.ModuleType("torchvision.ops.stochastic_depth")

        class StochasticDepth(nn.Module):
            def __init__(self, p, mode):
                super().__init__()
                self.p = p

            def forward(self, x):
                return x
        sd.StochasticDepth = StochasticDepth
        ops.stochastic_depth = sd
        tv.ops = ops
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.ops"] = ops
        sys.modules["torchvision.ops.stochastic_depth"] = sd
    import src.losses as losses
    import src.utils.distributed as dist
    import src.models.components.transformer as transformer
    import src.models.components.multi_head_attention as mha
    import src.models.components.multimodal_transformer as mmt
    import src.models.components.text_encoder as text_encoder
    return types.SimpleNamespace(
        swin=swin, losses=losses, dist=dist, transformer=transformer, mha=mha,
        mmt=mmt, text_encoder=text_encoder, torch=torch)
