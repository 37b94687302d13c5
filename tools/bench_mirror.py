"""The reference-named path a Hydra user runs - MedMoEPretrainingLightningModule.training_step -> backward -> clip -> torch Adam - timed next
to the fused Engine.train_step on the same geometry (cfg2 by default).  python tools/bench_mirror.py [batch] [steps] [arch]
arch = swin_t: the reference's own image encoder (Swin-T + pyramid experts, 3136 local regions; text max_length 25 and six experts as
in its configs) - there is no fused engine step for it, only the module path is timed."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medmoe_amd.engine import Engine                                                      # noqa: E402
from src.losses import GLORIAGlobalContrastiveLoss, GLORIALocalContrastiveLoss           # noqa: E402
from src.models.components.med_moe import MedMoE                                          # noqa: E402
from src.models.medmoe_module import MedMoEPretrainingLightningModule                     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
arch = sys.argv[3] if len(sys.argv) > 3 else "vit_b16"
if arch == "swin_t":
    model = MedMoE({"arch": "swin_t", "num_experts": 6}, {"max_length": 25})
else:
    model = MedMoE({"arch": "vit_b16", "num_experts": 8, "top_k": 2}, {"max_length": 77})
cfg = model.cfg
loss_cfg = {"global_loss": GLORIAGlobalContrastiveLoss(), "local_loss": GLORIALocalContrastiveLoss(), "global_loss_weight": 0.5,
            "local_loss_weight": 0.5, "classifier_loss_weight": 2.0, "temp1": 4.0, "temp2": 5.0, "temp3": 10.0, "soft_label": False}
lit = MedMoEPretrainingLightningModule(model, loss_cfg, optimizer=lambda params: torch.optim.Adam(params, lr=5e-5, fused=os.environ.get("ADAM_FUSED") == "1"))
opt = lit.configure_optimizers()["optimizer"]
g = torch.Generator(device="cuda").manual_seed(0)
lens = torch.randint(8, cfg.max_len + 1, (B,), device="cuda", generator=g)
ids = torch.randint(3, cfg.vocab - 1, (B, cfg.max_len), device="cuda", generator=g)
pos = torch.arange(cfg.max_len, device="cuda")[None]
ids = torch.where(pos < lens[:, None] - 1, ids, torch.zeros_like(ids))
ids[:, 0] = 1
ids[torch.arange(B), lens - 1] = 2
batch = {"image": torch.randn(B, 3, cfg.img_size, cfg.img_size, device="cuda", generator=g).bfloat16(),
         "label": torch.randint(0, cfg.n_expert, (B,), device="cuda", generator=g),
         "caption": {"ids": ids, "attn_mask": (ids != 0).long(), "token_type": torch.zeros_like(ids)}}


def mirror_step():
    opt.zero_grad()
    loss = lit.training_step(batch, 0)
    loss.backward()
    torch.nn.utils.clip_grad_norm_([p for p in lit.parameters() if p.requires_grad], 0.25)
    opt.step()
    return loss


def timed(f, n):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


tm = timed(mirror_step, steps)
if arch == "swin_t":
    n_train = sum(p.numel() for p in lit.parameters() if p.requires_grad)
    print(f"swin_t, batch {B}: Lightning-module path {tm:.1f} ms/step ({B / tm * 1e3:.0f} pairs/s), {n_train / 1e6:.1f} M trainable parameters, "
          f"peak HBM {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
    sys.exit(0)
eb = {"image": batch["image"], "label": batch["label"], **batch["caption"]}
eng = model.engine
te = timed(lambda: eng.train_step(eb), steps)
print(f"batch {B}: Lightning-module path {tm:.1f} ms/step ({B / tm * 1e3:.0f} pairs/s), fused Engine.train_step {te:.1f} ms/step ({B / te * 1e3:.0f} pairs/s)")
