"""Per-shape GEMM times of the fused Swin step (ops.PROFILE: HIP events around every launch): which shapes land on which kernel.
python tools/swin_shape_times.py [batch]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)
import bench  # noqa: E402
from medmoe_amd import ops  # noqa: E402
from medmoe_amd.hydra_lite import compose, instantiate  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hc = compose(os.path.join(ROOT, "configs"), "train.yaml", ["experiment=pretraining_medmoe", "model.model.vision.arch=swin_t", "model.fused_step=true"])
lit = instantiate(hc.model)
lit.train(); lit.configure_optimizers(); lit.configure_fused(1, 0.25)
b = bench.synthetic_batch(lit.model.cfg, B, 1, lit.model.device)
b["label"] = b["label"] % 6
mb = {"image": b["image"], "label": b["label"], "caption": {"ids": b["ids"], "attn_mask": b["attn_mask"], "token_type": b["token_type"]}}
for _ in range(3):
    lit.training_step(mb, 0)
torch.cuda.synchronize()
ops.PROFILE = []
lit.training_step(mb, 0)
torch.cuda.synchronize()
rows = ops.PROFILE
ops.PROFILE = None
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for label, work, unit, e0, e1, detail in rows:
    ms = e0.elapsed_time(e1)
    tot += ms
    key = (label.split("(")[0].strip()[:40], tuple(detail[:4]) if detail else ())
    agg[key][0] += 1
    agg[key][1] += ms
print(f"{len(rows)} timed launches, {tot:.2f} ms of event time")
for (label, d), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{ms:7.3f} ms {n:4d} x {ms / n * 1e3:7.1f} us  {label:40s} {d}")
