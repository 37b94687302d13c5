"""Host-side profile of the fused Swin step (cProfile over a few steps): where the Python / launch time of a 17 ms step goes.
python tools/host_profile_swin.py [batch]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PROJECT_ROOT", ROOT)
import bench  # noqa: E402
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
N = 5
t0 = time.perf_counter()
for _ in range(N):
    lit.training_step(mb, 0)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={B}: host returns after {t_host / N * 1e3:.2f} ms/step, device done after {t_all / N * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    lit.training_step(mb, 0)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
