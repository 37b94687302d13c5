#!/usr/bin/env python3
"""Per-shape time of every gemm_nt launch in one cfg2 training step (HIP events from ops.PROFILE)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine
import bench
cfg = config_by_name(sys.argv[1] if len(sys.argv) > 1 else "cfg2")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
eng = Engine(cfg, "cuda:0", seed=0)
batch = bench.synthetic_batch(cfg, B, 1, eng.device)
for _ in range(2): eng.train_step(batch)
prof = []; ops.PROFILE = prof
torch.cuda.synchronize(); eng.train_step(batch); torch.cuda.synchronize(); ops.PROFILE = None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for fl, e0, e1, key in prof:
    a = agg[key]; a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f"total gemm_nt {tot:.1f} ms")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[1]:8.2f} ms {100*a[1]/tot:5.1f}%  n={a[0]:3d}  {a[2]/a[1]/1e9:6.0f} TF/s  {key}")
