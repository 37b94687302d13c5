#!/usr/bin/env python3
"""CPU-side issue time of one training step (no device sync inside the timed span except the one the step itself
needs for the ragged-layout metadata) vs the device time, at a small per-rank batch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = config_by_name("cfg2")
eng = Engine(cfg, "cuda:0", seed=0)
batch = bench.synthetic_batch(cfg, B, 1, eng.device)
for _ in range(3): eng.train_step(batch)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    eng.train_step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: issue {1e3*(t1-t0):.1f} ms, until device idle {1e3*(t2-t0):.1f} ms", flush=True)
