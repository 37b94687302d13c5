#!/usr/bin/env python3
"""Host time to ENQUEUE one training step (train_step returns when its launches are queued) against the step's GPU time: tells whether the
host can keep ahead of the device at small per-rank batches.  usage: host_enqueue_time.py [batch] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = config_by_name(sys.argv[2] if len(sys.argv) > 2 else "cfg2")
eng = Engine(cfg, "cuda:0")
batch = bench.synthetic_batch(cfg, B, 1, eng.device)
for _ in range(3):
    eng.train_step(batch)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.train_step(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"batch {B}: host enqueue median {host[5]:.2f} ms (min {host[0]:.2f}), step from an idle queue {total[5]:.2f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng.train_step(batch)
torch.cuda.synchronize()
print(f"back to back: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step")
