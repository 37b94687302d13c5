#!/usr/bin/env python3
"""MEDMOE_GRAPH=1 (forward and backward replayed from hipGraphs) against the eager step: same losses and parameters after the same steps, and the
step time of both.  usage: graph_check.py [batch] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
res = {}
for mode in ("0", "1"):
    os.environ["MEDMOE_GRAPH"] = mode
    eng = Engine(config_by_name(name), "cuda:0", seed=0)
    batches = [bench.synthetic_batch(eng.cfg, B, 100 + i, eng.device) for i in range(3)]
    losses = []
    for i in range(6):
        losses.append(float(eng.train_step(batches[i % 3])["loss"]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        eng.train_step(batches[i % 3])
    torch.cuda.synchronize()
    res[mode] = (losses, eng.params.p32.clone(), (time.perf_counter() - t0) / 20 * 1e3)
    del eng
    torch.cuda.empty_cache()
l0, p0, t0_ = res["0"]; l1, p1, t1_ = res["1"]
print("eager losses ", [round(v, 5) for v in l0])
print("graph losses ", [round(v, 5) for v in l1])
d = float((p1 - p0).norm() / p0.norm())
print(f"batch {B} {name}: eager {t0_:.2f} ms/step, graphs {t1_:.2f} ms/step; parameters after 26 steps differ by {d:.2e} (relative)")
# two EAGER runs already differ by ~1e-4 per step at cfg2 (fp32 atomics order) and drift apart over the steps: the bars only catch a wrong replay
assert all(abs(a - b) < 5e-3 * max(1.0, abs(a)) for a, b in zip(l0, l1)), "losses differ"
assert d < 2e-2
print("graph replay OK")
