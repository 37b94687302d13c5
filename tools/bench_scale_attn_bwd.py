#!/usr/bin/env python3
"""scale_attn_bwd_kernel alone on an engine's own buffers (after one training step), rows per wave swept through medmoe_set_option(13).
usage: bench_scale_attn_bwd.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from medmoe_amd import ops
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = config_by_name("cfg2")
eng = Engine(cfg, "cuda:0")
eng.train_step(bench.synthetic_batch(cfg, B, 1, eng.device), optimizer=False)
c, p, ws = eng.cfg, eng.params, eng.ws
k, P, Do, Dh, R = c.top_k, c.n_patch, c.d_out, c.d_out // 2, eng.R


def run():
    ops.call("scale_attn_bwd", ws["d_img_l"], ws["d_img_g"], ws["G"], ws["H1"], ws["wts"], p.f32("moe.attn2.weight"),
             ws["eout"], ws["expert_of_slot"], ws["item_of_slot"], ws["gates"], k, P, ws["dG"], ws["dH1"],
             p.grad("moe.attn2.weight"), p.grad("moe.attn2.bias"), ws["dgate"], R, Do, Dh)


bytes_ = 2.0 * R * (4 * (2 * Do + 2 * Dh) + 2 * Do)
for rows in (0, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96):
    ops.set_option(13, rows)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"R {R} rows/wave {rows:3d}: {us:8.1f} us  {bytes_ / us / 1e6:6.2f} TB/s")
