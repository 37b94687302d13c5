#!/usr/bin/env python3
"""Step time of cfg2 for several row-range sizes of the grouped wgrad kernel (medmoe_set_option key 4)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import load_library
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine
import bench
cfg = config_by_name("cfg2")
eng = Engine(cfg, "cuda:0", seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = bench.synthetic_batch(cfg, B, 1, eng.device)
lib = load_library()
for rows in (256, 512, 1024, 2048, 4096):
    lib.medmoe_set_option(ctypes.c_int(4), ctypes.c_int(rows))
    for _ in range(2): eng.train_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): eng.train_step(batch)
    torch.cuda.synchronize()
    print(rows, f"{(time.perf_counter() - t0) / 4 * 1e3:.1f} ms/step", flush=True)
