#!/usr/bin/env python3
"""Swin-T tower (medmoe_amd/swin.py) forward + backward time per batch on the GPU, random weights of the published geometry."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmoe_amd.swin import SwinTower  # noqa: E402


def main():
    from transformers import SwinConfig, SwinModel
    B = int(os.environ.get("SWIN_B", "128"))
    torch.manual_seed(0)
    tower = SwinTower(SwinModel(SwinConfig()).state_dict(), "cuda")
    x = torch.randn(B, 3, 224, 224, device="cuda").to(torch.bfloat16)
    out = tower.forward(x)
    d_hs = [torch.randn_like(h) * 0.01 for h in out["hidden_states"]]
    d_last = torch.randn_like(out["last_hidden_state"]) * 0.01
    tower.backward(d_hs, d_last)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        tower.forward(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        tower.forward(x); tower.backward(d_hs, d_last)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fwd, both = (t1 - t0) / n * 1e3, (t2 - t1) / n * 1e3
    gf = 4.5e9 * B          # published forward FLOPs of Swin-T at 224 x 224 (multiply-adds counted as two)
    print(f"batch {B}: forward {fwd:.2f} ms ({gf / fwd / 1e9:.0f} TFLOP/s), forward + backward {both:.2f} ms ({3 * gf / both / 1e9:.0f} TFLOP/s), "
          f"{B / both * 1e3:.0f} images/s, peak HBM {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")


if __name__ == "__main__":
    main()
