#!/usr/bin/env python3
"""Grouped / row-mapped wgrad kernel timing: expert-projection shape, balanced vs skewed groups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bf = torch.bfloat16
M, Nn, Kk, E = 401408, 768, 768, 8
g = torch.randn(M, Nn, device="cuda").to(bf); x = torch.randn(201728, Kk, device="cuda").to(bf)
ident = torch.arange(M, device="cuda", dtype=torch.int32) % 201728
gath = ((torch.arange(M, device="cuda") // 196 % 1024) * 197 + 1 + torch.arange(M, device="cuda") % 196).int()
dw = torch.zeros(E, Nn, Kk, device="cuda"); db = torch.zeros(E, Nn, device="cuda")
bal = torch.tensor([i * (M // E) for i in range(E + 1)], device="cuda", dtype=torch.int32)
skew = torch.tensor([0, M // 2, M // 2, M // 2, M, M, M, M, M], device="cuda", dtype=torch.int32)
fl = 2.0 * M * Nn * Kk
for opt in (1, 0):
  ops.set_option(8, opt)
  print("gemm_tn4w" if opt else "gemm_tn512", flush=True)
  for name, off, xm in (("balanced, no map", bal, None), ("balanced, identity-ish map", bal, ident), ("balanced, expert gather map", bal, gath),
                        ("2 of 8 groups, gather map", skew, gath), ("2 of 8 groups, no map", skew, None)):
    xx = x if xm is not None else torch.randn(M, Kk, device="cuda").to(bf)
    ms = timeit(lambda: ops.gemm_tn(g, xx, dw, db=db, x_rowmap=xm, row_off=off, n_groups=E, stride_w=Nn * Kk, stride_db=Nn, M=M))
    print(f"  {name:32s} {ms:.3f} ms {fl/ms/1e9:.0f} TF/s", flush=True)
ops.set_option(8, 1)
ms = timeit(lambda: ops.gemm_tn(g[:200704], x[:200704], dw[0], db=db[0]))
print(f"{'plain 200704 rows':32s} {ms:.3f} ms {fl/2/ms/1e9:.0f} TF/s")
