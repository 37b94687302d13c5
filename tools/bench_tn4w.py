#!/usr/bin/env python3
"""Measurement aid: gemm_tn4w (option 8) against gemm_tn512 and the vendor library on the step's wgrad shapes; exact check
on small-integer operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lib = ops.load_library()
bf = torch.bfloat16
for (M, Nn, Kk) in [(4096, 256, 256), (8192, 512, 768), (25216, 2304, 768), (201728, 2304, 768), (201728, 768, 768), (201728, 3072, 768), (201728, 768, 3072), (25216, 3072, 768)]:
    g = torch.randint(-3, 4, (M, Nn), device="cuda").to(bf); x = torch.randint(-3, 4, (M, Kk), device="cuda").to(bf)
    outs = []
    for opt in (0, 1):
        lib.medmoe_set_option(8, opt)
        dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
        ops.gemm_tn(g, x, dw, db=db)
        torch.cuda.synchronize()
        outs.append((dw, db))
    exact = True
    if M <= 32768:      # integers: every partial sum is exact in fp32
        ref = g.float().t() @ x.float()
        exact = bool(torch.equal(outs[1][0], ref)) and bool(torch.equal(outs[1][1], g.float().sum(0)))
    same = bool(torch.equal(outs[0][0], outs[1][0])) and bool(torch.equal(outs[0][1], outs[1][1]))
    g = torch.randn(M, Nn, device="cuda").to(bf); x = torch.randn(M, Kk, device="cuda").to(bf)
    dw = torch.zeros(Nn, Kk, device="cuda"); db = torch.zeros(Nn, device="cuda")
    ts = []
    for opt in (0, 1):
        lib.medmoe_set_option(8, opt)
        ts.append(timeit(lambda: ops.gemm_tn(g, x, dw, db=db)))
    lib.medmoe_set_option(8, 0)
    out = torch.empty(Nn, Kk, device="cuda", dtype=bf)
    tl = timeit(lambda: torch.matmul(g.t(), x, out=out))
    f = 2 * M * Nn * Kk / 1e9
    print(f"tn {M}x{Nn}x{Kk}: tn512 {f/ts[0]:.0f} | tn4w {f/ts[1]:.0f} | library (bf16 out, no bias grad) {f/tl:.0f} TF/s   exact {exact} equal-to-tn512 {same}", flush=True)
