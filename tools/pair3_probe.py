#!/usr/bin/env python3
"""Times the transposed pair stage (local_pair3, forward launch and backward launch) per caption length class at cfg2 / B = 1024
(196 regions; class sizes of lengths uniform in 8..77) on synthetic log-probabilities.  Prints ms per class and the totals, to be read
beside local_pair2's 47 ms + scale_blocks' 15 ms of profiles/r02_cfg2_gb1024_kernel_stats.csv."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medmoe_amd import ops  # noqa: E402

BF = torch.bfloat16


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    B = int(os.environ.get("PROBE_B", "1024"))
    HW, HWp, T, D, GR = 196, 208, 77, 768, 224
    dev = "cuda"
    torch.manual_seed(0)
    lens = torch.randint(8, 78, (B,))
    ctx = (torch.randn(B, HW, 64, device=dev) * 0.3)
    gm = torch.zeros(B, GR, GR, device=dev, dtype=BF)
    gm[:, :HW, :HW] = torch.bmm(ctx, ctx.transpose(1, 2)).to(BF)
    wn = torch.rand(B, T, device=dev) + 0.5
    lse = torch.randn(B, B, HWp, device=dev)
    gs = (torch.randn(B, B, device=dev) * 0.1).contiguous()
    sim = torch.empty(B, B, device=dev)
    ld = B * HWp
    tot_f = tot_b = 0.0
    for ntt in range(1, 6):
        members = torch.nonzero((lens + 15) // 16 == ntt).flatten().int().to(dev)
        n_c = int(members.numel())
        if n_c == 0:
            continue
        rows = n_c * 16 * ntt
        lp = (-(torch.randn(rows, B * 224, device=dev).abs() * 2.0)).to(torch.float16).view(BF)
        A = torch.empty(rows, ld, device=dev, dtype=BF); U = torch.empty_like(A); dS = torch.empty_like(A)
        capd = lens.int().to(dev)
        stats = torch.empty(B, rows, 2, device=dev)
        PW = int(os.environ.get("PROBE_PW", "224"))
        LD, BS = (PW, rows * PW) if os.environ.get("PROBE_IMAGE_MAJOR", "1") != "0" else (B * PW, PW)
        f = lambda: ops.call("local_pair3", lp, None, A, None, lse, gm, wn, capd, None, sim, None, stats, rows, B, B, HW, T, 4.0, 5.0, 1e-8, members, n_c, ntt, 0, LD, BS, PW, None)
        bw = lambda: ops.call("local_pair3", lp, dS, A, U, lse, gm, wn, capd, gs, sim, None, stats, rows, B, B, HW, T, 4.0, 5.0, 1e-8, members, n_c, ntt, 0, LD, BS, PW, None)
        tf, tb = timed(f), timed(bw)
        gb = rows * ld * 2 / 1e9
        print(f"class {ntt}: {n_c} captions, {rows} rows: fwd {tf:.2f} ms ({gb / tf * 1e3:.0f} GB/s of lp), bwd {tb:.2f} ms ({4 * gb / tb * 1e3:.0f} GB/s)", flush=True)
        tot_f += tf; tot_b += tb
        del lp, A, U, dS
    print(f"total: fwd {tot_f:.2f} ms, bwd {tot_b:.2f} ms, sum {tot_f + tot_b:.2f} ms")
    # the score GEMM of each class: transposed output (local_scores_t) beside the [region][word] output (local_scores_ragged)
    c16 = (torch.randn(B * HW, D, device=dev) * 0.2).to(BF); w16 = (torch.randn(B, T, D, device=dev) * 0.2).to(BF)
    capd = lens.int().to(dev)
    lse2 = torch.empty(B * HWp, B, device=dev)
    ts = tr = 0.0
    for ntt in range(1, 6):
        members = torch.nonzero((lens + 15) // 16 == ntt).flatten().int().to(dev)
        n_c = int(members.numel())
        if n_c == 0:
            continue
        rows = n_c * 16 * ntt
        lpT = torch.empty(rows, ld, device=dev, dtype=BF)
        lA = torch.empty(B * HWp, rows, device=dev, dtype=BF)
        lpT = torch.empty(rows, B * 224, device=dev, dtype=BF)
        t_t = timed(lambda: ops.call("local_scores_t", c16, w16, capd, lpT, lse, B, B, HW, T, D, members, n_c, ntt, 0, 224, rows * 224))
        t_r = timed(lambda: ops.call("local_scores_ragged", c16, w16, capd, lA, lse2, B, B, HW, T, D, members, n_c, ntt, 0, rows))
        ops.set_option(12, 1)
        t_n = timed(lambda: ops.call("local_scores_t", c16, w16, capd, lpT, lse, B, B, HW, T, D, members, n_c, ntt, 0, 224, rows * 224))
        ops.set_option(12, 0)
        print(f"scores class {ntt}: transposed {t_t:.2f} ms (k-loop alone {t_n:.2f}), [region][word] {t_r:.2f} ms", flush=True)
        ts += t_t; tr += t_r
        del lpT, lA
    print(f"scores total: transposed {ts:.2f} ms, [region][word] {tr:.2f} ms")


if __name__ == "__main__":
    main()
