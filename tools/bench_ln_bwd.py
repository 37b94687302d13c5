"""LayerNorm backward timing at one rank's row counts, alone on the GPU (in the step at per-rank batch 128 the same launch takes 113 us instead of 47: it shares the chip with the second stream's wgrads)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from medmoe_amd import ops
    D = 768
    for rows in (25216, 50432, 201728):
        dy = torch.randn(rows, D, device="cuda").bfloat16(); x = torch.randn(rows, D, device="cuda").bfloat16(); add = torch.randn(rows, D, device="cuda").bfloat16()
        mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda"); g = torch.ones(D, device="cuda")
        dx = torch.empty_like(x); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
        f = lambda: ops.layernorm_bwd(dy, x, mean, rstd, g, dx, dg, db, add)
        for _ in range(5): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        b.record(); torch.cuda.synchronize()
        print(f"rows {rows:6d}: {a.elapsed_time(b) / 50 * 1000:7.1f} us ({4 * rows * D * 2 / (a.elapsed_time(b) / 50 * 1e-3) / 1e12:.2f} TB/s)", flush=True)
else:
    subprocess.run([sys.executable, __file__, "child"], check=True)
