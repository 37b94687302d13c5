import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
M = 201728
for (N, K, name) in [(3072, 768, "mul_aux"), (3072, 768, "gelu_daux"), (3072, 768, "none")]:
    a = torch.randn(M, K, device="cuda").to(bf); b = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf); aux = torch.randn(M, N, device="cuda").to(bf); bias = torch.randn(N, device="cuda")
    def run():
        if name == "gelu_daux": ops.gemm_nt(a, b, c, bias=bias, aux=aux, epi=ops.EPI_GELU_DAUX)
        elif name == "mul_aux": ops.gemm_nt(a, b, c, aux=aux, epi=ops.EPI_MUL_AUX)
        else: ops.gemm_nt(a, b, c)
    res = []
    for (o7, o2, label) in ((1, 1, "nt4w"), (0, 1, "nt512"), (0, 0, "nt256")):
        ops.set_option(7, o7); ops.set_option(2, o2)
        res.append(f"{label} {timeit(run)*1e3:.0f} us")
    ops.set_option(7, 1); ops.set_option(2, 1)
    print(N, K, name, " | ".join(res), flush=True)
