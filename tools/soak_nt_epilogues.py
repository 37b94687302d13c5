"""Soak check of the gemm_nt4w row-major epilogues (inline-asm operand loads behind counted vmcnt waits): random full-tile shapes, the three
builds that take that path, bit-compared with gemm_nt512 (option 7 = 0), every launch repeated back to back so that stores of one launch's
last tiles are in flight under the next."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops

bf = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
bad = 0
for it in range(40):
    M = 256 * int(torch.randint(1, 120, (1,), device="cuda", generator=g))
    N = [768, 1024, 2304, 3072][it % 4]
    K = [768, 3072, 128, 1536][(it // 4) % 4]
    a = torch.randn(M, K, device="cuda", generator=g).to(bf); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(bf)
    aux = torch.randn(M, N, device="cuda", generator=g).to(bf); res = torch.randn(M, N, device="cuda", generator=g).to(bf)
    bias = torch.randn(N, device="cuda", generator=g)
    for name in ("bias_res", "mul_aux", "gelu_daux", "res"):
        outs = []
        for opt in (1, 0):
            ops.set_option(7, opt)
            c = torch.empty(M, N, device="cuda", dtype=bf); x = aux.clone()
            for _ in range(3):
                if name == "bias_res": ops.gemm_nt(a, b, c, bias=bias, residual=res)
                elif name == "res": ops.gemm_nt(a, b, c, residual=res)
                elif name == "mul_aux": ops.gemm_nt(a, b, c, aux=x, epi=ops.EPI_MUL_AUX)
                else: ops.gemm_nt(a, b, c, bias=bias, aux=x, epi=ops.EPI_GELU_DAUX)
            torch.cuda.synchronize()
            outs.append((c, x))
        ops.set_option(7, 1)
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        if not same:
            bad += 1
            print("MISMATCH", M, N, K, name, float((outs[0][0].float() - outs[1][0].float()).abs().max()), flush=True)
print("soak: 160 cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
