import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from medmoe_amd import ops, load_library
lib = load_library()
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bf = torch.bfloat16
for (M, N, K) in [(50432, 3072, 768), (8192, 3072, 3072)]:
    a = torch.randn(M, K, device="cuda").to(bf); b = torch.randn(N, K, device="cuda").to(bf)
    c = torch.empty(M, N, device="cuda", dtype=bf)
    for dbg, name in [(0, "full sc1-stores"), (4, "full plain-stores"), (1, "no-epilogue"), (3, "loads only")]:
        lib.medmoe_set_option(ctypes.c_int(2), ctypes.c_int(dbg))
        ms = timeit(lambda: ops.gemm_nt(a, b, c))
        print(f"nt256 {M}x{N}x{K} {name}: {ms:.3f} ms  ({2*M*N*K/ms/1e9:.0f} TF/s equiv)", flush=True)
lib.medmoe_set_option(ctypes.c_int(2), ctypes.c_int(0))
