#!/usr/bin/env python3
"""What would a forward / backward split of local_pair2 cost?  Times, per caption length class at cfg2 / B = 1024: the full
single-pass kernel (sim + gradients for dL/dsim = 1) and its forward-only mode (dS = NULL: sim only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from medmoe_amd import ops
from medmoe_amd.config import config_by_name
from medmoe_amd.engine import Engine, ragged_layout

cfg = config_by_name("cfg2"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine(cfg, "cuda:0"); batch = bench.synthetic_batch(cfg, B, 1, eng.device)
eng.train_step(batch, optimizer=False); torch.cuda.synchronize()
ws, c = eng.ws, cfg
P, T, Do, HWp, Tp = c.n_patch, c.max_len, c.d_out, eng.HWp, eng.Tp
perm, col, ntts, chunk, classes, Kc, Kp = ragged_layout(eng._cap_lens_host(), T, Tp)
meta = torch.from_numpy(np.concatenate((perm, col, 16 * ntts, chunk)).astype(np.int32)).to(eng.device)
d_perm = meta[:B]
rag = lambda name: ws[name].view(-1)[:B * HWp * Kp].view(B * HWp, Kp)
lA, ldS, lU = rag("l_A"), rag("l_dS"), rag("l_U")
ctx = ws["img_l"].view(B * P, Do)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)
tot_full = tot_fwd = 0.0
for ntt, start, n_c, cbase in classes:
    members = d_perm[start:start + n_c]
    scores = lambda: ops.call("local_scores_ragged", ctx, ws["words"], eng.cap_lens, lA, ws["l_lse"], B, B, P, T, Do, members, n_c, ntt, cbase, Kp)
    scores(); t_fwd = timed(lambda: ops.call("local_pair2_ragged", lA, ws["l_lse"], ws["gmp"], ws["wn"], eng.cap_lens, None, ws["sim"], None, None,
                                             B, B, P, T, c.temp1, c.temp2, 1e-8, members, n_c, ntt, cbase, Kp))
    t_full = timed(lambda: ops.call("local_pair2_ragged", lA, ws["l_lse"], ws["gmp"], ws["wn"], eng.cap_lens, None, ws["sim"], ldS, lU,
                                    B, B, P, T, c.temp1, c.temp2, 1e-8, members, n_c, ntt, cbase, Kp))
    print(f"class {ntt}: {n_c} captions  full {t_full:.2f} ms  forward-only {t_fwd:.2f} ms"); tot_full += t_full; tot_fwd += t_fwd
print(f"total full {tot_full:.2f} ms, forward-only {tot_fwd:.2f} ms")
