#!/usr/bin/env python3
"""Where does the bf16 engine's gradient leave the fp32 oracle's?  Compares the gradients of intermediate tensors
(d img_l, d img_g, d router_in, d stage features, d embedding output) for one seeded batch.
  python tools/grad_diag.py [cfg_name] [seed] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import medmoe_oracle as O
from test_parity2_gpu import make, to_dev, rel

name = sys.argv[1] if len(sys.argv) > 1 else "tinyL"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ocfg, cfg, p, batch, eng, vocab = make(name, B, seed=seed)
pr = {k: v.clone().requires_grad_(not k.startswith("text.")) for k, v in p.items()}
last, hs = O.vit_forward(batch["image"], pr, ocfg)
for h in hs: h.retain_grad()
last.retain_grad()
router_in = last[:, 1:, :].mean(dim=1); router_in.retain_grad()
feats = [hs[l][:, 1:, :] for l in ocfg.stage_layers()]
for f in feats: f.retain_grad()
img_g, img_l, probs, idx = O.moe_forward(feats, router_in, pr, ocfg.n_expert, ocfg.top_k)
img_g.retain_grad(); img_l.retain_grad(); probs.retain_grad()
with torch.no_grad():
    txt_l, txt_g, cap = O.text_tower(batch["ids"], batch["attn_mask"], batch["token_type"], pr, ocfg, vocab)
l0, l1, _ = O.gloria_local(img_l, txt_l, cap, ocfg.temp1, ocfg.temp2, ocfg.temp3)
gl = O.gloria_global(img_g, txt_g, ocfg.temp3)
cl = O.router_ce(probs, batch["label"])
loss = ocfg.w_local * (l0 + l1) + ocfg.w_global * gl + ocfg.w_cls * cl
loss.backward()
from medmoe_amd import ops as _ops
_snap = []
_orig_ln = _ops.layernorm_bwd
def _spy(dy, x, mean, rstd, gamma, dx, dgamma=None, dbeta=None, add=None):
    r = _orig_ln(dy, x, mean, rstd, gamma, dx, dgamma, dbeta, add=add)
    _snap.append(dx.float().clone())
    return r
_ops.layernorm_bwd = _spy
eng.train_step(to_dev(batch), optimizer=False)
_ops.layernorm_bwd = _orig_ln
torch.cuda.synchronize()
ws = eng.ws
P, Do, Nt, Dv, k = cfg.n_patch, cfg.d_out, cfg.n_tok_v, cfg.d_v, cfg.top_k
print("idx equal:", torch.equal(ws["idx"].cpu().long(), idx))
print("fwd img_l", rel(eng.outputs()["img_l"], img_l), "img_g", rel(ws["img_g"], img_g), "probs", rel(ws["probs"], probs))
print("d_img_g ", rel(ws["d_img_g"], img_g.grad))
print("d_img_l ", rel(ws["d_img_l"].float(), img_l.grad.reshape(B, Do, P).transpose(1, 2)))
print("d router_in", rel(ws["drouter_in"], router_in.grad), " |ref|", float(router_in.grad.norm()))
slot_of = ws["slot_of"].cpu().long().view(B, k)
for s in range(4):
    dF = ws["dF"][s].float().cpu().view(B * k, P, Dv)
    tot = dF[slot_of].sum(1)
    print(f"d feats[{s}]", rel(tot, feats[s].grad))
print("d x0 (after all layers)", rel(ws["dxa"].float().view(B, Nt, Dv), hs[0].grad) if cfg.n_layer_v % 2 == 0 else "n/a", rel(ws["dxb"].float().view(B, Nt, Dv), hs[0].grad))
# _snap[0] = final LN bwd (router path only, before the stage gradients join); then per layer L-1..0: [d xmid_l, d x_l]
L = cfg.n_layer_v
for l in range(L - 1, -1, -1):
    dxl = _snap[1 + 2 * (L - 1 - l) + 1].view(B, Nt, Dv)
    print(f"d hs[{l}]", rel(dxl, hs[l].grad), end="   ")
print()
got = eng.params.export_named(eng.params.g32)
errs = {kk: rel(got[kk].reshape(v.grad.shape), v.grad) for kk, v in pr.items() if not kk.startswith("text.") and v.grad is not None and v.grad.norm() > 1e-7}
import numpy as np
print("param grads: median", float(np.median(list(errs.values()))), "worst", sorted(errs.items(), key=lambda kv: -kv[1])[:6])
# loss-term split: which loss dominates d_img_l ?
for nm, term in (("local", ocfg.w_local * (l0 + l1)), ("global", ocfg.w_global * gl), ("cls", ocfg.w_cls * cl)):
    pass
# ---- gate gradient (top-k > 1): dgate[b, j] = <d out_b, expert_{idx[b,j]}(x_b)> ----
if cfg.top_k > 1:
    with torch.no_grad():
        Dref = img_l.grad.reshape(B, Do, P).transpose(1, 2) + img_g.grad[:, None, :] / P          # fp32 oracle d out
        Deng = ws["d_img_l"].float().cpu() + ws["d_img_g"].cpu()[:, None, :] / P                   # what the engine's kernel reads
        fe = [f.detach() for f in feats]
        dg_ref = torch.zeros(B, k); dg_mix = torch.zeros(B, k); dg_mix16 = torch.zeros(B, k); ynorm = torch.zeros(B, k)
        for b_ in range(B):
            ys = [O.expert_forward([f[b_:b_ + 1] for f in fe], pr, int(idx[b_, j]))[0] for j in range(k)]
            for j in range(k):
                dg_ref[b_, j] = (Dref[b_] * ys[j]).sum(); dg_mix[b_, j] = (Deng[b_] * ys[j]).sum()
                dg_mix16[b_, j] = (Deng[b_] * ys[j].to(torch.bfloat16).float()).sum(); ynorm[b_, j] = ys[j].norm()
            if b_ == 0:
                print("expert outputs of sample 0: |y0| %.3f |y1| %.3f |y0 - y1| %.3f ; cos(d out, y0) %.2e" % (
                    float(ys[0].norm()), float(ys[1].norm()), float((ys[0] - ys[1]).norm()),
                    float((Dref[0] * ys[0]).sum() / (Dref[0].norm() * ys[0].norm()))))
        dg_eng = ws["dgate"].cpu().view(B, k)
        print("dgate: engine vs oracle", rel(dg_eng, dg_ref), "| engine's d out with fp32 y", rel(dg_mix, dg_ref),
              "| engine's d out with bf16-rounded y", rel(dg_mix16, dg_ref), "| engine vs (engine's d out, bf16 y)", rel(dg_eng, dg_mix16))
    # ---- router backward in isolation: autograd (fp32, CPU) on the engine's own router input, gate gradients and labels ----
    rin = ws["router_in"].cpu().clone().requires_grad_(True)
    pw = {kk: pr[kk].detach() for kk in ("moe.router.0.weight", "moe.router.0.bias", "moe.router.2.weight", "moe.router.2.bias")}
    probs_r = O.router_probs(rin, pw)
    gates_r = O.gates_from_probs(probs_r, idx)
    obj = (gates_r * dg_eng).sum() + ocfg.w_cls * O.router_ce(probs_r, batch["label"])
    obj.backward()
    print("router backward in isolation: d router_in", rel(ws["drouter_in"], rin.grad), "| oracle chain", rel(ws["drouter_in"], router_in.grad),
          "| autograd-on-engine-inputs vs oracle chain", rel(rin.grad, router_in.grad))
    print("probs engine vs oracle", rel(ws["probs"], probs), " router_in engine vs oracle", rel(ws["router_in"], router_in))
