"""TEST INFRASTRUCTURE ONLY.  tests/golden/soft_gloria.npz: the REFERENCE's SoftGLORIAGlobalContrastiveLoss / SoftGLORIALocalContrastiveLoss
(src/losses.py:814-883, 1111-1214; imported through oracle/_ref_import.py) on seeded fp32 CPU inputs, with the gradients of their inputs;
tests/golden/hard_negative.npz: its HardNegativeContrastiveLoss (:885-927) likewise.
Run in the build container only:  python oracle/gen_golden_soft.py   (data only - no reference source is written)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    R = _ref_import.load()
    torch.manual_seed(4321)
    B, D, T, H = 8, 64, 16, 4
    f = torch.nn.functional.normalize(torch.randn(B, 6) + torch.tensor([1.5, 0, 0, 0, 0, 0.0]), dim=1)
    soft = f @ f.t()                                          # caption-to-caption scores, diagonal 1
    thr = (0.55, 0.35)
    npos = (soft > thr[0]).sum(1)
    nneg = (soft <= thr[1]).sum(1)
    assert int(npos.max()) >= 3 and int(npos.min()) >= 1 and int(nneg.max()) >= 2, (npos, nneg)
    a = torch.randn(B, D, requires_grad=True)
    t = torch.randn(B, D, requires_grad=True)
    g = R.losses.SoftGLORIAGlobalContrastiveLoss()(a, t, temp3=10.0, idx=soft, probs=thr)
    g.backward()
    il = torch.randn(B, D, H, H, requires_grad=True)
    wl = torch.randn(B, D, T, requires_grad=True)
    cl = [16, 5, 9, 12, 3, 16, 7, 10]
    o = R.losses.SoftGLORIALocalContrastiveLoss()(il, wl, cl, temp1=4.0, temp2=5.0, temp3=10.0, idx=soft, probs=thr)
    (o.loss0 + 2.0 * o.loss1).backward()
    d = {"soft": soft, "thresholds": np.asarray(thr, np.float32), "a": a, "t": t, "g_loss": g, "grad_a": a.grad, "grad_t": t.grad,
         "img_l": il, "words": wl, "cap_lens": np.asarray(cl, np.int64), "loss0": o.loss0, "loss1": o.loss1,
         "grad_img_l": il.grad, "grad_words": wl.grad}
    for i, m in enumerate(o.att_maps):
        d[f"att{i}"] = m
    np.savez(os.path.join(OUT, "soft_gloria.npz"),
             **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()})
    # HardNegativeContrastiveLoss (losses.py:885-927): rows with an active and an inactive margin
    torch.manual_seed(99)
    hi = torch.randn(B, D, requires_grad=True)
    ht = (hi.detach() * 0.35 + torch.randn(B, D)).requires_grad_(True)
    hl = R.losses.HardNegativeContrastiveLoss()(hi, ht)
    hl.backward()
    hl2 = R.losses.HardNegativeContrastiveLoss(margin=0.9)(hi.detach(), ht.detach())
    np.savez(os.path.join(OUT, "hard_negative.npz"), imgs=hi.detach().numpy(), caps=ht.detach().numpy(), loss=hl.detach().numpy(),
             grad_imgs=hi.grad.numpy(), grad_caps=ht.grad.numpy(), loss_margin09=hl2.numpy())
    print("hard_negative.npz: loss", float(hl), "margin 0.9:", float(hl2))
    print("soft_gloria.npz: g", float(g), "l0", float(o.loss0), "l1", float(o.loss1), "pos", npos.tolist(), "neg", nneg.tolist())


if __name__ == "__main__":
    main()
