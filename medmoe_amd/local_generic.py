"""GLoRIA local loss (reference losses.py:961-1026) for ANY number of regions (the reference's Swin stage 0 has 3136; the LDS-tiled pair
kernels hold 64 / 196 / 256 / 576): the reference's own formulation - weighted context = bmm(ctx, attn) (losses.py:732), cosine against the word
(:690-695, :1002) - as grouped GEMMs over uniform pair matrices [B*HWp, B*Tp] plus four elementwise kernels (loss.hip "GENERIC-GEOMETRY").
The same launch sequence as `Engine._local_loss_generic`, with its own buffers, so that `src.losses.GLORIALocalContrastiveLoss` can run it
behind torch autograd for the Swin tower's 56 x 56 local features."""
from typing import Optional

import torch

from . import ops

BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class GenericLocalLoss:
    def __init__(self, B: int, HW: int, T: int, D: int, device):
        if T > 80 or D % 64:
            raise ValueError("GenericLocalLoss: at most 80 words per caption, embedding width a multiple of 64")
        dev = torch.device(device)
        self.B, self.HW, self.T, self.D, self.dev = B, HW, T, D, dev
        HWp, Tp = (HW + 15) // 16 * 16, (T + 15) // 16 * 16
        Kp = (B * Tp + 63) // 64 * 64
        self.HWp, self.Tp, self.Kp = HWp, Tp, Kp
        z = lambda *s, dt=BF: torch.zeros(*s, device=dev, dtype=dt)
        self.lp, self.A, self.dS, self.wT = z(B * HWp, Kp), z(B * HWp, Kp), z(B * HWp, Kp), z(D, Kp)
        self.WC, self.DWC, self.DWCt = z(B, Kp, D, dt=F32), z(B, Kp, D), z(B, D, Kp)
        self.stats, self.sume, self.lse = z(B, Kp, 4, dt=F32), z(B, B, dt=F32), z(B * HWp, B, dt=F32)
        self.wn, self.sim = z(B, T, dt=F32), z(B, B, dt=F32)
        self.members = torch.arange(B, device=dev, dtype=I32)
        self.col = (torch.arange(B, device=dev) * Tp).to(I32); self.tp = torch.full((B,), Tp, device=dev, dtype=I32)
        self.trtab = torch.tensor([[b * Kp * D, b * Kp * D, Kp, D] for b in range(B)], device=dev, dtype=torch.int64)
        tl = [[b, m, (b + 1) * HWp, 0] for b in range(B) for m in range(b * HWp, (b + 1) * HWp, 128)]
        self.tiles = torch.tensor(tl, device=dev, dtype=I32); self.tile_count = torch.tensor([len(tl)], device=dev, dtype=I32)
        self.row_off = (torch.arange(B + 1, device=dev) * HWp).to(I32)
        # no row padding (HW a multiple of 16: 3136, 576, 256) and K-steps of the 256-row grouped GEMM available: the two per-image GEMMs of the
        # backward run on the 256x256 kernel, and the context gradient leaves as bf16 through a residual epilogue (no fp32 partials, no unpad pass)
        self.dense = HWp == HW and D >= 128 and D % 64 == 0 and Kp >= 128
        if self.dense:
            tl = [[b, m, (b + 1) * HWp, 0] for b in range(B) for m in range(b * HWp, (b + 1) * HWp, 256)]
            self.tiles256 = torch.tensor(tl, device=dev, dtype=I32); self.tile256_count = torch.tensor([len(tl)], device=dev, dtype=I32)
            self.X1 = torch.empty(B * HWp, D, device=dev, dtype=BF)
        else:                                                         # fp32 partial sums + unpad pass
            self.dC32, self.dC32b = z(B * HWp, D, dt=F32), z(B * HWp, D, dt=F32)
        arp = torch.arange(B * HWp, device=dev)
        self.xmap = (arp // HWp * HW + torch.clamp(arp % HWp, max=HW - 1)).to(I32)
        self.generation = 0                                       # forward calls so far: a backward must belong to the latest one

    def forward(self, ctx16: torch.Tensor, words16: torch.Tensor, cap_lens: torch.Tensor, temp1: float, temp2: float) -> torch.Tensor:
        """ctx16 bf16 [B*HW, D] (region-major), words16 bf16 [B, T, D], cap_lens int32 [B] -> sim fp32 [B images, B captions] (before temp3)."""
        B, HW, T, D, HWp, Tp, Kp = self.B, self.HW, self.T, self.D, self.HWp, self.Tp, self.Kp
        self.ctx, self.words, self.cap, self.t1, self.t2 = ctx16, words16, cap_lens, temp1, temp2
        self.generation += 1
        if Kp > B * Tp:                                           # the instance is reused: the k-step padding columns stay exact zeros
            for t_ in (self.lp, self.A, self.dS, self.wT):
                t_[:, B * Tp:].zero_()
            self.DWC[:, B * Tp:].zero_()
        ops.call("words_prep_ragged", words16, self.wn, self.wT, B, T, Tp, D, self.col, self.tp, Kp)
        ops.call("local_scores_ragged", ctx16, words16, cap_lens, self.lp, self.lse, B, B, HW, T, D, self.members, B, Tp // 16, 0, Kp)
        ops.call("local_gen_fwd_a", self.lp, cap_lens, self.A, B, B, HW, HWp, T, Tp, temp1, Kp)
        self.WC.zero_()
        ops.gemm_tn(self.A, ctx16, self.WC, x_rowmap=self.xmap, row_off=self.row_off, n_groups=B, stride_w=Kp * D, nsplit=1, M=B * HWp)
        ops.call("local_gen_cos", self.WC, words16, self.wn, cap_lens, self.sim, self.stats, self.sume, B, B, T, Tp, D, temp2, 1e-8, Kp)
        return self.sim

    def attention_maps(self) -> torch.Tensor:
        """[B images, B captions, Tp words, HW regions] region attention of every pair (bf16); the reference keeps the matching pairs' maps."""
        B, HW, HWp, Tp = self.B, self.HW, self.HWp, self.Tp
        return self.A.view(B, HWp, -1)[:, :HW, :B * Tp].reshape(B, HW, B, Tp).permute(0, 2, 3, 1)

    def matching_attention_maps(self) -> torch.Tensor:
        """fp32 [B, T, HW]: the region attention of the MATCHING pairs (image i, caption i) - what the reference returns (losses.py:993-995) -
        read with one strided view instead of B slices: element (i, t, hw) sits at row i*HWp + hw, column i*Tp + t of A."""
        B, HW, HWp, Tp, T, Kp = self.B, self.HW, self.HWp, self.Tp, self.T, self.Kp
        v = torch.as_strided(self.A, (B, T, HW), (HWp * Kp + Tp, 1, Kp))
        return v.float()

    def backward(self, gsim: torch.Tensor, generation: Optional[int] = None) -> torch.Tensor:
        """gsim fp32 [B, B] = dL/dsim -> d ctx bf16 [B*HW, D] (the text tower is frozen: no word gradient).  `generation`: the value of
        self.generation right after the forward this backward belongs to - the pair matrices are consumed in place, a backward that arrives
        after ANOTHER forward is refused."""
        if generation is not None and generation != self.generation:
            raise RuntimeError("GenericLocalLoss: backward of an earlier forward - the instance keeps ONE forward's pair matrices")
        B, HW, T, D, HWp, Tp, Kp = self.B, self.HW, self.T, self.D, self.HWp, self.Tp, self.Kp
        ops.call("local_gen_dwctx", self.WC, self.words, self.wn, self.cap, gsim.contiguous(), self.stats, self.sume, self.DWC, B, B, T, Tp, D,
                 self.t2, 1e-8, Kp)
        ops.call("transpose_many", self.DWC, self.DWCt, self.trtab, B, ((Kp + 63) // 64) * ((D + 63) // 64))
        if self.dense:
            grp = dict(tiles=self.tiles256, tile_count=self.tile256_count, max_tiles=self.tiles256.shape[0], M=B * HWp, tile_rows=256)
            ops.gemm_nt(self.ctx, self.DWC, self.dS, stride_b=Kp * D, N=Kp, **grp)                             # dA_b = ctx_b dwctx_b^T
            ops.gemm_nt(self.A, self.DWCt, self.X1, stride_b=D * Kp, N=D, **grp)                               # d ctx_b (direct) = A_b dwctx_b
            ops.call("local_gen_bwd_s", self.lp, self.A, self.dS, self.cap, B, B, HW, HWp, T, Tp, self.t1, Kp) # dS over dA in place
            dctx = torch.empty(B * HW, D, device=self.dev, dtype=BF)
            ops.gemm_nt(self.dS, self.wT, dctx, residual=self.X1)                                              # d ctx = dS . W + the direct part
            return dctx
        grp = dict(tiles=self.tiles, tile_count=self.tile_count, max_tiles=self.tiles.shape[0], M=B * HWp)
        ops.gemm_nt(self.ctx, self.DWC, self.dS, a_rowmap=self.xmap, stride_b=Kp * D, N=Kp, **grp)            # dA_b = ctx_b dwctx_b^T
        ops.gemm_nt(self.A, self.DWCt, self.dC32b, stride_b=D * Kp, N=D, **grp)                                # d ctx_b (direct) = A_b dwctx_b
        ops.call("local_gen_bwd_s", self.lp, self.A, self.dS, self.cap, B, B, HW, HWp, T, Tp, self.t1, Kp)     # dS over dA in place
        ops.gemm_nt(self.dS, self.wT, self.dC32)                                                               # d ctx += dS . W
        dctx = torch.empty(B * HW, D, device=self.dev, dtype=BF)
        ops.call("unpad_cast2", self.dC32, self.dC32b, dctx, B, HW, HWp, D)
        return dctx
