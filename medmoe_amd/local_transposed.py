"""GLoRIA local loss (reference losses.py:961-1026, attention_fn :698-736) on TRANSPOSED ragged pair matrices - the fast path for 196 / 64
regions (csrc/pair3.hip): score GEMM with the word softmax fused -> forward pair launch (sim, A, per-word sums) -> [a head over the
similarity matrix chosen by the caller: cross-entropy, Soft-GLoRIA, ...] -> backward pair launch (dS over the log-probabilities in place,
one row weight d2 per word) -> two wgrad-shaped GEMMs.  One implementation for the fused `Engine.train_step` (which hands its own workspace
in) and for `src.losses.GLORIALocalContrastiveLoss` behind torch autograd (`TransposedLocalLoss.standalone`)."""
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import ops

BF, F32, I32 = torch.bfloat16, torch.float32, torch.int32


def ragged_layout(cap_lens, T: int, Tp: int):
    """Column layout of the local-loss pair matrices (losses.py:961-1026 computes them per pair; the build stores them
    as [B*HWp, Kp] matrices).  Caption i (clamped to 1..T words) belongs to length class ntt_i = ceil(len_i / 16) and is
    16*ntt_i columns wide; classes are stored one after the other, members in original order.
    Returns perm (captions in column order), col_of_cap[i] (first column), ntts[i], cap_of_chunk (caption of every 8-column
    chunk, -1 for the zero padding up to Kp), classes = [(ntt, first index into perm, count, first column)], Kc, Kp."""
    lens = np.clip(np.asarray(cap_lens, dtype=np.int64), 1, T)
    B = lens.shape[0]
    ntts = (lens + 15) // 16
    perm = np.argsort(ntts, kind="stable")                       # class-major, original order inside a class
    width = 16 * ntts[perm]
    start = np.concatenate(([0], np.cumsum(width)))              # first column of each caption, in perm order
    col_of_cap = np.empty(B, np.int64); col_of_cap[perm] = start[:-1]
    Kc = int(start[-1]); Kp = (Kc + 63) // 64 * 64
    cap_of_chunk = np.full(Kp // 8, -1, np.int64)
    cap_of_chunk[:Kc // 8] = np.repeat(perm, width // 8)
    classes, pos = [], 0
    for ntt in range(1, Tp // 16 + 1):
        n_c = int((ntts == ntt).sum())
        if n_c:
            classes.append((ntt, pos, n_c, int(start[pos])))
            pos += n_c
    return perm, col_of_cap, ntts, cap_of_chunk, classes, Kc, Kp


class TransposedLocalLoss:
    """forward(ctx, words, cap_lens, cap_lens_host, temp1, temp2) fills ws["sim"] ([B, B] fp32, BEFORE temp3); the caller turns it into
    ws["gsim"] = d loss / d sim; backward(d_img_l) writes the bf16 gradient of the region features.
    Workspace entries (the engine's names): wn, sim, gsim, l_lse, gm3, gm3_crowmap, img_tiles, img_tile_count, dGm32, dGmq, dC32q, rowoff_q,
    ctx_xmap_q and - through `ensure_pair(Kp)` - the ragged l_dS, l_A, (l_U,) wT, words_r, l_stats3, (l_d2)."""

    def __init__(self, B: int, P: int, T: int, Do: int, HWp: int, Tp: int, HWq: int, device, ws: Dict[str, torch.Tensor],
                 ensure_pair: Callable[[int], None], gram: bool = True, Bc: Optional[int] = None, word_grad: bool = False):
        self.B, self.P, self.T, self.Do, self.HWp, self.Tp, self.HWq = B, P, T, Do, HWp, Tp, HWq
        self.Bc = B if Bc is None else Bc                        # captions: B images against Bc captions (Bc > B: the gathered captions of all ranks)
        self.device, self.ws, self.ensure_pair, self.gram = device, ws, ensure_pair, gram
        # word_grad: backward also returns d loss / d words (losses.py:985-1012 differentiates the word embeddings too; needed when the text
        # tower trains).  The pair matrices then use the ROW-MAJOR layout [word rows][image x region columns] (pitch ldk = B * HWq rounded up
        # to the GEMM k-step): the gradient through the scores, dS . ctx, is one NT GEMM over it.
        self.word_grad = word_grad
        self.ldk = (B * HWq + 63) // 64 * 64
        self._st = None
        self.generation = 0                                       # forward calls so far: a backward must belong to the latest one

    # --------------------------------------------------------------------------------------------------------------------------
    @classmethod
    def standalone(cls, B: int, P: int, T: int, Do: int, device, gram: bool = True, Bc: Optional[int] = None,
                   word_grad: bool = False) -> "TransposedLocalLoss":
        """Own workspace (what Engine._alloc provides for the fused step), pair matrices sized on first use."""
        Bc = B if Bc is None else Bc
        HWp, Tp, _ = ops.local_geometry(P, T)
        dev = torch.device(device)
        ws: Dict[str, torch.Tensor] = {}
        GR = (P + 31) // 32 * 32
        Q = HWp
        ws["wn"] = torch.empty(Bc, T, device=dev, dtype=F32)
        ws["sim"] = torch.empty(B, Bc, device=dev, dtype=F32); ws["gsim"] = torch.empty(B, Bc, device=dev, dtype=F32)
        ws["l_lse"] = torch.empty(B * HWp, Bc, device=dev, dtype=F32)
        ws["gm3"] = torch.zeros(B * GR, GR, device=dev, dtype=BF)
        arg = torch.arange(B * P, device=dev)
        ws["gm3_crowmap"] = (arg // P * GR + arg % P).to(I32)
        ws["dGm32"] = torch.empty(B, Q, Q, device=dev, dtype=F32); ws["dGmq"] = torch.empty(B * Q, Q, device=dev, dtype=BF)
        ws["dC32q"] = torch.zeros(B * Q, Do, device=dev, dtype=F32)
        ws["rowoff_q"] = (torch.arange(B + 1, device=dev) * Q).to(I32)
        arq = torch.arange(B * Q, device=dev)
        ws["ctx_xmap_q"] = (arq // Q * P + torch.clamp(arq % Q, max=P - 1)).to(I32)
        tl = [[b, m, (b + 1) * P, 0] for b in range(B) for m in range(b * P, (b + 1) * P, 128)]
        ws["img_tiles"] = torch.tensor(tl, device=dev, dtype=I32); ws["img_tile_count"] = torch.tensor([len(tl)], device=dev, dtype=I32)
        state = {"cap": 0}

        def ensure_pair(Kp: int):
            if Kp <= state["cap"]:
                return
            cap = min((Bc * Tp + 63) // 64 * 64, (int(Kp * 1.1) + 63) // 64 * 64)
            for name in ("l_A", "l_dS", "l_U", "wT", "words_r", "l_stats3", "l_d2", "l_dwn"):
                ws.pop(name, None)
            ldk = (B * Q + 63) // 64 * 64
            for name in (("l_A", "l_dS") if gram else ("l_A", "l_dS", "l_U")):
                # row-major matrices are zero-filled once: their pad columns (beyond B * Q) are operands of the word-gradient GEMM
                ws[name] = torch.zeros((cap, ldk), device=dev, dtype=BF) if word_grad else torch.empty((B * Q, cap), device=dev, dtype=BF)
            ws["wT"] = torch.empty((Do, cap), device=dev, dtype=BF)
            ws["words_r"] = torch.empty((cap, Do), device=dev, dtype=BF)
            ws["l_stats3"] = torch.empty((B, cap, 2), device=dev, dtype=F32)
            if gram:
                ws["l_d2"] = torch.empty((B, cap), device=dev, dtype=F32)
            if word_grad:
                ws["l_dwn"] = torch.empty((B, cap), device=dev, dtype=F32)
            state["cap"] = cap
        return cls(B, P, T, Do, HWp, Tp, Q, dev, ws, ensure_pair, gram, Bc, word_grad)

    # --------------------------------------------------------------------------------------------------------------------------
    def forward(self, ctx: torch.Tensor, words: torch.Tensor, cap_lens: torch.Tensor, cap_lens_host, temp1: float, temp2: float,
                att: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ctx bf16 [B*P, Do] region features, words bf16 [Bc, T, Do], cap_lens int32 [Bc] on the device + the same lengths on the host
        (the class tables are built there), att: optional fp32 [B, T, P] for the attention maps of the matching pairs (B == Bc)."""
        ws, B, Bc, P, T, Do, Tp, HWq = self.ws, self.B, self.Bc, self.P, self.T, self.Do, self.Tp, self.HWq
        perm, col_of_cap, ntts, cap_of_chunk, classes, Kc, Kp = ragged_layout(cap_lens_host, T, Tp)
        # row r of the matrices = word t of caption cap_of_chunk[r // 8]: its row in `words` (rows of padding words point at a
        # real row: their dS is exactly zero)
        rows = np.arange(Kp, dtype=np.int64)
        cap_of_row = np.repeat(cap_of_chunk, 8)
        t_of_row = rows - col_of_cap[np.maximum(cap_of_row, 0)]
        word_row = np.where(cap_of_row >= 0, cap_of_row * T + np.minimum(t_of_row, T - 1), 0)
        meta = torch.from_numpy(np.concatenate((perm, col_of_cap, 16 * ntts, word_row)).astype(np.int32)).to(self.device, non_blocking=True)
        d_perm, d_col, d_tp, d_wrow = meta[:Bc], meta[Bc:2 * Bc], meta[2 * Bc:3 * Bc], meta[3 * Bc:]
        self.ensure_pair(Kp)
        # image-major: element (row, image, region) at image*Kp*HWq + row*HWq + region - one (image, caption, word tile) unit of the pair
        # kernel is 16 x 448 contiguous bytes, and an image's block is a plain [Kp][HWq] matrix for the two wgrad-shaped GEMMs
        # (measured against [row][image][region] at batch 1024: pair launches 44.5 -> 36.3 ms)
        if self.word_grad:                                       # row-major: element (row, image, region) at row*ldk + image*HWq + region
            ld, bs = self.ldk, HWq
            tr = lambda name: ws[name].view(-1)[:Kp * ld].view(Kp, ld)
        else:
            ld, bs = HWq, Kp * HWq
            tr = lambda name: ws[name].view(-1)[:B * bs].view(B, Kp, HWq)
        X, AT = tr("l_dS"), tr("l_A")                           # X: log2-probabilities, then dS in place
        UT = None if self.gram else tr("l_U")
        Wr = ws["words_r"][:Kp]
        stats, srows = ws["l_stats3"], ws["l_stats3"].shape[1]      # (num, n2) of every (image, caption word): forward -> backward launch
        if Kp > Kc:
            for t_ in ((X, AT) if self.gram else (X, AT, UT)):
                if self.word_grad:
                    t_[Kc:].zero_()
                else:
                    t_[:, Kc:].zero_()
        ops.call("words_prep_ragged", words, ws["wn"], None, Bc, T, Tp, Do, d_col, d_tp, Kp)               # word norms only (no transposed copy: wT = null)
        torch.index_select(words.view(Bc * T, Do), 0, d_wrow, out=Wr)
        ops.gemm_nt(ctx, ctx, ws["gm3"], c_rowmap=ws["gm3_crowmap"], tiles=ws["img_tiles"], tile_count=ws["img_tile_count"],
                    max_tiles=ws["img_tiles"].shape[0], stride_b=P * Do, M=B * P, N=P)
        for ntt, start, n_c, cbase in classes:
            members = d_perm[start:start + n_c]
            ops.call("local_scores_t", ctx, words, cap_lens, X, ws["l_lse"], B, Bc, P, T, Do, members, n_c, ntt, cbase, ld, bs)
            ops.call("local_pair3", X, None, AT, None, ws["l_lse"], ws["gm3"], ws["wn"], cap_lens, None, ws["sim"], att,
                     stats, srows, B, Bc, P, T, temp1, temp2, 1e-8, members, n_c, ntt, cbase, ld, bs, HWq, None)
        self._st = (ctx, cap_lens, classes, d_perm, Kp, X, AT, UT, Wr, stats, srows, ld, bs, temp1, temp2)
        if self.word_grad:
            # matrix row of every (caption, word): the rows of the words gradient are gathered back through it (-1: padding word)
            lens = np.clip(np.asarray(cap_lens_host, dtype=np.int64), 1, T)
            tpos = np.arange(T, dtype=np.int64)[None, :]
            row_of_word = np.where(tpos < lens[:, None], col_of_cap[:, None] + tpos, -1)
            self._wg = (words, torch.from_numpy(row_of_word.reshape(-1)).to(self.device, non_blocking=True))
        self.generation += 1
        return ws["sim"]

    def backward(self, gsim: torch.Tensor, d_img_l: torch.Tensor, generation: Optional[int] = None) -> Optional[torch.Tensor]:
        """gsim fp32 [B, Bc] = d loss / d sim; d_img_l bf16 [B, P, Do] receives d loss / d region features.  The pair matrices of the
        forward pass are consumed in place: `generation` (the value of self.generation right after that forward) makes a backward that
        arrives after ANOTHER forward fail loudly instead of differentiating the wrong batch.
        word_grad instances return d loss / d words, fp32 [Bc, T, Do] (None otherwise)."""
        if generation is not None and generation != self.generation:
            raise RuntimeError("TransposedLocalLoss: backward of an earlier forward - the instance keeps ONE forward's pair matrices "
                               "(call backward before the next forward of the same geometry)")
        ws, B, Bc, P, T, Do, HWq = self.ws, self.B, self.Bc, self.P, self.T, self.Do, self.HWq
        ctx, cap_lens, classes, d_perm, Kp, X, AT, UT, Wr, stats, srows, ld, bs, temp1, temp2 = self._st
        d2 = None
        if self.gram:
            # dGm_b = sum over the words of d2 a a^T: the backward launch stores the row weight d2 (4 bytes per word) instead of the
            # matrix U = d2 * A, and the Gram GEMM scales its first operand's fragments (medmoe_gemm_tn_gram); rows no launch covers
            # must hold finite weights
            d2 = ws["l_d2"]
            d2.zero_()
        dwn = None
        if self.word_grad:
            dwn = ws["l_dwn"]
            dwn.zero_()
        for ntt, start, n_c, cbase in classes:
            members = d_perm[start:start + n_c]
            if dwn is not None:
                ops.call("local_pair3_wgrad", X, X, AT, UT, ws["l_lse"], ws["gm3"], ws["wn"], cap_lens, gsim, ws["sim"], None,
                         stats, srows, B, Bc, P, T, temp1, temp2, 1e-8, members, n_c, ntt, cbase, ld, bs, HWq, d2, dwn)
            else:
                ops.call("local_pair3", X, X, AT, UT, ws["l_lse"], ws["gm3"], ws["wn"], cap_lens, gsim, ws["sim"], None,
                         stats, srows, B, Bc, P, T, temp1, temp2, 1e-8, members, n_c, ntt, cbase, ld, bs, HWq, d2)
        dC = ws["dC32q"]
        dC.zero_(); ws["dGm32"].zero_()
        # dC = dS^T . W with the B image blocks seen as ONE [Kp][B*HWq] operand (chunks of HWq columns, bs apart): full 256-column tiles
        ops.call("gemm_tn_cols", X, ld, Wr, Do, dC, Do, Kp, B * HWq, Do, 1, 0, 0, 0, HWq, bs)
        if self.gram:
            ops.call("gemm_tn_gram", AT, ld, d2, srows, 1, ws["dGm32"], HWq, Kp, HWq, B, bs, HWq * HWq)               # dGm_b = A_b^T diag(d2_b) A_b
        else:
            ops.call("gemm_tn_cols", UT, ld, AT, ld, ws["dGm32"], HWq, Kp, HWq, HWq, B, bs, bs, HWq * HWq, 0, 0)        # dGm_b = U_b^T A_b
        ws["dGmq"].copy_(ws["dGm32"].view(B * HWq, HWq))
        ops.gemm_tn(ws["dGmq"], ctx, dC.view(B, HWq, Do), x_rowmap=ws["ctx_xmap_q"], row_off=ws["rowoff_q"], n_groups=B,
                    stride_w=HWq * Do, nsplit=1, M=B * HWq)                                  # dC_b += dGm_b . ctx_b
        ops.call("unpad_cast", dC, d_img_l, B, P, HWq, Do)
        if not self.word_grad:
            return None
        # ---- d loss / d words (losses.py:985-1012): through the scores S = ctx . w  ->  dS . ctx, one NT GEMM over the row-major dS with
        # the contraction over (image, region); through the word's own norm in the cosine -> (sum over images of dwn) * w ----
        words, row_of_word = self._wg
        ctxT = torch.zeros(Do, self.ldk, device=self.device, dtype=BF)
        ctxT[:, :B * HWq] = ctx.index_select(0, ws["ctx_xmap_q"].long()).t()        # pad regions repeat a row: their dS columns are exact zeros
        dW = torch.empty(Kp, Do, device=self.device, dtype=F32)
        ops.gemm_nt(X, ctxT, dW)
        cw = dwn[:, :Kp].sum(dim=0)
        sel = row_of_word.clamp(min=0)
        g = dW.index_select(0, sel) + cw.index_select(0, sel)[:, None] * words.reshape(Bc * T, Do).float()
        return (g * (row_of_word >= 0)[:, None]).view(Bc, T, Do)
