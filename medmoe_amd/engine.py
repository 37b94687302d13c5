"""Hand-scheduled forward / backward / optimiser step of the MedMoE contrastive hot path.

One process drives one MI355X.  Every arithmetic step is a C-ABI HIP kernel launch
(`medmoe_amd.ops`) on torch's current stream into buffers allocated once per batch size;
torch is used for device memory, streams and (multi-GPU) `torch.distributed` only.

Reference call stack being replaced: MedMoE.forward (med_moe.py:102-108) ->
encode_text / encode_image -> SWIN.forward (swin.py:130-149) -> MoE.forward (swin.py:94-117);
model_step (medmoe_module.py:284-316) -> GLORIA local/global losses (losses.py:757-794,
954-1026) + CE on router probabilities (medmoe_module.py:235-237).
"""
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .local_transposed import TransposedLocalLoss, ragged_layout  # noqa: F401  (ragged_layout: the [region][word] path and the tests use it from here)
from .config import MedMoEConfig
from .params import ParamStore

BF = torch.bfloat16
F32 = torch.float32
I32 = torch.int32


class VocabTables:
    """What word-piece aggregation needs from the tokenizer vocabulary (text_encoder.py:23,47-74)."""

    def __init__(self, is_continuation: torch.Tensor, starts_bracket: torch.Tensor, sep_id=2, cls_id=1, pad_id=0):
        self.is_cont = is_continuation.bool()
        self.starts_bracket = starts_bracket.bool()
        self.sep_id, self.cls_id, self.pad_id = sep_id, cls_id, pad_id

    @staticmethod
    def synthetic(vocab: int, device, n_continuation: int = 0):
        cont = torch.zeros(vocab, dtype=torch.bool, device=device)
        if n_continuation:
            cont[vocab - n_continuation:] = True
        br = torch.zeros(vocab, dtype=torch.bool, device=device)
        br[:3] = True
        return VocabTables(cont, br)

    def segment_map(self, ids: torch.Tensor, out=None):
        """(out: optional (seg, cap) int32 device buffers to write into - the engine keeps one pair per batch size.)  The token loop of text_encoder.py:45-76 on the device (no host sync): seg[b,t] = word index of token t (-1 = dropped), cap_lens
        per medmoe_module.py:221-223.  One kernel launch (medmoe_segment_map); `segment_map_torch` is the same rule in torch ops (the
        tests compare the two and the oracle)."""
        B, T = ids.shape
        if not ids.is_cuda:
            return self.segment_map_torch(ids)
        if ids.dtype not in (torch.int64, torch.int32) or not ids.is_contiguous():
            ids = ids.to(torch.int64).contiguous()
        if out is not None:
            seg, cap = out
        else:
            seg = torch.empty(B, T, device=ids.device, dtype=I32); cap = torch.empty(B, device=ids.device, dtype=I32)
        ops.call("segment_map", ids, 1 if ids.dtype == torch.int64 else 0, self.is_cont.view(torch.uint8), self.starts_bracket.view(torch.uint8),
                 seg, cap, B, T, self.is_cont.numel(), self.sep_id)
        return seg, cap

    def segment_map_torch(self, ids: torch.Tensor):
        B, T = ids.shape
        pos = torch.arange(T, device=ids.device)[None]
        is_sep = ids == self.sep_id
        has_sep = is_sep.any(dim=1, keepdim=True)
        sep_pos = torch.where(has_sep, is_sep.float().argmax(dim=1, keepdim=True), torch.full_like(ids[:, :1], T))
        valid = pos <= sep_pos
        start = (~self.is_cont[ids]) & valid
        start[:, 0] = True
        seg = torch.cumsum(start.int(), dim=1) - 1
        n_words = torch.where(has_sep[:, 0], seg.gather(1, sep_pos.clamp(max=T - 1))[:, 0] + 1, seg[:, -1])
        # without a [SEP] the loop never flushes the last bank: that word is dropped
        seg = torch.where(valid & (seg < n_words[:, None]), seg, torch.full_like(seg, -1))
        first_br = self.starts_bracket[ids] & start & (seg >= 0)
        n_real = (start & (seg >= 0)).sum(dim=1) - first_br.sum(dim=1)
        return seg.to(I32).contiguous(), (n_real + 1).to(I32).contiguous()


class Engine:
    def __init__(self, cfg: MedMoEConfig, device="cuda:0", seed: int = 0, vocab: Optional[VocabTables] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("medmoe_amd.Engine needs a GPU: the HIP path is the only path")
        cfg.validate()
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.params = ParamStore(cfg, self.device, seed)
        self.vocab = vocab or VocabTables.synthetic(cfg.vocab, self.device)
        self.B = 0
        self.ws: Dict[str, torch.Tensor] = {}
        self.rank, self.world = 0, 1
        self._seg = None; self._cap_host = None; self._cap_event = None; self.cap_lens = None
        self._tl = None                                              # TransposedLocalLoss over this engine's workspace
        self._tlg = None                                             # ... against the gathered captions (cfg.local_loss_global)
        self.dist = False        # take the data-parallel exchange steps (all-gather / reduce-scatter / bucketed all-reduce)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world = torch.distributed.get_rank(), torch.distributed.get_world_size()
            # MEDMOE_DIST_WORLD1=1: run the collectives even in a one-rank group, so that the RCCL path (backend "nccl",
            # async buckets, stream ordering) can be exercised on a single GPU (tests/test_rccl_world1_gpu.py)
            self.dist = self.world > 1 or os.environ.get("MEDMOE_DIST_WORLD1") == "1"
        # trainable text tower (cfg.freeze_text = False; reference freeze_bert: false): flat master / gradient / Adam buffers of its own, the padded
        # text pass with saved activations, a text backward, and the local loss in its word-gradient mode
        self.train_text = not cfg.freeze_text
        self.tstore = None
        if self.train_text:
            if cfg.soft_label:
                raise NotImplementedError("soft_label with a trainable text tower: the reference scores captions with a SEPARATE frozen BERT "
                                          "(medmoe_module.py:207-210); this build takes them from the tower itself, which must then stay frozen")
            from .text_params import TextStore
            self.tstore = TextStore(cfg, self.device, self.params.text)
            self.params.text = self.tstore.as_dict()             # views of the flat buffers: an optimiser step updates them in place
        self._tlw = None                                             # TransposedLocalLoss in word-gradient mode (own buffers)
        self.local_dense = False
        self.HWp, self.Tp, self.GW = ops.local_geometry(cfg.n_patch, cfg.max_len)
        # LDS-tiled pair kernels exist for 64 / 208 / 256 regions; any other geometry (576 regions of ViT-L/14 at 336 px) runs
        # the generic GEMM formulation (_local_loss_generic)
        self.text_varlen = os.environ.get("MEDMOE_TEXT_VARLEN", "1") != "0" and not self.train_text
        self.local_fast = ops.local_fast_path(cfg.n_patch, cfg.max_len)
        # transposed pair matrices + one wave per (image, caption, word tile): geometries pair3.hip is instantiated for (196 / 64
        # regions); MEDMOE_LOCAL_PAIR3=0 keeps the [region][word] kernels (local_pair2) for A/B runs
        self.local_t = (self.local_fast and ops.local_pair3_supported(cfg.n_patch, cfg.max_len) and cfg.d_out % 32 == 0 and cfg.d_out >= 128
                        and os.environ.get("MEDMOE_LOCAL_PAIR3", "1") != "0")
        # the per-image Gram gradient from ONE operand: the backward pair launch stores a row weight per word instead of the U matrix
        # (MEDMOE_LOCAL_GRAM=0: the two-operand form, for A/B runs)
        self.local_gram = self.local_t and os.environ.get("MEDMOE_LOCAL_GRAM", "1") != "0"
        # gradient buckets in flat-buffer order: [embeddings | layer 0 | ... | layer L-1 | final LN + router + experts]
        off = self.params.offsets
        self.bucket_bounds = [0] + [off[f"vit.layer.{l}.attention_layernorm.weight"] for l in range(cfg.n_layer_v)] \
            + [off["vit.final_layer_norm.weight"], self.params.numel]
        self._reducer = None
        self._side = None
        self.overlap_wgrad = os.environ.get("MEDMOE_OVERLAP_WGRAD", "1") == "1"    # weight-gradient GEMMs on a second stream (backward)
        # MEDMOE_WGRAD_STAGED=1: plain wgrads without atomics on dW (medmoe_gemm_tn_staged: partial tiles + a summing kernel, deterministic).
        # Measured no faster than the atomic form (batch 128: 25.20-25.34 against 25.13-25.19 ms; batch 1024: 217.4 against 216.4): default off
        self.wgrad_staged = os.environ.get("MEDMOE_WGRAD_STAGED", "0") == "1"
        # MEDMOE_GRAPH=1: the two fixed launch sequences of a step - [zero the gradient, both towers' forward, MoE forward] and [the whole
        # backward] - are captured into hipGraphs (torch.cuda.CUDAGraph around the C-ABI launches, the second stream forked and joined inside
        # the capture) and replayed; the losses in between stay eager (the local loss builds its class tables on the host) and so does the
        # optimiser (its bias-correction scalars are host values).  Single rank, frozen text tower.
        self.use_graph = os.environ.get("MEDMOE_GRAPH", "0") == "1"
        self._graph = None

    # ------------------------------------------------------------------------------------------
    # workspace
    # ------------------------------------------------------------------------------------------
    def _alloc(self, B: int):
        if B == self.B:
            return
        c, dev = self.cfg, self.device
        self.B = B
        ws = self.ws = {}
        Nt, Dv, P, L = c.n_tok_v, c.d_v, c.n_patch, c.n_layer_v
        M = B * Nt
        H = c.n_head_v

        def buf(name, shape, dtype=BF):
            ws[name] = torch.empty(shape, device=dev, dtype=dtype)
        ws["im2col"] = torch.zeros((B * P, c.patch_dim_pad), device=dev, dtype=BF)    # pad columns (patch 14: 588..639) stay zero
        for l in range(L + 1):
            buf(f"x{l}", (M, Dv))
        for l in range(L):
            buf(f"ln1_{l}", (M, Dv)); buf(f"st1_{l}", (2, M), F32)
            buf(f"qkv{l}", (M, 3 * Dv)); buf(f"att{l}", (M, Dv)); buf(f"lse{l}", (B * H * Nt,), F32)
            buf(f"xmid{l}", (M, Dv)); buf(f"ln2_{l}", (M, Dv)); buf(f"st2_{l}", (2, M), F32)
            buf(f"z{l}", (M, c.ff_v)); buf(f"h{l}", (M, c.ff_v))
        buf("lnf", (M, Dv)); buf("stf", (2, M), F32)
        # backward scratch
        buf("dxa", (M, Dv)); buf("dxb", (M, Dv)); buf("dln", (M, Dv)); buf("dqkv", (M, 3 * Dv)); buf("datt", (M, Dv))
        buf("dz", (M, c.ff_v)); buf("delta", (B * H * Nt,), F32)
        ws["rowmap_patch"] = (torch.arange(B * P, device=dev) // P * Nt + 1 + torch.arange(B * P, device=dev) % P).to(I32)
        # MoE
        E, k, Do, Dh = c.n_expert, c.top_k, c.d_out, c.d_out // 2
        R = B * k * P
        self.R = R
        self.max_tiles = (R + 127) // 128 + E
        buf("router_in", (B, Dv), F32); buf("router_h", (B, c.router_hidden), F32)
        buf("probs", (B, E), F32); buf("idx", (B, k), I32); buf("gates", (B, k), F32)
        buf("slot_of", (B * k,), I32); buf("item_of_slot", (B * k,), I32); buf("expert_of_slot", (B * k,), I32)
        buf("row_off", (E + 1,), I32); buf("tiles", (2 * self.max_tiles, 4), I32); buf("tile_count", (2,), I32)   # 128-row table, then 256-row table
        buf("rowmap", (R,), I32)
        buf("G", (4, R, Do)); buf("H1", (4, R, Dh)); buf("eout", (R, Do)); buf("wts", (R, 4), F32)
        buf("dG", (4, R, Do)); buf("dH1", (4, R, Dh)); buf("dF", (4, R, Dv))
        if c.expert_fp8:           # e4m3 activation rows + their scales (one set, reused by every fp8 GEMM of the step)
            buf("q8", (R, max(Dv, Do)), torch.uint8); buf("q8s", (R,), F32)
        buf("img_l", (B, P, Do)); buf("img_g", (B, Do), F32)
        buf("d_img_g", (B, Do), F32); buf("d_img_l", (B, P, Do)); buf("dgate", (B * k,), F32)
        buf("dlogits", (B, E), F32); buf("drouter_h", (B, c.router_hidden), F32); buf("drouter_in", (B, Dv), F32)
        buf("ones", (B,), F32); ws["ones"].fill_(1.0)
        buf("loss_parts", (8,), F32)    # [0]=cls loss [1]=cls acc [2]=global loss [3]=local loss
        # text tower
        T, Dt = c.max_len, c.d_t
        Mt = B * T
        buf("tx0", (Mt, Dt)); buf("tx1", (Mt, Dt)); buf("tx2", (Mt, Dt)); buf("tr", (Mt, Dt)); buf("tqkv", (Mt, 3 * Dt)); buf("tatt", (Mt, Dt))
        buf("th", (Mt, c.ff_t)); buf("tlse", (B * c.n_head_t * T,), F32); buf("tstat", (2, Mt), F32)
        ws["tpack"] = torch.zeros(2 * Mt + B + 2, device=dev, dtype=I32)           # text_pack: tok_row | src_of_row | seq_off | count
        for j in range(min(c.last_n_layers, c.n_layer_t + 1)):
            buf(f"ths{j}", (Mt, Dt))
        buf("words", (B, T, Dt)); buf("words32", (B, T, Dt), F32); buf("txt_g", (B, Dt), F32)
        buf("seg", (B, T), I32); buf("cap", (B,), I32)
        if self.train_text:        # every layer's activations stay for the text backward (padded pass: B x T rows)
            Lt, Ht = c.n_layer_t, c.n_head_t
            for l in range(Lt + 1):
                buf(f"t_x{l}", (Mt, Dt))
            for l in range(Lt):
                buf(f"t_qkv{l}", (Mt, 3 * Dt)); buf(f"t_att{l}", (Mt, Dt)); buf(f"t_lse{l}", (B * Ht * T,), F32)
                buf(f"t_x1{l}", (Mt, Dt)); buf(f"t_st1{l}", (2, Mt), F32); buf(f"t_r{l}", (Mt, Dt))
                buf(f"t_h{l}", (Mt, c.ff_t)); buf(f"t_dg{l}", (Mt, c.ff_t)); buf(f"t_x2{l}", (Mt, Dt)); buf(f"t_st2{l}", (2, Mt), F32)
            buf("t_dH", (Mt, Dt)); buf("t_da", (Mt, Dt)); buf("t_db", (Mt, Dt)); buf("t_dc", (Mt, Dt)); buf("t_datt", (Mt, Dt))
            buf("t_dz", (Mt, c.ff_t)); buf("t_dqkv", (Mt, 3 * Dt)); buf("t_delta", (B * Ht * T,), F32); buf("t_dxemb", (Mt, Dt), F32)
            buf("d_txt_g", (B, Dt), F32); buf("cb", (B * self.world,), F32)
            if self.dist:
                buf("d_txt_all", (B * self.world, Dt), F32); buf("cb1", (B * self.world,), F32)
        # global loss
        Bg = B * self.world
        buf("na", (B,), F32); buf("nb", (Bg,), F32); buf("S", (B, Bg), F32); buf("dS", (B, Bg), F32)
        buf("ca", (B,), F32)
        if self.dist:
            buf("nb2", (Bg,), F32); buf("na2", (B,), F32); buf("S2", (B, Bg), F32); buf("dS2", (B, Bg), F32)
            buf("cb2", (Bg,), F32); buf("ca2", (B,), F32); buf("d_img_all", (Bg, Do), F32)
        # local loss
        HWp, Tp, GW = self.HWp, self.Tp, self.GW
        Kmax = (B * Tp + 63) // 64 * 64      # widest ragged row: every caption in the longest class, rounded up to the GEMM k-step
        buf("wn", (B, T), F32)
        if not self.local_fast:
            buf("wT", (Dt, Kmax))
        buf("sim", (B, B), F32); buf("gsim", (B, B), F32)
        buf("l_lse", (B * HWp, B), F32); ws["dC32"] = torch.zeros(B * HWp, Do, device=dev, dtype=F32)
        if self.local_fast:
            # the three ragged pair matrices [B*HWp, Kp] are sized from the batch's own sum of pad16(caption length) (plus 10 %
            # head-room, grown when a later batch is longer) instead of the B*Tp worst case: 3 x 24 GB instead of 3 x 35 GB at
            # B = 1024 with lengths uniform in 8..77 (_pair_buffers)
            self._pair_cap = 0
            ws["gmp"] = torch.zeros(B * HWp, GW, device=dev, dtype=BF)
            buf("dGm", (B * HWp, HWp))
            if self.local_t:       # plain Gram matrices [B][GR][GR] (zero outside [P][P]: written once here, the GEMM only fills [P][P])
                GR = (P + 31) // 32 * 32
                ws["gm3"] = torch.zeros(B * GR, GR, device=dev, dtype=BF)
                arg = torch.arange(B * P, device=dev)
                ws["gm3_crowmap"] = (arg // P * GR + arg % P).to(I32)
                # region columns stored per (word, image) in the pair matrices: HWp (208 for 196 regions).  MEDMOE_PAIR_PITCH=224 makes
                # every 64-byte wave segment 64-byte aligned (448-byte rows); measured on one box at batch 1024: pair launches 36.3 ->
                # 37.1 ms and 7.7 % more GEMM work, step 234.6 -> 241.2 ms - not used
                Q = self.HWq = int(os.environ.get("MEDMOE_PAIR_PITCH", HWp))
                if Q % 8 or not P <= Q <= GR:
                    raise ValueError(f"MEDMOE_PAIR_PITCH={Q}: need a multiple of 8 in [{P}, {GR}]")
                buf("dGm32", (B, Q, Q), F32); buf("dGmq", (B * Q, Q)); ws["dC32q"] = torch.zeros(B * Q, Do, device=dev, dtype=F32)
                ws["rowoff_q"] = (torch.arange(B + 1, device=dev) * Q).to(I32)
                arq = torch.arange(B * Q, device=dev)
                ws["ctx_xmap_q"] = (arq // Q * P + torch.clamp(arq % Q, max=P - 1)).to(I32)
        else:       # generic path: word log-probabilities, weighted contexts and their gradients
            buf("l_dS", (B * HWp, Kmax)); buf("l_A", (B * HWp, Kmax))
            buf("l_LP", (B * HWp, Kmax)); buf("l_WC", (B, Kmax, Do), F32); buf("l_DWC", (B, Kmax, Do)); buf("l_DWCt", (B, Do, Kmax))
            buf("l_stats", (B, Kmax, 4), F32); buf("l_sume", (B, B), F32); buf("dC32b", (B * HWp, Do), F32)
            ws["l_members"] = torch.arange(B, device=dev, dtype=I32)
            ws["l_col"] = (torch.arange(B, device=dev) * Tp).to(I32); ws["l_tp"] = torch.full((B,), Tp, device=dev, dtype=I32)
            ws["l_trtab"] = torch.tensor([[b * Kmax * Do, b * Kmax * Do, Kmax, Do] for b in range(B)], device=dev, dtype=torch.int64)
            # region rows without padding (576 = 36 x 16): the backward's per-image GEMMs on 256-row tiles, the context gradient through a
            # residual epilogue (medmoe_amd/local_generic.py, the same form)
            self.local_dense = HWp == P and Do >= 128 and Do % 64 == 0 and Kmax >= 128
            if self.local_dense:
                tl = [[b, m, (b + 1) * HWp, 0] for b in range(B) for m in range(b * HWp, (b + 1) * HWp, 256)]
                ws["imgp_tiles256g"] = torch.tensor(tl, device=dev, dtype=I32); ws["imgp_tile256g_count"] = torch.tensor([len(tl)], device=dev, dtype=I32)
                buf("l_X1", (B * HWp, Do))
        # static per-image group tables
        tl = []
        for b in range(B):
            for m in range(b * P, (b + 1) * P, 128):
                tl.append([b, m, (b + 1) * P, 0])
        ws["img_tiles"] = torch.tensor(tl, device=dev, dtype=I32); ws["img_tile_count"] = torch.tensor([len(tl)], device=dev, dtype=I32)
        ar = torch.arange(B * P, device=dev)
        ws["gm_crowmap"] = (ar // P * HWp + ar % P).to(I32)
        tl = []
        for b in range(B):
            for m in range(b * HWp, (b + 1) * HWp, 128):
                tl.append([b, m, (b + 1) * HWp, 0])
        ws["imgp_tiles"] = torch.tensor(tl, device=dev, dtype=I32); ws["imgp_tile_count"] = torch.tensor([len(tl)], device=dev, dtype=I32)
        tl = [[b, b * HWp, (b + 1) * HWp, 0] for b in range(B)] if HWp <= 256 else []      # one 256-row tile per image
        ws["imgp_tiles256"] = torch.tensor(tl, device=dev, dtype=I32).reshape(-1, 4); ws["imgp_tile256_count"] = torch.tensor([len(tl)], device=dev, dtype=I32)
        ws["imgp_row_off"] = (torch.arange(B + 1, device=dev) * HWp).to(I32)
        arp = torch.arange(B * HWp, device=dev)
        ws["ctx_xmap"] = (arp // HWp * P + torch.clamp(arp % HWp, max=P - 1)).to(I32)

    # ------------------------------------------------------------------------------------------
    # image tower forward (ViT blocks = transformer.py:98-114 pre-norm; embeddings build-defined)
    # ------------------------------------------------------------------------------------------
    def _expert_tiles(self, K: int) -> dict:
        """Tile-table arguments of a grouped expert GEMM with reduction length K: the 256x256 kernel and the 256-row
        table medmoe_dispatch writes behind the 128-row one, or (K < 128) the 128x128 kernel."""
        ws, R, E = self.ws, self.R, self.cfg.n_expert
        if K >= 128:
            return dict(tiles=ws["tiles"][self.max_tiles:], tile_count=ws["tile_count"][1:], max_tiles=(R + 255) // 256 + E, M=R, tile_rows=256)
        return dict(tiles=ws["tiles"], tile_count=ws["tile_count"], max_tiles=self.max_tiles, M=R)

    def _fp8_gemm(self, x, K, rowmap, colscale, wq, wscale, bias, out, N, K_, epi, residual=None, aux=None):
        """out[r, :N] = epi(fp8 product of the rows of x (gathered through rowmap; times the per-expert column scale when the
        weight scales were folded into the rows, i.e. dgrad) with the expert's e4m3 weight [N, K]) over the dispatch's 128-row tiles."""
        ws, P = self.ws, self.cfg.n_patch
        q, qs = ws["q8"].view(-1)[:self.R * K].view(self.R, K), ws["q8s"]
        ops.call("quant_rows_e4m3", x, x.stride(-2), rowmap, colscale, ws["expert_of_slot"] if colscale is not None else None, P, q, qs, self.R, K)
        ops.call("gemm_fp8_grouped", q, qs, wq, wscale, bias, out, out.stride(-2), residual, aux, ws["tiles"], ws["tile_count"], self.max_tiles,
                 N, K_, N * K_, N if wscale is not None else 0, N if bias is not None else 0, epi)

    def forward_image(self, images: torch.Tensor):
        B = images.shape[0]
        self._alloc(B)
        c, p, ws = self.cfg, self.params, self.ws
        Nt, Dv, P, H = c.n_tok_v, c.d_v, c.n_patch, c.n_head_v
        M = B * Nt
        if images.dtype not in (F32, BF) or tuple(images.shape[1:]) != (3, c.img_size, c.img_size) or not images.is_contiguous():
            raise ValueError("images must be contiguous [B,3,H,W] fp32/bf16")
        ops.call("patchify_ld", images, ws["im2col"], B, 3, c.img_size, c.img_size, c.patch, 1 if images.dtype == F32 else 0,
                 c.patch_dim_pad)        # pad columns were zeroed at allocation and are never written
        x = ws["x0"]
        ops.call("init_tokens", x, p.f32("vit.cls_token"), p.f32("vit.pos_embed"), B, Nt, Dv)
        ops.gemm_nt(ws["im2col"], p.w16("vit.patch_embed.weight"), x, bias=p.f32("vit.patch_embed.bias"), residual=x,
                    c_rowmap=ws["rowmap_patch"], M=B * P)
        self._vit_blocks(B)
        ops.call("mean_tokens", ws["lnf"], ws["router_in"], B, Nt, Dv, 1, P)          # swin.py:137
        self._moe_forward(B)

    def _vit_blocks(self, B):
        """The pre-norm blocks (transformer.py:98-114) from ws["x0"] to ws["x{L}"] and the final LayerNorm (-> ws["lnf"]); every
        intermediate the backward needs stays in the workspace.  tests/test_ref_fixtures_gpu.py feeds the reference's own
        TransformerEncoder fixture in here."""
        c, p, ws = self.cfg, self.params, self.ws
        Nt, Dv, H = c.n_tok_v, c.d_v, c.n_head_v
        for l in range(c.n_layer_v):
            pre = f"vit.layer.{l}."
            x, xo = ws[f"x{l}"], ws[f"x{l + 1}"]
            st1, st2 = ws[f"st1_{l}"], ws[f"st2_{l}"]
            ops.layernorm_fwd(x, p.f32(pre + "attention_layernorm.weight"), p.f32(pre + "attention_layernorm.bias"),
                              ws[f"ln1_{l}"], st1[0], st1[1], c.eps_v)
            ops.gemm_nt(ws[f"ln1_{l}"], p.w16(pre + "attention.input_proj.weight"), ws[f"qkv{l}"],
                        bias=p.f32(pre + "attention.input_proj.bias"))
            ops.attn_fwd(ws[f"qkv{l}"], ws[f"att{l}"], ws[f"lse{l}"], None, B, Nt, H)
            ops.gemm_nt(ws[f"att{l}"], p.w16(pre + "attention.output_proj.weight"), ws[f"xmid{l}"],
                        bias=p.f32(pre + "attention.output_proj.bias"), residual=x)
            ops.layernorm_fwd(ws[f"xmid{l}"], p.f32(pre + "feedforward_layernorm.weight"),
                              p.f32(pre + "feedforward_layernorm.bias"), ws[f"ln2_{l}"], st2[0], st2[1], c.eps_v)
            ops.gemm_nt(ws[f"ln2_{l}"], p.w16(pre + "feedforward.model.0.weight"), ws[f"h{l}"],
                        bias=p.f32(pre + "feedforward.model.0.bias"), aux=ws[f"z{l}"], epi=ops.EPI_GELU_DAUX)    # aux <- GELU'(z): the backward epilogue is one multiply
            ops.gemm_nt(ws[f"h{l}"], p.w16(pre + "feedforward.model.2.weight"), xo,
                        bias=p.f32(pre + "feedforward.model.2.bias"), residual=ws[f"xmid{l}"])
        xl = ws[f"x{c.n_layer_v}"]
        ops.layernorm_fwd(xl, p.f32("vit.final_layer_norm.weight"), p.f32("vit.final_layer_norm.bias"), ws["lnf"],
                          ws["stf"][0], ws["stf"][1], c.eps_v)

    def _moe_forward(self, B):
        c, p, ws = self.cfg, self.params, self.ws
        E, k, Do, Dh, Dv, P, Nt = c.n_expert, c.top_k, c.d_out, c.d_out // 2, c.d_v, c.n_patch, c.n_tok_v
        R = self.R
        ops.call("router_fwd", ws["router_in"], p.f32("moe.router.0.weight"), p.f32("moe.router.0.bias"),
                 p.f32("moe.router.2.weight"), p.f32("moe.router.2.bias"), ws["router_h"], ws["probs"], ws["idx"],
                 ws["gates"], B, Dv, c.router_hidden, E, k)
        ops.call("dispatch", ws["idx"], B, k, E, P, Nt, ws["slot_of"], ws["item_of_slot"], ws["expert_of_slot"],
                 ws["row_off"], ws["tiles"], ws["tile_count"], self.max_tiles, ws["rowmap"])
        grp = self._expert_tiles
        for s, l in enumerate(c.stage_layers()):
            if c.expert_fp8:
                # e4m3 weights (per-output-channel scales) x e4m3 activation rows (one dynamic scale per row) on the fp8 MFMA
                self._fp8_gemm(ws[f"x{l}"], Dv, ws["rowmap"], None, p.q8(f"moe.proj.{s}.weight"), p.s8(f"moe.proj.{s}.weight"),
                               p.f32(f"moe.proj.{s}.bias"), ws["G"][s], Do, Dv, 1)
                self._fp8_gemm(ws["G"][s], Do, None, None, p.q8("moe.attn0.weight"), p.s8("moe.attn0.weight"), p.f32("moe.attn0.bias"),
                               ws["H1"][s], Dh, Do, 1)
                continue
            ops.gemm_nt(ws[f"x{l}"], p.w16(f"moe.proj.{s}.weight"), ws["G"][s], bias=p.f32(f"moe.proj.{s}.bias"),
                        a_rowmap=ws["rowmap"], stride_b=Do * Dv, stride_bias=Do, epi=ops.EPI_RELU, **grp(Dv))   # swin.py:40-41
            ops.gemm_nt(ws["G"][s], p.w16("moe.attn0.weight"), ws["H1"][s], bias=p.f32("moe.attn0.bias"),
                        stride_b=Dh * Do, stride_bias=Dh, epi=ops.EPI_RELU, **grp(Do))                      # swin.py:25-27
        ops.call("scale_attn_fwd", ws["G"], ws["H1"], p.f32("moe.attn2.weight"), p.f32("moe.attn2.bias"),
                 ws["expert_of_slot"], P, ws["eout"], ws["wts"], R, Do, Dh)
        ops.call("combine_fwd", ws["eout"], ws["slot_of"], ws["gates"], ws["img_l"], B, k, P, Do)
        ops.call("mean_tokens", ws["img_l"], ws["img_g"], B, P, Do, 0, P)              # swin.py:112

    # ------------------------------------------------------------------------------------------
    # text tower forward (frozen; transformer.py:116-130 post-norm; text_encoder.py:92-144)
    # ------------------------------------------------------------------------------------------
    def forward_text(self, ids: torch.Tensor, attn_mask: torch.Tensor, token_type: Optional[torch.Tensor] = None,
                     embedded: Optional[torch.Tensor] = None):
        """`embedded` [B, T, d_t] bf16: use these rows as the encoder's input instead of the embedding front-end's output (the
        reference's TransformerEncoder fixture is fed to the post-norm blocks this way, tests/test_ref_fixtures_gpu.py)."""
        c, ws, t = self.cfg, self.ws, self.params.text
        B, T = ids.shape
        if B != self.B or T != c.max_len:
            raise ValueError("forward_text: call forward_image first with the same batch; T must equal cfg.max_len")
        if self.train_text and embedded is None:
            return self._forward_text_train(ids, attn_mask, token_type)
        Dt, H = c.d_t, c.n_head_t
        ids32 = ids.to(I32).contiguous()
        tt32 = token_type.to(I32).contiguous() if token_type is not None else None
        km = attn_mask.to(torch.uint8).contiguous()
        st = ws["tstat"]
        L, last = c.n_layer_t, c.last_n_layers
        if last < 1 or last > 4:
            raise ValueError("last_n_layers must be in 1..4")
        # hidden_states = [embedding output, layer 1 .. layer L]; the reference sums hidden_states[-last:]
        # (text_encoder.py:97-103), which includes the embedding output when last > L
        first_sel = max(0, L + 1 - last)
        n_sel = L + 1 - first_sel
        hs = []
        x = ws["ths0"] if first_sel == 0 else ws["tx0"]
        # VARIABLE LENGTH: the tower runs on the tokens with attention mask 1 only (55 % of the B x T positions for captions of 8..77 words),
        # packed in (caption, position) order by a device-side scan - GEMM / LayerNorm row counts and the attention's sequence offsets stay
        # on the device, nothing is copied to the host.  Padding tokens never reach a result: they are masked keys and carry no word
        # (text_encoder.py:32-90).  MEDMOE_TEXT_VARLEN=0 computes all B x T positions as the reference does.
        vl = self.text_varlen and B <= 1024 and T <= 80
        if vl:
            pk = ws["tpack"]
            tok_row, src_row, seq_off, cnt = pk[:B * T], pk[B * T:2 * B * T], pk[2 * B * T:2 * B * T + B + 1], pk[2 * B * T + B + 1:2 * B * T + B + 2]
            ops.call("text_pack", km, tok_row, src_row, seq_off, cnt, B, T)
            ops.call("text_embed_ln_packed", ids32, tt32, t["word_embeddings"], t["position_embeddings"], t["token_type_embeddings"],
                     t["emb_layernorm.weight"], t["emb_layernorm.bias"], x, B, T, Dt, c.vocab, c.eps_t, src_row, cnt)
            if embedded is not None:       # packed rows of the given input (rows past the count are never read)
                x[:B * T].copy_(embedded.reshape(B * T, Dt).to(BF).index_select(0, src_row.long()))
            gemm = lambda a, w_, out, **kw: ops.gemm_nt_rows(a, w_, out, cnt, **kw)
            ln = lambda xin, g_, b_, y: ops.call("layernorm_fwd_rows", xin, g_, b_, y, st[0], st[1], B * T, Dt, c.eps_t, 0, cnt)
            attn = lambda: ops.call("attn_fwd_varlen", ws["tqkv"], ws["tatt"], ws["tlse"], seq_off, B, T, H, 64)
        else:
            ops.call("text_embed_ln", ids32, tt32, t["word_embeddings"], t["position_embeddings"], t["token_type_embeddings"],
                     t["emb_layernorm.weight"], t["emb_layernorm.bias"], x, B, T, Dt, c.vocab, c.eps_t)
            if embedded is not None:
                x[:B * T].copy_(embedded.reshape(B * T, Dt).to(BF))
            gemm = lambda a, w_, out, **kw: ops.gemm_nt(a, w_, out, **kw)
            ln = lambda xin, g_, b_, y: ops.layernorm_fwd(xin, g_, b_, y, st[0], st[1], c.eps_t)
            attn = lambda: ops.attn_fwd(ws["tqkv"], ws["tatt"], ws["tlse"], km, B, T, H)
        if first_sel == 0:
            hs.append(x)
        for l in range(L):
            b = f"layer.{l}."
            gemm(x, t[b + "attention.input_proj.weight"], ws["tqkv"], bias=t[b + "attention.input_proj.bias"])
            attn()
            gemm(ws["tatt"], t[b + "attention.output_proj.weight"], ws["tx1"], bias=t[b + "attention.output_proj.bias"], residual=x)
            ln(ws["tx1"], t[b + "attention_layernorm.weight"], t[b + "attention_layernorm.bias"], ws["tr"])
            gemm(ws["tr"], t[b + "feedforward.model.0.weight"], ws["th"], bias=t[b + "feedforward.model.0.bias"], epi=ops.EPI_GELU)
            gemm(ws["th"], t[b + "feedforward.model.2.weight"], ws["tx1"], bias=t[b + "feedforward.model.2.bias"], residual=ws["tr"])
            j = l + 1 - first_sel
            out = ws[f"ths{j}"] if j >= 0 else (ws["tx2"] if x is ws["tx0"] else ws["tx0"])
            ln(ws["tx1"], t[b + "feedforward_layernorm.weight"], t[b + "feedforward_layernorm.bias"], out)
            if j >= 0:
                hs.append(out)
            x = out
        assert len(hs) == n_sel
        self._text_last = (hs[-1], seq_off if vl else None)
        if self._seg is None:
            self.prefetch_cap_lens(ids)
        seg = self._seg
        self._seg = None
        h = hs + [None] * (4 - len(hs))
        if vl:
            ops.call("text_aggregate_packed", h[0], h[1], h[2], h[3], len(hs), seg, tok_row, ws["words"], ws["words32"], ws["txt_g"], B, T, Dt)
        else:
            ops.call("text_aggregate", h[0], h[1], h[2], h[3], len(hs), seg, ws["words"], ws["words32"], ws["txt_g"], B, T, Dt)

    def _forward_text_train(self, ids, attn_mask, token_type):
        """The text pass of a TRAINABLE tower (cfg.freeze_text = False): all B x T positions (key-masked attention, as the reference computes
        them, text_encoder.py:92-117), every layer's activations kept in the workspace for `backward_text`."""
        c, ws, t = self.cfg, self.ws, self.params.text
        B, T = ids.shape
        Dt, H, L, last = c.d_t, c.n_head_t, c.n_layer_t, c.last_n_layers
        if last < 1 or last > 4:
            raise ValueError("last_n_layers must be in 1..4")
        ids32 = ids.to(I32).contiguous()
        tt32 = token_type.to(I32).contiguous() if token_type is not None else None
        km = attn_mask.to(torch.uint8).contiguous()
        self._tt_state = (ids32, tt32, km)
        x = ws["t_x0"]
        ops.call("text_embed_ln", ids32, tt32, t["word_embeddings"], t["position_embeddings"], t["token_type_embeddings"],
                 t["emb_layernorm.weight"], t["emb_layernorm.bias"], x, B, T, Dt, c.vocab, c.eps_t)
        for l in range(L):
            b = f"layer.{l}."
            st1, st2 = ws[f"t_st1{l}"], ws[f"t_st2{l}"]
            ops.gemm_nt(x, t[b + "attention.input_proj.weight"], ws[f"t_qkv{l}"], bias=t[b + "attention.input_proj.bias"])
            ops.attn_fwd(ws[f"t_qkv{l}"], ws[f"t_att{l}"], ws[f"t_lse{l}"], km, B, T, H)
            ops.gemm_nt(ws[f"t_att{l}"], t[b + "attention.output_proj.weight"], ws[f"t_x1{l}"], bias=t[b + "attention.output_proj.bias"], residual=x)
            ops.layernorm_fwd(ws[f"t_x1{l}"], t[b + "attention_layernorm.weight"], t[b + "attention_layernorm.bias"], ws[f"t_r{l}"], st1[0], st1[1], c.eps_t)
            ops.gemm_nt(ws[f"t_r{l}"], t[b + "feedforward.model.0.weight"], ws[f"t_h{l}"], bias=t[b + "feedforward.model.0.bias"],
                        aux=ws[f"t_dg{l}"], epi=ops.EPI_GELU_DAUX)
            ops.gemm_nt(ws[f"t_h{l}"], t[b + "feedforward.model.2.weight"], ws[f"t_x2{l}"], bias=t[b + "feedforward.model.2.bias"], residual=ws[f"t_r{l}"])
            ops.layernorm_fwd(ws[f"t_x2{l}"], t[b + "feedforward_layernorm.weight"], t[b + "feedforward_layernorm.bias"], ws[f"t_x{l + 1}"],
                              st2[0], st2[1], c.eps_t)
            x = ws[f"t_x{l + 1}"]
        first_sel = max(0, L + 1 - last)                          # hidden_states[-last:] of [embedding output, layer 1 .. layer L]
        hs = [ws[f"t_x{j}"] for j in range(first_sel, L + 1)]
        self._text_first_sel = first_sel
        self._text_last = (hs[-1], None)
        if self._seg is None:
            self.prefetch_cap_lens(ids)
        seg = self._seg
        self._seg = None
        self._seg_used = seg
        h = hs + [None] * (4 - len(hs))
        ops.call("text_aggregate", h[0], h[1], h[2], h[3], len(hs), seg, ws["words"], ws["words32"], ws["txt_g"], B, T, Dt)

    def backward_text(self, d_words: Optional[torch.Tensor], d_txt_g: Optional[torch.Tensor]):
        """Back-propagate d loss / d word embeddings (fp32 [B, T, D]) and d loss / d sentence embeddings (fp32 [B, D]) through the aggregation,
        the post-norm blocks (transformer.py:116-130) and the embedding front-end into the text store's flat gradient buffer."""
        c, ws, ts = self.cfg, self.ws, self.tstore
        B, T, Dt, H, L = self.B, c.max_len, c.d_t, c.n_head_t, c.n_layer_t
        ids32, tt32, km = self._tt_state
        w16t, grad, f32 = ts.w16t, ts.grad, ts.f32
        dH = ws["t_dH"]
        ops.call("text_aggregate_bwd", d_words, d_txt_g, self._seg_used, dH, B, T, Dt)
        dy, d1, d2 = ws["t_da"], ws["t_db"], ws["t_dc"]
        dy.copy_(dH)                                                 # the last layer's output is always among the summed states
        for l in range(L - 1, -1, -1):
            b = f"layer.{l}."
            st1, st2 = ws[f"t_st1{l}"], ws[f"t_st2{l}"]
            # y = LN2(x2), x2 = r + FC2(GELU(FC1(r)))
            ops.layernorm_bwd(dy, ws[f"t_x2{l}"], st2[0], st2[1], f32(b + "feedforward_layernorm.weight"), d1,
                              grad(b + "feedforward_layernorm.weight"), grad(b + "feedforward_layernorm.bias"))
            ops.gemm_tn(d1, ws[f"t_h{l}"], grad(b + "feedforward.model.2.weight"), db=grad(b + "feedforward.model.2.bias"))
            ops.gemm_nt(d1, w16t(b + "feedforward.model.2.weight"), ws["t_dz"], aux=ws[f"t_dg{l}"], epi=ops.EPI_MUL_AUX)
            ops.gemm_tn(ws["t_dz"], ws[f"t_r{l}"], grad(b + "feedforward.model.0.weight"), db=grad(b + "feedforward.model.0.bias"))
            ops.gemm_nt(ws["t_dz"], w16t(b + "feedforward.model.0.weight"), d2, residual=d1)                  # d r = d x2 + dz W1
            # r = LN1(x1), x1 = x + out_proj(attention(qkv(x)))
            ops.layernorm_bwd(d2, ws[f"t_x1{l}"], st1[0], st1[1], f32(b + "attention_layernorm.weight"), d1,
                              grad(b + "attention_layernorm.weight"), grad(b + "attention_layernorm.bias"))
            ops.gemm_tn(d1, ws[f"t_att{l}"], grad(b + "attention.output_proj.weight"), db=grad(b + "attention.output_proj.bias"))
            ops.gemm_nt(d1, w16t(b + "attention.output_proj.weight"), ws["t_datt"])
            ops.attn_bwd(ws[f"t_qkv{l}"], ws[f"t_att{l}"], ws["t_datt"], ws[f"t_lse{l}"], km, ws["t_dqkv"], ws["t_delta"], B, T, H)
            ops.gemm_tn(ws["t_dqkv"], ws[f"t_x{l}"], grad(b + "attention.input_proj.weight"), db=grad(b + "attention.input_proj.bias"))
            ops.gemm_nt(ws["t_dqkv"], w16t(b + "attention.input_proj.weight"), dy, residual=d1)               # d x = d x1 + dqkv Wqkv
            if l >= self._text_first_sel:                          # hidden_states[l] is one of the summed states too
                dy.add_(dH)
        g_word = grad("word_embeddings")
        ops.call("text_embed_ln_bwd", ids32, tt32, f32("word_embeddings"), f32("position_embeddings"), f32("token_type_embeddings"),
                 f32("emb_layernorm.weight"), dy, ws["t_dxemb"], grad("emb_layernorm.weight"), grad("emb_layernorm.bias"), g_word, B, T, Dt,
                 c.vocab, c.eps_t)
        dxe = ws["t_dxemb"].view(B, T, Dt)
        grad("position_embeddings").add_(dxe.sum(dim=0))
        tt = tt32.view(-1).long() if tt32 is not None else torch.zeros(B * T, device=self.device, dtype=torch.long)
        grad("token_type_embeddings").index_add_(0, tt, ws["t_dxemb"])

    def text_soft_target(self) -> torch.Tensor:
        """Caption-to-caption scores of the Soft-GLoRIA losses (medmoe_module.py:258-281 get_text_soft_target): the frozen text model's last
        hidden state at [CLS] (token_pooling :243-244), L2-normalised, all pairwise products -> fp32 [B, B].  The reference runs a second
        pretrained BertModel (`tool_bert`) for it; with `freeze_bert: true` that is the text tower's own BERT, whose last layer the
        forward pass just produced.  Call after forward_text."""
        h, seq_off = self._text_last
        B, T, Dt = self.B, self.cfg.max_len, self.cfg.d_t
        rows = seq_off[:B].long() if seq_off is not None else torch.arange(B, device=self.device) * T
        cls = h.view(-1, Dt).index_select(0, rows).float().contiguous()
        n = torch.empty(B, device=self.device); S = torch.empty(B, B, device=self.device)
        ops.call("rownorm", cls, n, B, Dt)
        ops.call("sgemm", cls, cls, S, B, B, Dt, Dt, 1, 1, Dt, B, 1.0, 0.0)
        ops.call("cos_scale", S, n, n, B, B, 1e-24)
        return S

    def _head(self, S, dS, rs, cs, w, accumulate, loss):
        """Cross-entropy against the diagonal, or (cfg.soft_label) the Soft-GLoRIA head, over the rows / columns of a [B, B] matrix."""
        c, B = self.cfg, self.B
        if c.soft_label:
            ops.call("soft_xent_strided", S, dS, self._soft, B, B, rs, cs, c.temp3, c.threshold0, c.threshold1, w, accumulate, loss)
        else:
            ops.call("ce_strided", S, dS, B, B, rs, cs, 0, c.temp3, w, accumulate, loss)

    def prefetch_cap_lens(self, ids: torch.Tensor):
        """Word-piece segment map + caption lengths (text_encoder.py:32-90) and an ASYNCHRONOUS copy of the lengths to
        the host.  train_step calls this first, when the device queue is empty, so the one host-side read of the
        step (class tables of the ragged local-loss layout) never waits for the towers and the host keeps issuing
        launches ahead of the device."""
        out = None
        if ids.is_cuda and "seg" in self.ws and tuple(self.ws["seg"].shape) == tuple(ids.shape):
            out = (self.ws["seg"], self.ws["cap"])                  # persistent buffers: no allocation per step, stable addresses for graph capture
        seg, cap = self.vocab.segment_map(ids, out) if out is not None else self.vocab.segment_map(ids)
        self._seg, self.cap_lens = seg, cap
        if cap.is_cuda:
            if self._cap_host is None or self._cap_host.numel() != cap.numel():
                self._cap_host = torch.empty(cap.numel(), dtype=cap.dtype, pin_memory=True)
                self._cap_event = torch.cuda.Event()
            self._cap_host.copy_(cap, non_blocking=True)
            self._cap_event.record()
        else:
            self._cap_host = cap

    def _cap_lens_host(self) -> np.ndarray:
        if self.cap_lens.is_cuda:
            self._cap_event.synchronize()
        return self._cap_host.numpy().astype(np.int64)

    # ------------------------------------------------------------------------------------------
    # losses: forward values + gradients w.r.t. img_g / img_l (text is frozen)
    # ------------------------------------------------------------------------------------------
    def global_loss(self, loss_scale: float = 1.0):
        """GLoRIA global loss (losses.py:766-794) of ws["img_g"] against ws["txt_g"], forward and backward: zeroes ws["loss_parts"], adds the
        weighted loss to loss_parts[2], leaves dL/d img_g in ws["d_img_g"] (and dL/d txt_g in ws["d_txt_g"] when the text tower trains).
        Any image encoder that fills ws["img_g"] can call it (the ViT towers here, the Swin-T encoder in medmoe_amd.swin_engine)."""
        c, ws = self.cfg, self.ws
        B, Do = self.B, c.d_out
        ws["loss_parts"].zero_()
        lp = ws["loss_parts"]
        # ---- rows = images, cols = captions ----
        img_g, txt_g = ws["img_g"], ws["txt_g"]
        wg = c.w_global * loss_scale / B
        if c.soft_label:
            if self.dist:
                raise NotImplementedError("soft_label with more than one rank: the reference's Soft-GLoRIA losses are written for one process "
                                          "(losses.py:826-883 has no gather)")
            self._soft = self.text_soft_target()
        if not self.dist:
            ops.call("rownorm", img_g, ws["na"], B, Do)
            ops.call("rownorm", txt_g, ws["nb"], B, Do)
            ops.call("sgemm", img_g, txt_g, ws["S"], B, B, Do, Do, 1, 1, Do, B, 1.0, 0.0)
            ops.call("cos_scale", ws["S"], ws["na"], ws["nb"], B, B, 1e-8)
            self._head(ws["S"], ws["dS"], B, 1, wg, 0, lp[2:])
            self._head(ws["S"], ws["dS"], 1, B, wg, 1, lp[2:])
            cb = None
            if self.train_text:
                cb = ws["cb"]; cb.zero_()
            ops.call("cos_scale_bwd", ws["dS"], ws["S"], ws["na"], ws["nb"], ws["ca"], cb, B, B, 1e-8)
            ops.call("sgemm", ws["dS"], txt_g, ws["d_img_g"], B, Do, B, B, 1, Do, 1, Do, 1.0, 0.0)
            ops.call("add_rowscaled", ws["d_img_g"], img_g, ws["ca"], B, Do)
            if self.train_text:                                   # the caption side of the same matrix: d txt_g = dS^T img_g + cb txt_g
                ops.call("sgemm", ws["dS"], img_g, ws["d_txt_g"], B, Do, B, 1, B, Do, 1, Do, 1.0, 0.0)
                ops.call("add_rowscaled", ws["d_txt_g"], txt_g, cb, B, Do)
        else:
            # all-gather + local-rows InfoNCE (losses.py:503-524,566-572 with GLoRIA's cosine/temp3):
            # rows = my images vs ALL captions, and my captions vs ALL images; labels offset by rank
            from . import dist as D_
            Bg = B * self.world
            off = D_.label_offset(B)
            img_all, txt_all = D_.gather_embeddings(img_g, txt_g)
            ops.call("rownorm", img_g, ws["na"], B, Do); ops.call("rownorm", txt_all, ws["nb"], Bg, Do)
            ops.call("sgemm", img_g, txt_all, ws["S"], B, Bg, Do, Do, 1, 1, Do, Bg, 1.0, 0.0)
            ops.call("cos_scale", ws["S"], ws["na"], ws["nb"], B, Bg, 1e-8)
            ops.call("ce_strided", ws["S"], ws["dS"], B, Bg, Bg, 1, off, c.temp3, wg, 0, lp[2:])
            cb1 = None
            if self.train_text:
                cb1 = ws["cb1"]; cb1.zero_()
            ops.call("cos_scale_bwd", ws["dS"], ws["S"], ws["na"], ws["nb"], ws["ca"], cb1, B, Bg, 1e-8)
            ops.call("sgemm", ws["dS"], txt_all, ws["d_img_g"], B, Do, Bg, Bg, 1, Do, 1, Do, 1.0, 0.0)
            ops.call("add_rowscaled", ws["d_img_g"], img_g, ws["ca"], B, Do)
            if self.train_text:       # my images against ALL captions: the gathered captions' gradient, summed over ranks, my slice comes back
                ops.call("sgemm", ws["dS"], img_g, ws["d_txt_all"], Bg, Do, B, 1, Bg, Do, 1, Do, 1.0, 0.0)
                ops.call("add_rowscaled", ws["d_txt_all"], txt_all, cb1, Bg, Do)
            ops.call("rownorm", txt_g, ws["na2"], B, Do); ops.call("rownorm", img_all, ws["nb2"], Bg, Do)
            ops.call("sgemm", txt_g, img_all, ws["S2"], B, Bg, Do, Do, 1, 1, Do, Bg, 1.0, 0.0)
            ops.call("cos_scale", ws["S2"], ws["na2"], ws["nb2"], B, Bg, 1e-8)
            ops.call("ce_strided", ws["S2"], ws["dS2"], B, Bg, Bg, 1, off, c.temp3, wg, 0, lp[2:])
            ws["cb2"].zero_()
            ops.call("cos_scale_bwd", ws["dS2"], ws["S2"], ws["na2"], ws["nb2"], ws["ca2"], ws["cb2"], B, Bg, 1e-8)
            # d img_all = dM^T txt_local + cb * img_all ; summed over ranks, my slice comes back
            ops.call("sgemm", ws["dS2"], txt_g, ws["d_img_all"], Bg, Do, B, 1, Bg, Do, 1, Do, 1.0, 0.0)
            ops.call("add_rowscaled", ws["d_img_all"], img_all, ws["cb2"], Bg, Do)
            ws["d_img_g"].add_(D_.scatter_key_grads(ws["d_img_all"]))
            if self.train_text:       # my captions against ALL images (rows of S2): d txt_g = dS2 img_all + ca2 txt_g, plus the scattered part
                ops.call("sgemm", ws["dS2"], img_all, ws["d_txt_g"], B, Do, Bg, Bg, 1, Do, 1, Do, 1.0, 0.0)
                ops.call("add_rowscaled", ws["d_txt_g"], txt_g, ws["ca2"], B, Do)
                ws["d_txt_g"].add_(D_.scatter_key_grads(ws["d_txt_all"]))

    def forward_backward_losses(self, labels: torch.Tensor, loss_scale: float = 1.0):
        c, ws = self.cfg, self.ws
        B, P, Do, T = self.B, c.n_patch, c.d_out, c.max_len
        self.global_loss(loss_scale)
        lp = ws["loss_parts"]
        # ---- GLoRIA local (losses.py:961-1026) ----
        HWp, Tp, GW = self.HWp, self.Tp, self.GW
        ctx = ws["img_l"].view(B * P, Do)
        self._d_words = None
        if self.train_text:
            if not self.local_t or (self.dist and c.local_loss_global):
                raise NotImplementedError("trainable text tower: the word gradient of the local loss is built for the 196- / 64-region "
                                          "geometries (csrc/pair3.hip), rank-local captions")
            return self._local_loss_transposed_words(loss_scale)
        if not self.local_fast:
            return self._local_loss_generic(loss_scale)
        if self.dist and c.local_loss_global:
            if not self.local_t:
                raise NotImplementedError("local_loss_global needs the transposed local-loss path (196 / 64 regions)")
            return self._local_loss_global(loss_scale)
        if self.local_t:
            return self._local_loss_transposed(loss_scale)
        # RAGGED pair matrices: the [B*HWp, B*Tp] score / gradient matrices are the largest tensors of the step
        # (3 x 35 GB at B = 1024) and most of their columns are caption padding.  Captions are grouped into
        # length classes (<= 16, 32, ... words); class c stores its members side by side, 16*c columns each, so a
        # row is Kp = sum_i pad16(len_i) (rounded up to 64) columns instead of B*Tp.  Needs the lengths on the
        # host: ONE small device-to-host copy per step (the only host sync of the step).
        perm, col_of_cap, ntts, cap_of_chunk, classes, Kc, Kp = ragged_layout(self._cap_lens_host(), T, Tp)
        meta = torch.from_numpy(np.concatenate((perm, col_of_cap, 16 * ntts, cap_of_chunk)).astype(np.int32)).to(self.device, non_blocking=True)
        d_perm, d_col, d_tp, d_chunk = meta[:B], meta[B:2 * B], meta[2 * B:3 * B], meta[3 * B:]
        self._pair_buffers(Kp)
        rag = lambda name: ws[name].view(-1)[:B * HWp * Kp].view(B * HWp, Kp)
        lA, ldS, lU = rag("l_A"), rag("l_dS"), rag("l_U")
        wT = ws["wT"].view(-1)[:Do * Kp].view(Do, Kp)
        if Kp > Kc:
            for t_ in (lA, ldS, lU, wT):
                t_[:, Kc:].zero_()
        ops.call("words_prep_ragged", ws["words"], ws["wn"], wT, B, T, Tp, Do, d_col, d_tp, Kp)
        ops.gemm_nt(ctx, ctx, ws["gmp"], c_rowmap=ws["gm_crowmap"], tiles=ws["img_tiles"], tile_count=ws["img_tile_count"],
                    max_tiles=ws["img_tiles"].shape[0], stride_b=P * Do, M=B * P, N=P, col_perm=True)
        for ntt, start, n_c, cbase in classes:
            members = d_perm[start:start + n_c]
            # all word-region scores of the class as ONE tiled GEMM with the word-softmax fused (A1 + row LSE); the
            # A1 tiles live in the l_A buffer (each pair's tile is read before the same workgroup overwrites it)
            ops.call("local_scores_ragged", ctx, ws["words"], self.cap_lens, lA, ws["l_lse"], B, B, P, T, Do, members, n_c, ntt, cbase, Kp)
            # single pass over the (image, caption) pairs: sim AND the gradients for dL/dsim = 1 ...
            ops.call("local_pair2_ragged", lA, ws["l_lse"], ws["gmp"], ws["wn"], self.cap_lens, None, ws["sim"], ldS, lU,
                     B, B, P, T, c.temp1, c.temp2, 1e-8, members, n_c, ntt, cbase, Kp)
        wl = c.w_local * loss_scale / B
        self._head(ws["sim"], ws["gsim"], B, 1, wl, 0, lp[3:])
        self._head(ws["sim"], ws["gsim"], 1, B, wl, 1, lp[3:])
        # ... then the CE over the sim matrix supplies the per-pair factor
        ops.call("scale_blocks_ragged", ldS, lU, ws["gsim"], B, B, HWp, d_chunk, Kp)
        ops.gemm_nt(ldS, wT, ws["dC32"])                                                    # dC = dS . W
        if Kp >= 128 and ws["imgp_tiles256"].shape[0]:                                      # dGm_b = U_b A_b^T
            ops.gemm_nt(lU, lA, ws["dGm"], tiles=ws["imgp_tiles256"], tile_count=ws["imgp_tile256_count"], max_tiles=B,
                        stride_b=HWp * Kp, M=B * HWp, N=HWp, tile_rows=256)
        else:
            ops.gemm_nt(lU, lA, ws["dGm"], tiles=ws["imgp_tiles"], tile_count=ws["imgp_tile_count"],
                        max_tiles=ws["imgp_tiles"].shape[0], stride_b=HWp * Kp, M=B * HWp, N=HWp)
        ops.gemm_tn(ws["dGm"], ctx, ws["dC32"].view(B, HWp, Do), x_rowmap=ws["ctx_xmap"], row_off=ws["imgp_row_off"], n_groups=B,
                    stride_w=HWp * Do, nsplit=1, M=B * HWp)                                  # dC_b += dGm_b . ctx_b
        ops.call("unpad_cast", ws["dC32"], ws["d_img_l"], B, P, HWp, Do)

    def _local_loss_transposed(self, loss_scale: float):
        """GLoRIA local loss (losses.py:961-1026) on TRANSPOSED ragged pair matrices [Kp caption-word rows][B*HWp region columns]
        (csrc/pair3.hip, medmoe_amd/local_transposed.py): forward launches -> head over the similarity matrix (cross-entropy or
        Soft-GLoRIA) -> backward launches and the two wgrad-shaped GEMMs.  The pair matrices are views of l_dS / l_A (/ l_U)."""
        c, ws, B = self.cfg, self.ws, self.B
        lp = ws["loss_parts"]
        if self._tl is None or self._tl.B != B or self._tl.ws is not ws:
            self._tl = TransposedLocalLoss(B, c.n_patch, c.max_len, c.d_out, self.HWp, self.Tp, self.HWq, self.device, ws,
                                           self._pair_buffers, self.local_gram)
        self._tl.forward(ws["img_l"].view(B * c.n_patch, c.d_out), ws["words"], self.cap_lens, self._cap_lens_host(), c.temp1, c.temp2)
        wl = c.w_local * loss_scale / B
        self._head(ws["sim"], ws["gsim"], B, 1, wl, 0, lp[3:])
        self._head(ws["sim"], ws["gsim"], 1, B, wl, 1, lp[3:])
        self._tl.backward(ws["gsim"], ws["d_img_l"])

    def _local_loss_transposed_words(self, loss_scale: float):
        """The same loss with the word embeddings differentiated too (trainable text tower): `TransposedLocalLoss` in its word-gradient mode
        (row-major pair matrices of its own; d words = dS . ctx + the word-norm term, medmoe_amd/local_transposed.py)."""
        c, ws, B = self.cfg, self.ws, self.B
        lp = ws["loss_parts"]
        if self._tlw is None or self._tlw.B != B:
            self._tlw = TransposedLocalLoss.standalone(B, c.n_patch, c.max_len, c.d_out, self.device, word_grad=True)
        tl = self._tlw
        sim = tl.forward(ws["img_l"].view(B * c.n_patch, c.d_out), ws["words"], self.cap_lens, self._cap_lens_host(), c.temp1, c.temp2)
        ws["sim"].copy_(sim)
        wl = c.w_local * loss_scale / B
        self._head(ws["sim"], ws["gsim"], B, 1, wl, 0, lp[3:])
        self._head(ws["sim"], ws["gsim"], 1, B, wl, 1, lp[3:])
        self._d_words = tl.backward(ws["gsim"], ws["d_img_l"])

    def _local_loss_global(self, loss_scale: float):
        """cfg.local_loss_global under data parallelism: this rank's images against the captions of every rank (the
        variant the reference does not have - its local loss stays rank-local, losses.py:961-1026).  Words and caption lengths are
        all-gathered (the text tower is frozen: no gradient goes back), the rank's [B, B_g] block of similarities is all-gathered into
        the [B_g, B_g] matrix, both cross-entropies run over it on every rank (1 M elements), and the rank back-propagates its own
        rows.  Gradients are averaged over ranks afterwards, so the rows carry W / B_g = 1 / B."""
        from . import dist as D_
        c, ws, B, W = self.cfg, self.ws, self.B, self.world
        Bg = B * W
        lp = ws["loss_parts"]
        words_all = D_.gather_rows(ws["words"])
        caps_all = D_.gather_rows(self.cap_lens)
        caps_host = caps_all.cpu().numpy().astype(np.int64)
        if self._tlg is None or self._tlg.B != B or self._tlg.Bc != Bg:
            self._tlg = TransposedLocalLoss.standalone(B, c.n_patch, c.max_len, c.d_out, self.device, self.local_gram, Bc=Bg)
        sim = self._tlg.forward(ws["img_l"].view(B * c.n_patch, c.d_out), words_all, caps_all, caps_host, c.temp1, c.temp2)
        S = D_.gather_rows(sim)                                       # [B_g, B_g]: rows = images in rank order, columns = captions
        G = torch.empty_like(S)
        wl = c.w_local * loss_scale / Bg
        ops.call("ce_strided", S, G, Bg, Bg, Bg, 1, 0, c.temp3, wl, 0, lp[3:])
        ops.call("ce_strided", S, G, Bg, Bg, 1, Bg, 0, c.temp3, wl, 1, lp[3:])
        r0 = D_.label_offset(B)
        self._tlg.backward((G[r0:r0 + B] * float(W)).contiguous(), ws["d_img_l"])

    def _pair_buffers(self, Kp: int):
        """(Re)allocate the ragged pair matrices for rows of Kp columns (capacity grows by 10 % steps, never above B*Tp)."""
        if Kp <= self._pair_cap:
            return
        ws, B = self.ws, self.B
        Kmax = (B * self.Tp + 63) // 64 * 64
        cap = min(Kmax, (int(Kp * 1.1) + 63) // 64 * 64)
        for name in ("l_A", "l_dS", "l_U", "wT", "words_r", "l_stats3", "l_d2"):
            ws.pop(name, None)                                 # release before allocating: the old and new sets must not coexist
        rows = B * (max(self.HWp, self.HWq) if self.local_t else self.HWp)
        for name in (("l_A", "l_dS") if (self.local_t and self.local_gram) else ("l_A", "l_dS", "l_U")):
            ws[name] = torch.empty((rows, cap), device=self.device, dtype=BF)
        ws["wT"] = torch.empty((self.cfg.d_t, cap), device=self.device, dtype=BF)
        if self.local_t:
            ws["words_r"] = torch.empty((cap, self.cfg.d_t), device=self.device, dtype=BF)
            ws["l_stats3"] = torch.empty((B, cap, 2), device=self.device, dtype=torch.float32)
            if self.local_gram:
                ws["l_d2"] = torch.empty((B, cap), device=self.device, dtype=torch.float32)
        self._pair_cap = cap

    def _local_loss_generic(self, loss_scale: float):
        """GLoRIA local loss for a geometry without LDS-tiled pair kernels (loss.hip "GENERIC-GEOMETRY"): the reference's own
        formulation - weighted context = bmm(ctx, attn) (losses.py:732), cosine against the word (:690-695, :1002) - as grouped
        GEMMs over the uniform pair matrices [B*HWp, B*Tp], plus four elementwise kernels."""
        c, ws = self.cfg, self.ws
        B, P, Do, T = self.B, c.n_patch, c.d_out, c.max_len
        HWp, Tp = self.HWp, self.Tp
        Kp = ws["l_A"].shape[1]                                   # B*Tp rounded up to the GEMM k-step
        ctx = ws["img_l"].view(B * P, Do)
        lp_, LA, DA, wT = ws["l_LP"], ws["l_A"], ws["l_dS"], ws["wT"]
        if Kp > B * Tp:
            for t_ in (lp_, LA, DA, wT):
                t_[:, B * Tp:].zero_()
            ws["l_DWC"][:, B * Tp:].zero_()
        lp = ws["loss_parts"]
        ops.call("words_prep_ragged", ws["words"], ws["wn"], wT, B, T, Tp, Do, ws["l_col"], ws["l_tp"], Kp)
        ops.call("local_scores_ragged", ctx, ws["words"], self.cap_lens, lp_, ws["l_lse"], B, B, P, T, Do, ws["l_members"], B, Tp // 16, 0, Kp)
        ops.call("local_gen_fwd_a", lp_, self.cap_lens, LA, B, B, P, HWp, T, Tp, c.temp1, Kp)
        ws["l_WC"].zero_()
        ops.gemm_tn(LA, ctx, ws["l_WC"], x_rowmap=ws["ctx_xmap"], row_off=ws["imgp_row_off"], n_groups=B, stride_w=Kp * Do, nsplit=1,
                    M=B * HWp)                                    # wctx_b = A_b^T ctx_b
        ops.call("local_gen_cos", ws["l_WC"], ws["words"], ws["wn"], self.cap_lens, ws["sim"], ws["l_stats"], ws["l_sume"], B, B, T, Tp, Do,
                 c.temp2, 1e-8, Kp)
        wl = c.w_local * loss_scale / B
        self._head(ws["sim"], ws["gsim"], B, 1, wl, 0, lp[3:])
        self._head(ws["sim"], ws["gsim"], 1, B, wl, 1, lp[3:])
        ops.call("local_gen_dwctx", ws["l_WC"], ws["words"], ws["wn"], self.cap_lens, ws["gsim"], ws["l_stats"], ws["l_sume"], ws["l_DWC"],
                 B, B, T, Tp, Do, c.temp2, 1e-8, Kp)
        ops.call("transpose_many", ws["l_DWC"], ws["l_DWCt"], ws["l_trtab"], B, ((Kp + 63) // 64) * ((Do + 63) // 64))
        if self.local_dense:
            grp = dict(tiles=ws["imgp_tiles256g"], tile_count=ws["imgp_tile256g_count"], max_tiles=ws["imgp_tiles256g"].shape[0], M=B * HWp,
                       tile_rows=256)
            ops.gemm_nt(ctx, ws["l_DWC"], DA, stride_b=Kp * Do, N=Kp, **grp)                                # dA_b = ctx_b dwctx_b^T
            ops.gemm_nt(LA, ws["l_DWCt"], ws["l_X1"], stride_b=Do * Kp, N=Do, **grp)                        # d ctx_b (direct) = A_b dwctx_b
            ops.call("local_gen_bwd_s", lp_, LA, DA, self.cap_lens, B, B, P, HWp, T, Tp, c.temp1, Kp)       # dS over dA in place
            ops.gemm_nt(DA, wT, ws["d_img_l"].view(B * P, Do), residual=ws["l_X1"])                         # d ctx = dS . W + the direct part
            return
        grp = dict(tiles=ws["imgp_tiles"], tile_count=ws["imgp_tile_count"], max_tiles=ws["imgp_tiles"].shape[0], M=B * HWp)
        ops.gemm_nt(ctx, ws["l_DWC"], DA, a_rowmap=ws["ctx_xmap"], stride_b=Kp * Do, N=Kp, **grp)          # dA_b = ctx_b dwctx_b^T
        ops.gemm_nt(LA, ws["l_DWCt"], ws["dC32b"], stride_b=Do * Kp, N=Do, **grp)                           # d ctx_b (direct) = A_b dwctx_b
        ops.call("local_gen_bwd_s", lp_, LA, DA, self.cap_lens, B, B, P, HWp, T, Tp, c.temp1, Kp)           # dS over dA in place
        ops.gemm_nt(DA, wT, ws["dC32"])                                                                     # d ctx += dS . W
        ops.call("unpad_cast2", ws["dC32"], ws["dC32b"], ws["d_img_l"], B, P, HWp, Do)

    # ------------------------------------------------------------------------------------------
    # backward through MoE and the ViT
    # ------------------------------------------------------------------------------------------
    def backward(self, labels: Optional[torch.Tensor], loss_scale: float = 1.0, dprobs_ext: Optional[torch.Tensor] = None,
                 bucket_ready=None):
        """Back-propagate ws["d_img_l"] / ws["d_img_g"] (+ the router CE when `labels` is given, + an
        external dL/dprobs) through MoE and the ViT into the flat gradient buffer."""
        c, p, ws = self.cfg, self.params, self.ws
        B, P, Nt, Dv = self.B, c.n_patch, c.n_tok_v, c.d_v
        self._wgrad_begin()
        w_moe = self._moe_backward(labels, loss_scale, dprobs_ext)
        # ---- mean-pool backward: the router-input gradient onto every patch token of the final LayerNorm's output ----
        ops.call("broadcast_tokens", ws["drouter_in"], ws["dln"], B, Nt, Dv, 1, P, 1.0 / P)
        w_last = self._vit_backward(w_moe, bucket_ready)
        # ---- embeddings backward ----
        dx = ws["dxa"]
        ops.call("pos_cls_grad", dx, p.grad("vit.pos_embed"), p.grad("vit.cls_token"), B, Nt, Dv)
        ops.gemm_tn(dx, ws["im2col"], p.grad("vit.patch_embed.weight"), db=p.grad("vit.patch_embed.bias"),
                    g_rowmap=ws["rowmap_patch"], M=B * P)
        self._wait(w_last)                                   # join: every weight gradient is final before the optimiser / the caller
        if bucket_ready is not None:
            bucket_ready(0)              # patch / CLS / position embeddings: complete

    # The four weight-gradient GEMMs of a layer run on a SECOND stream: each only needs its gradient operand (event from the
    # main stream) and nothing downstream needs its result before the bucket all-reduce / the optimiser.  At small per-rank
    # batches the dgrad GEMMs leave most CUs idle in their last round of tiles (B = 128: 296 tiles of a K = 2304 dgrad on
    # 256 CUs); the concurrent wgrad fills them.  The scratch gradients (dx, dx2, dz, dqkv) are rewritten one layer later: the
    # main stream waits for the wgrad that read a buffer before the kernel that overwrites it.
    # (measured on one box, cfg2: per-rank batch 128 31.2 -> 29.8 ms, 256 55.9 -> 54.7 ms; at 1024 every GEMM already fills the chip for
    # ~28 rounds and the second stream costs 1.4 %, so it is used up to 131072 token rows)
    def _wgrad_begin(self):
        ws = self.ws
        M = self.B * self.cfg.n_tok_v
        self._wg_side = self._side_stream() if (self.overlap_wgrad and ws["dxa"].is_cuda and M <= 131072) else None
        self._wg_main = torch.cuda.current_stream() if self._wg_side is not None else None

    def _wgrad(self, *a, **kw):
        side, main = self._wg_side, self._wg_main
        if self.wgrad_staged and "row_off" not in kw and "x_rowmap" not in kw and "g_rowmap" not in kw:
            # plain wgrads: partial tiles through a scratch buffer + one summing kernel instead of 64 MB of fp32 atomics per launch
            # (medmoe_gemm_tn_staged); every launch of this method runs on one stream, so one scratch serves them all
            sc = self.ws.get("wg_scratch")
            if sc is None:
                sc = self.ws["wg_scratch"] = torch.empty(256 * 65536, device=self.device, dtype=F32)
            kw["scratch"] = sc
        if side is None:
            ops.gemm_tn(*a, **kw)
            return None
        ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.gemm_tn(*a, **kw)
        done = torch.cuda.Event(); done.record(side)
        return done

    def _wait(self, ev):
        if ev is not None:
            self._wg_main.wait_event(ev)

    def _moe_backward(self, labels: Optional[torch.Tensor], loss_scale: float = 1.0, dprobs_ext: Optional[torch.Tensor] = None):
        """MoE backward (swin.py:32-117): from ws["d_img_l"] / ws["d_img_g"] (+ router CE on `labels`, + an external dL/dprobs) to the
        stage-feature gradients ws["dF"], the router-input gradient ws["drouter_in"] and every expert / router weight gradient.
        Returns the event of the last expert wgrad on the second stream (None without one).  Call _wgrad_begin() first."""
        c, p, ws = self.cfg, self.params, self.ws
        B = self.B
        E, k, Do, Dh, Dv, P = c.n_expert, c.top_k, c.d_out, c.d_out // 2, c.d_v, c.n_patch
        R = self.R
        lab32 = labels.to(I32).contiguous() if labels is not None else None
        wgrad = self._wgrad
        use_gate = k > 1
        if use_gate:
            ws["dgate"].zero_()
        ops.call("scale_attn_bwd", ws["d_img_l"], ws["d_img_g"], ws["G"], ws["H1"], ws["wts"], p.f32("moe.attn2.weight"),
                 ws["eout"], ws["expert_of_slot"], ws["item_of_slot"], ws["gates"], k, P, ws["dG"], ws["dH1"],
                 p.grad("moe.attn2.weight"), p.grad("moe.attn2.bias"), ws["dgate"] if use_gate else None, R, Do, Dh)
        grp = self._expert_tiles
        w_moe = None
        for s, l in enumerate(c.stage_layers()):
            wgrad(ws["dH1"][s], ws["G"][s], p.grad("moe.attn0.weight"), db=p.grad("moe.attn0.bias"),
                  row_off=ws["row_off"], n_groups=E, stride_w=Dh * Do, stride_db=Dh, nsplit=4, M=R)
            if c.expert_fp8:      # dgrad on the TRANSPOSED e4m3 weights; their output-channel scales ride on the gradient rows
                self._fp8_gemm(ws["dH1"][s], Dh, None, p.s8("moe.attn0.weight"), p.q8t("moe.attn0.weight"), None, None, ws["dG"][s], Do, Dh, 2,
                               residual=ws["dG"][s], aux=ws["G"][s])
            else:
                ops.gemm_nt(ws["dH1"][s], p.w16t("moe.attn0.weight"), ws["dG"][s], residual=ws["dG"][s], aux=ws["G"][s],
                            stride_b=Dh * Do, epi=ops.EPI_MUL_DRELU, **grp(Dh))
            w_moe = wgrad(ws["dG"][s], ws[f"x{l}"], p.grad(f"moe.proj.{s}.weight"), db=p.grad(f"moe.proj.{s}.bias"),
                          x_rowmap=ws["rowmap"], row_off=ws["row_off"], n_groups=E, stride_w=Do * Dv, stride_db=Do,
                          nsplit=4, M=R)
            if c.expert_fp8:
                self._fp8_gemm(ws["dG"][s], Do, None, p.s8(f"moe.proj.{s}.weight"), p.q8t(f"moe.proj.{s}.weight"), None, None, ws["dF"][s], Dv, Do, 0)
            else:
                ops.gemm_nt(ws["dG"][s], p.w16t(f"moe.proj.{s}.weight"), ws["dF"][s], stride_b=Do * Dv, **grp(Do))
        # ---- router backward: CE on probabilities (medmoe_module.py:235-237) + gate gradients ----
        Hd = c.router_hidden
        ops.call("router_bwd", ws["probs"], ws["router_h"], p.f32("moe.router.2.weight"), ws["idx"],
                 ws["dgate"] if use_gate else None, lab32, dprobs_ext, c.w_cls * loss_scale / B, ws["dlogits"], ws["drouter_h"],
                 ws["loss_parts"], B, Hd, E, k)
        sg = lambda *a: ops.call("sgemm", *a)
        sg(ws["dlogits"], ws["router_h"], p.grad("moe.router.2.weight"), E, Hd, B, 1, E, Hd, 1, Hd, 1.0, 1.0)
        sg(ws["ones"], ws["dlogits"], p.grad("moe.router.2.bias"), 1, E, B, 0, 1, E, 1, E, 1.0, 1.0)
        sg(ws["drouter_h"], ws["router_in"], p.grad("moe.router.0.weight"), Hd, Dv, B, 1, Hd, Dv, 1, Dv, 1.0, 1.0)
        sg(ws["ones"], ws["drouter_h"], p.grad("moe.router.0.bias"), 1, Hd, B, 0, 1, Hd, 1, Hd, 1.0, 1.0)
        sg(ws["drouter_h"], p.f32("moe.router.0.weight"), ws["drouter_in"], B, Dv, Hd, Hd, 1, Dv, 1, Dv, 1.0, 0.0)
        return w_moe

    def _vit_backward(self, w_moe=None, bucket_ready=None, stage_grads: bool = True):
        """Final LayerNorm + the pre-norm blocks backward: ws["dln"] (gradient w.r.t. the final LayerNorm's output) -> ws["dxa"]
        (gradient w.r.t. ws["x0"]) and every block weight gradient; `stage_grads`: add the experts' stage-feature gradients ws["dF"]
        at the tapped layers.  Returns the event of the last wgrad on the second stream.  Call _wgrad_begin() first."""
        c, p, ws = self.cfg, self.params, self.ws
        B = self.B
        k, Dv, P, Nt, H = c.top_k, c.d_v, c.n_patch, c.n_tok_v, c.n_head_v
        wgrad, wait = self._wgrad, self._wait
        L = c.n_layer_v
        dx, dx2 = ws["dxa"], ws["dxb"]
        ops.layernorm_bwd(ws["dln"], ws[f"x{L}"], ws["stf"][0], ws["stf"][1], p.f32("vit.final_layer_norm.weight"), dx,
                          p.grad("vit.final_layer_norm.weight"), p.grad("vit.final_layer_norm.bias"))
        if bucket_ready is not None:
            wait(w_moe)                  # the experts' weight gradients ran on the second stream
            bucket_ready(L + 1)          # final LN + router + experts: complete
        stage_of = {l: s for s, l in enumerate(c.stage_layers())} if stage_grads else {}
        w_dz = w_dx2 = w_dqkv = None                       # last wgrad that READ the scratch buffer
        for l in range(L - 1, -1, -1):
            pre = f"vit.layer.{l}."
            if (l + 1) in stage_of:
                ops.call("stage_grad_add", ws["dF"][stage_of[l + 1]], ws["slot_of"], dx, B, k, P, Nt, Dv)
            st1, st2 = ws[f"st1_{l}"], ws[f"st2_{l}"]
            # FFN: x_out = h W2^T + b2 + xmid
            w_dx = wgrad(dx, ws[f"h{l}"], p.grad(pre + "feedforward.model.2.weight"), db=p.grad(pre + "feedforward.model.2.bias"))
            wait(w_dz)
            ops.gemm_nt(dx, p.w16t(pre + "feedforward.model.2.weight"), ws["dz"], aux=ws[f"z{l}"], epi=ops.EPI_MUL_AUX)
            w_dz = wgrad(ws["dz"], ws[f"ln2_{l}"], p.grad(pre + "feedforward.model.0.weight"), db=p.grad(pre + "feedforward.model.0.bias"))
            ops.gemm_nt(ws["dz"], p.w16t(pre + "feedforward.model.0.weight"), ws["dln"])
            wait(w_dx2)
            ops.layernorm_bwd(ws["dln"], ws[f"xmid{l}"], st2[0], st2[1], p.f32(pre + "feedforward_layernorm.weight"), dx2,
                              p.grad(pre + "feedforward_layernorm.weight"), p.grad(pre + "feedforward_layernorm.bias"), add=dx)
            # attention: xmid = att Wo^T + bo + x
            w_dx2 = wgrad(dx2, ws[f"att{l}"], p.grad(pre + "attention.output_proj.weight"), db=p.grad(pre + "attention.output_proj.bias"))
            ops.gemm_nt(dx2, p.w16t(pre + "attention.output_proj.weight"), ws["datt"])
            wait(w_dqkv)
            ops.attn_bwd(ws[f"qkv{l}"], ws[f"att{l}"], ws["datt"], ws[f"lse{l}"], None, ws["dqkv"], ws["delta"], B, Nt, H)
            w_dqkv = wgrad(ws["dqkv"], ws[f"ln1_{l}"], p.grad(pre + "attention.input_proj.weight"), db=p.grad(pre + "attention.input_proj.bias"))
            ops.gemm_nt(ws["dqkv"], p.w16t(pre + "attention.input_proj.weight"), ws["dln"])
            wait(w_dx)                                       # the FC2 wgrad read dx: done before LayerNorm-backward rewrites it
            ops.layernorm_bwd(ws["dln"], ws[f"x{l}"], st1[0], st1[1], p.f32(pre + "attention_layernorm.weight"), dx,
                              p.grad(pre + "attention_layernorm.weight"), p.grad(pre + "attention_layernorm.bias"), add=dx2)
            if bucket_ready is not None:
                wait(w_dqkv)                                 # the side stream runs in order: its last wgrad of the layer covers all four
                bucket_ready(l + 1)      # layer l: complete
        return w_dqkv

    def _side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    # ------------------------------------------------------------------------------------------
    def train_step(self, batch: Dict[str, torch.Tensor], optimizer: bool = True, zero_grad: bool = True, loss_scale: float = 1.0):
        """medmoe_module.py:284-316 model_step + backward + clip + Adam.  Returns device scalars.
        Gradient accumulation (accumulate_grad_batches of the trainer config): call with optimizer=False for all but the
        last micro-batch, zero_grad=False for all but the first, loss_scale = 1 / number of micro-batches; the reported
        losses are scaled the same way."""
        if self.use_graph and not self.dist and not self.train_text and batch["image"].is_cuda and ops.PROFILE is None:
            return self._train_step_graphed(batch, optimizer, zero_grad, loss_scale)
        B = batch["image"].shape[0]
        self._alloc(B)
        self.prefetch_cap_lens(batch["ids"])
        if zero_grad:
            self.params.zero_grad()
            if self.train_text:
                self.tstore.zero_grad()
        if self.overlap_wgrad and batch["image"].is_cuda and B * self.cfg.n_tok_v <= 131072:
            # the frozen text tower is independent of the image tower: at small per-rank batches its GEMMs (77 tokens per pair) fill
            # a fraction of the chip, so it runs on the second stream underneath the image tower
            main, side = torch.cuda.current_stream(), self._side_stream()
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                self.forward_text(batch["ids"], batch["attn_mask"], batch.get("token_type"))
                done = torch.cuda.Event(); done.record(side)
            self.forward_image(batch["image"])
            main.wait_event(done)
        else:
            self.forward_image(batch["image"])
            self.forward_text(batch["ids"], batch["attn_mask"], batch.get("token_type"))
        self.forward_backward_losses(batch["label"], loss_scale)
        if self.dist:
            from . import dist as D_
            if optimizer:
                red = D_.BucketedAllReduce(self.params.g32, self.bucket_bounds)
                self.backward(batch["label"], loss_scale, bucket_ready=red.ready)     # all-reduce overlapped with backward
                red.finish()
            else:
                self.backward(batch["label"], loss_scale)                             # accumulate locally, reduce with the last micro-batch
        else:
            self.backward(batch["label"], loss_scale)
        if self.train_text:
            self.backward_text(self._d_words, self.ws["d_txt_g"])
            if self.dist and optimizer:                           # the text tower's gradient: one more all-reduce (not overlapped)
                from . import dist as D_
                D_.allreduce_mean_(self.tstore.g32)
        if optimizer:
            if self.train_text:                                   # ONE clip norm over both towers' gradients, as clip_grad_norm_ over all parameters
                self.params.adam_step(extra_normsq=self.tstore.sumsq())
                self.tstore.adam_step(self.params.normsq)
            else:
                self.params.adam_step()
        lp = self.ws["loss_parts"]
        c = self.cfg
        # loss_parts hold the WEIGHTED global/local parts; report the reference's unweighted names too
        # (the router CE kernel reports the plain mean: scale it here so that every reported loss follows loss_scale)
        cls = lp[0] * loss_scale
        return {"loss": c.w_cls * cls + lp[2] + lp[3], "classifier_loss": cls, "classifier_acc": lp[1],
                "g_loss": lp[2] / c.w_global, "l_loss": lp[3] / c.w_local}

    def _forward_both(self, b):
        if self.overlap_wgrad and b["image"].is_cuda and self.B * self.cfg.n_tok_v <= 131072:
            main, side = torch.cuda.current_stream(), self._side_stream()
            ev = torch.cuda.Event(); ev.record(main); side.wait_event(ev)
            with torch.cuda.stream(side):
                self.forward_text(b["ids"], b["attn_mask"], b.get("token_type"))
                done = torch.cuda.Event(); done.record(side)
            self.forward_image(b["image"])
            main.wait_event(done)
        else:
            self.forward_image(b["image"])
            self.forward_text(b["ids"], b["attn_mask"], b.get("token_type"))

    def _train_step_graphed(self, batch, optimizer, zero_grad, loss_scale):
        """train_step with the forward and the backward replayed from hipGraphs (MEDMOE_GRAPH=1).  The first two steps of a (batch size,
        loss_scale, zero_grad) combination run eagerly (every buffer gets allocated, the library's lazy state settles), the third is captured."""
        B = batch["image"].shape[0]
        self._alloc(B)
        key = (B, float(loss_scale), bool(zero_grad), tuple(sorted(batch.keys())))
        st = self._graph
        if st is None or st["key"] != key:
            st = self._graph = {"key": key, "in": {k: torch.empty_like(v) for k, v in batch.items()}, "fwd": None, "bwd": None, "warm": 0}
        b = st["in"]
        for k, v in batch.items():
            b[k].copy_(v)
        self.prefetch_cap_lens(b["ids"])                          # eager: its copy to the host and the event the losses wait on

        def fwd():
            if zero_grad:
                self.params.zero_grad()
            self._forward_both(b)

        if st["warm"] < 2:
            st["warm"] += 1
            fwd()
            self.forward_backward_losses(b["label"], loss_scale)
            self.backward(b["label"], loss_scale)
        else:
            if st["fwd"] is None:
                torch.cuda.synchronize()
                st["fwd"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st["fwd"]):
                    fwd()
            st["fwd"].replay()
            self._seg = None                                      # forward_text consumed it at capture time
            self.forward_backward_losses(b["label"], loss_scale)
            if st["bwd"] is None:
                torch.cuda.synchronize()
                st["bwd"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st["bwd"]):
                    self.backward(b["label"], loss_scale)
            st["bwd"].replay()
        if optimizer:
            self.params.adam_step()
        lp, c = self.ws["loss_parts"], self.cfg
        cls = lp[0] * loss_scale
        return {"loss": c.w_cls * cls + lp[2] + lp[3], "classifier_loss": cls, "classifier_acc": lp[1],
                "g_loss": lp[2] / c.w_global, "l_loss": lp[3] / c.w_local}

    # reference-layout views (med_moe.py:102-108)
    def outputs(self):
        c, ws = self.cfg, self.ws
        B, P = self.B, c.n_patch
        Hh = int(P ** 0.5)
        return {"img_g": ws["img_g"], "img_l": ws["img_l"].float().transpose(1, 2).reshape(B, c.d_out, Hh, Hh),
                "txt_g": ws["txt_g"], "txt_l": ws["words32"].transpose(1, 2), "probs": ws["probs"], "idx": ws["idx"],
                "cap_lens": self.cap_lens}
