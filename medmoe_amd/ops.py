"""Thin launch wrappers: torch tensors in, C-ABI calls on torch's CURRENT HIP stream out.

Every function checks dtypes/shapes on the host before the launch (a wrong shape must never
reach a hand-written kernel) and raises on a non-zero return code.  Nothing here computes.
"""
import ctypes
from typing import Optional

import torch

from ._lib import load_library

_c = ctypes
_vp = _c.c_void_p

EPI_NONE, EPI_GELU, EPI_RELU, EPI_MUL_DGELU, EPI_MUL_DRELU, EPI_GELU_DAUX, EPI_MUL_AUX = range(7)

# bench.py sets this to a list to time every launch with HIP events on the launch stream: entries are
# (kernel label, work, "flop" | "byte" | None, start event, end event, detail)
PROFILE = None
_NT_KERNELS = {0: "gemm_nt_kernel", 1: "gemm_nt256_kernel", 2: "gemm_nt512_kernel", 3: "gemm_nt512_kernel<GROUPED>", 4: "gemm_nt4w_kernel",
               5: "gemm_nt4w_kernel<GROUPED>", 6: "gemm_nt_direct_kernel"}


class _Timed:
    """`with _Timed(label, work, unit):` around one launch - HIP events on the current stream when PROFILE is a list, nothing otherwise.
    `label` may be a callable evaluated after the launch (the NT GEMM's kernel is chosen inside the library)."""
    __slots__ = ("label", "work", "unit", "detail", "ev0")

    def __init__(self, label, work=None, unit=None, detail=None):
        self.label, self.work, self.unit, self.detail, self.ev0 = label, work, unit, detail, None

    def __enter__(self):
        if PROFILE is not None:
            self.ev0 = torch.cuda.Event(enable_timing=True)
            self.ev0.record()
        return self

    def __exit__(self, et, ev, tb):
        if self.ev0 is not None and et is None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            PROFILE.append((self.label() if callable(self.label) else self.label, self.work, self.unit, self.ev0, ev1, self.detail))
        return False


def _nt_label():
    return _NT_KERNELS.get(load_library().medmoe_last_gemm_nt_kernel(), "gemm_nt?")


def _ptr(t: Optional[torch.Tensor]):
    return _vp(0) if t is None else _vp(t.data_ptr())


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _stream_handle() -> int:
    """torch's current HIP stream of the current device as a raw handle (torch.cuda.current_stream().cuda_stream is the same value through
    several microseconds of Python per call - milliseconds per step at ~600 launches, tools/host_profile_swin.py)."""
    return _raw_stream(_cur_device())


def _stream():
    return _vp(_stream_handle())


def _chk(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"medmoe_{name} failed with code {rc} (-1 bad argument, -2 bad shape, -3 launch error)")


def _require_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: medmoe_amd kernels only run on the GPU (no CPU fallback)")


def _need(t: torch.Tensor, dtype, name: str, contiguous_last=True):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    _require_gpu(t, name)
    if contiguous_last and t.stride(-1) != 1:
        raise ValueError(f"{name}: last dimension must be contiguous")


def gemm_nt(a, b, out, *, bias=None, residual=None, aux=None, a_rowmap=None, c_rowmap=None,
            tiles=None, tile_count=None, max_tiles=0, stride_b=0, stride_bias=0, alpha=1.0,
            epi=EPI_NONE, M=None, N=None, col_perm=False, tile_rows=128):
    """out[M,N] = epi(alpha * a[M,K] @ b[N,K]^T (+bias)) (+residual).  a, b bf16; out bf16/f32."""
    lib = load_library()
    _need(a, torch.bfloat16, "a"); _need(b, torch.bfloat16, "b")
    out_f32 = out.dtype == torch.float32
    if not out_f32:
        _need(out, torch.bfloat16, "out")
    K = a.shape[-1]
    if N is None:
        N = b.shape[-2]
    if b.shape[-1] != K:
        raise ValueError("gemm_nt: K mismatch")
    if M is None:
        M = out.shape[0]
    if (out.shape[-1] != N and not col_perm and c_rowmap is None) or out.shape[-1] < N:
        raise ValueError("gemm_nt: N mismatch")
    if a_rowmap is None and a.shape[0] < M:
        raise ValueError("gemm_nt: a has fewer rows than M")
    if bias is not None:
        _need(bias, torch.float32, "bias")
    if residual is not None:
        _need(residual, torch.bfloat16, "residual")
    if aux is not None:
        _need(aux, torch.bfloat16, "aux")
    for t, nm in ((a_rowmap, "a_rowmap"), (c_rowmap, "c_rowmap"), (tiles, "tiles"), (tile_count, "tile_count")):
        if t is not None:
            _need(t, torch.int32, nm)
    fn = _nt_fn(tile_rows == 256)
    dp = lambda t: None if t is None else t.data_ptr()
    cargs = (a.data_ptr(), a.stride(-2), b.data_ptr(), b.stride(-2), out.data_ptr(), out.stride(-2), M, N, K, dp(bias), dp(residual),
             residual.stride(-2) if residual is not None else 0, dp(aux), aux.stride(-2) if aux is not None else 0, dp(a_rowmap), dp(c_rowmap),
             dp(tiles), dp(tile_count), max_tiles, stride_b, stride_bias, alpha, epi, 1 if out_f32 else 0, 1 if col_perm else 0)
    if PROFILE is None:
        rc = fn(*cargs, _stream_handle())
        if rc != 0:
            _chk(rc, "gemm_nt")
        return out
    nbytes = 2.0 * (M * K + N * K) + (4.0 if out_f32 else 2.0) * M * N * (1 + (aux is not None)) + 2.0 * M * N * (residual is not None)
    with _Timed(_nt_label, 2.0 * M * N * K, "flop", ("nt", M, N, K, epi, nbytes)):
        _chk(fn(*cargs, _stream_handle()), "gemm_nt")
    return out


_NT_FN = {}


def _nt_fn(tiles256: bool):
    f = _NT_FN.get(tiles256)
    if f is None:
        lib = load_library()
        f = lib.medmoe_gemm_nt_tiles256 if tiles256 else lib.medmoe_gemm_nt
        I, L, P = _c.c_int, _c.c_longlong, _vp
        f.argtypes = [P, I, P, I, P, I, I, I, I, P, P, I, P, I, P, P, P, P, I, L, L, _c.c_float, I, I, I, P]
        f.restype = I
        _NT_FN[tiles256] = f
    return f


def gemm_nt_rows(a, b, out, m_dev, *, bias=None, residual=None, aux=None, alpha=1.0, epi=EPI_NONE):
    """gemm_nt on the first *m_dev rows of `a` (m_dev: int32 device tensor of one element, 1 <= value <= a.shape[0]): packed variable-length
    batches without a device-to-host copy of the count.  Rows past the count are neither read for results nor written."""
    lib = load_library()
    _need(a, torch.bfloat16, "a"); _need(b, torch.bfloat16, "b"); _need(m_dev, torch.int32, "m_dev")
    out_f32 = out.dtype == torch.float32
    if not out_f32:
        _need(out, torch.bfloat16, "out")
    M, K, N = a.shape[0], a.shape[-1], b.shape[-2]
    if b.shape[-1] != K or out.shape[-1] != N or out.shape[0] < M:
        raise ValueError("gemm_nt_rows: shape mismatch")
    if bias is not None:
        _need(bias, torch.float32, "bias")
    # the row count lives on the device: the work is filled in by the caller's `rows_hint` (bench.py: the batch's token count) or left open
    with _Timed(_nt_label, 2.0 * ROWS_HINT * N * K if ROWS_HINT else None, "flop" if ROWS_HINT else None, ("nt_rows", M, N, K, epi)):
        rc = lib.medmoe_gemm_nt_rows(_ptr(a), _c.c_int(a.stride(-2)), _ptr(b), _c.c_int(b.stride(-2)), _ptr(out), _c.c_int(out.stride(-2)),
                                     _c.c_int(M), _c.c_int(N), _c.c_int(K), _ptr(bias), _ptr(residual),
                                     _c.c_int(residual.stride(-2) if residual is not None else 0), _ptr(aux),
                                     _c.c_int(aux.stride(-2) if aux is not None else 0), _c.c_float(alpha), _c.c_int(epi),
                                     _c.c_int(1 if out_f32 else 0), _ptr(m_dev), _stream())
        _chk(rc, "gemm_nt_rows")
    return out


ROWS_HINT = 0       # bench.py: the packed text tower's row count (known on the host there), so that gemm_nt_rows launches carry their flops


def current_stream_handle() -> int:
    """Raw handle of torch's current HIP stream on the current device."""
    return _stream_handle()


def stream_fork(src: int, dst: int):
    """Stream `dst` waits for everything enqueued on `src` so far (raw handles: torch.cuda.Stream.cuda_stream / current_stream_handle())."""
    f = _FN.get("stream_fork")
    if f is None:
        f = load_library().medmoe_stream_fork
        f.argtypes = [_vp, _vp]
        f.restype = _c.c_int
        _FN["stream_fork"] = f
    rc = f(src, dst)
    if rc != 0:
        _chk(rc, "stream_fork")


def set_option(key: int, value: int):
    """medmoe_set_option: kernel-selection switches (1 nt256, 2 nt512, 3 tn512, 4 grouped-wgrad rows, 5 max NT grid,
    6 scores512, 7 gemm_nt4w, 8 gemm_tn4w, 9 plain-wgrad rows per range) - for tests and measurements; the defaults are the fastest measured."""
    _chk(load_library().medmoe_set_option(_c.c_int(key), _c.c_int(value)), "set_option")


def gemm_tn(g, x, dw, *, db=None, x_rowmap=None, g_rowmap=None, row_off=None, n_groups=1,
            stride_w=0, stride_db=0, nsplit=16, M=None, stream=None, scratch=None):
    """dw[g][Nn,Kk] += g[M,Nn]^T @ x[M,Kk]; db[g][Nn] += colsum(g).  fp32 atomic accumulation.  stream: a raw stream handle to launch on
    instead of torch's current stream (the caller orders it: stream_fork).  scratch (fp32, plain operands only): medmoe_gemm_tn_staged -
    partial tiles stored there and summed by a second kernel instead of atomics on dw, when the shape allows; the caller must not hand the
    same scratch to launches that can overlap."""
    lib = load_library()
    _need(g, torch.bfloat16, "g"); _need(x, torch.bfloat16, "x"); _need(dw, torch.float32, "dw")
    if M is None:
        M = g.shape[0]
    Nn, Kk = g.shape[-1], x.shape[-1]
    if dw.shape[-2] != Nn or dw.shape[-1] != Kk:
        raise ValueError("gemm_tn: dw shape mismatch")
    if db is not None:
        _need(db, torch.float32, "db")
    for t, nm in ((x_rowmap, "x_rowmap"), (g_rowmap, "g_rowmap"), (row_off, "row_off")):
        if t is not None:
            _need(t, torch.int32, nm)
    if scratch is not None and x_rowmap is None and g_rowmap is None and row_off is None and n_groups == 1:
        _need(scratch, torch.float32, "scratch")
        fs = _TN_FN.get(1)
        if fs is None:
            fs = lib.medmoe_gemm_tn_staged
            I, L, P = _c.c_int, _c.c_longlong, _vp
            fs.argtypes = [P, I, P, I, P, I, P, I, I, I, P, L, P]
            fs.restype = I
            _TN_FN[1] = fs
        sargs = (g.data_ptr(), g.stride(-2), x.data_ptr(), x.stride(-2), dw.data_ptr(), dw.stride(-2), None if db is None else db.data_ptr(),
                 M, Nn, Kk, scratch.data_ptr(), scratch.numel())
        if PROFILE is None or stream is not None:
            rc = fs(*sargs, _stream_handle() if stream is None else stream)
            if rc != 0:
                _chk(rc, "gemm_tn_staged")
            return dw
        with _Timed("gemm_tn (wgrad: gemm_tn4w_kernel / gemm_tn512_kernel / gemm_tn_kernel)", 2.0 * M * Nn * Kk, "flop", ("tn", M, Nn, Kk, n_groups)):
            _chk(fs(*sargs, _stream_handle()), "gemm_tn_staged")
        return dw
    fn = _TN_FN.get(0)
    if fn is None:
        fn = lib.medmoe_gemm_tn
        I, L, P = _c.c_int, _c.c_longlong, _vp
        fn.argtypes = [P, I, P, I, P, I, P, I, I, I, P, P, P, I, L, L, I, P]
        fn.restype = I
        _TN_FN[0] = fn
    dp = lambda t: None if t is None else t.data_ptr()
    cargs = (g.data_ptr(), g.stride(-2), x.data_ptr(), x.stride(-2), dw.data_ptr(), dw.stride(-2), dp(db), M, Nn, Kk, dp(x_rowmap), dp(g_rowmap),
             dp(row_off), n_groups, stride_w, stride_db, nsplit)
    if PROFILE is None or stream is not None:
        rc = fn(*cargs, _stream_handle() if stream is None else stream)
        if rc != 0:
            _chk(rc, "gemm_tn")
        return dw
    with _Timed("gemm_tn (wgrad: gemm_tn4w_kernel / gemm_tn512_kernel / gemm_tn_kernel)", 2.0 * M * Nn * Kk, "flop", ("tn", M, Nn, Kk, n_groups)):
        _chk(fn(*cargs, _stream_handle()), "gemm_tn")
    return dw


_TN_FN = {}


def layernorm_fwd(x, gamma, beta, y, mean, rstd, eps):
    lib = load_library()
    _need(x, torch.bfloat16, "x"); _need(gamma, torch.float32, "gamma"); _need(beta, torch.float32, "beta")
    rows, D = x.numel() // x.shape[-1], x.shape[-1]
    if not x.is_contiguous() or not y.is_contiguous() or y.numel() != x.numel():
        raise ValueError("layernorm_fwd: x/y must be contiguous and equally sized")
    with _Timed("layernorm_fwd_kernel", 4.0 * rows * D, "byte"):
        rc = lib.medmoe_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd),
                                      _c.c_int(rows), _c.c_int(D), _c.c_float(eps),
                                      _c.c_int(1 if y.dtype == torch.float32 else 0), _stream())
        _chk(rc, "layernorm_fwd")
    return y


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dgamma=None, dbeta=None, add=None):
    lib = load_library()
    for t, nm in ((dy, "dy"), (x, "x"), (dx, "dx")):
        _need(t, torch.bfloat16, nm)
        if not t.is_contiguous():
            raise ValueError(f"layernorm_bwd: {nm} must be contiguous")
    rows, D = x.numel() // x.shape[-1], x.shape[-1]
    with _Timed("layernorm_bwd_kernel", (8.0 if add is not None else 6.0) * rows * D, "byte"):
        rc = lib.medmoe_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(add), _ptr(dx),
                                      _ptr(dgamma), _ptr(dbeta), _c.c_int(rows), _c.c_int(D), _stream())
        _chk(rc, "layernorm_bwd")
    return dx


def attn_fwd(qkv, out, lse, key_mask, B, N, H):
    lib = load_library()
    _need(qkv, torch.bfloat16, "qkv"); _need(out, torch.bfloat16, "out"); _need(lse, torch.float32, "lse")
    D = H * 64
    if qkv.numel() != B * N * 3 * D or out.numel() != B * N * D or lse.numel() != B * H * N:
        raise ValueError("attn_fwd: buffer sizes do not match (B,N,H)")
    if key_mask is not None:
        _need(key_mask, torch.uint8, "key_mask")
        if key_mask.numel() != B * N:
            raise ValueError("attn_fwd: key_mask must be [B,N]")
    with _Timed("attn_fwd_kernel", 4.0 * N * N * 64 * B * H, "flop"):
        rc = lib.medmoe_attn_fwd(_ptr(qkv), _ptr(out), _ptr(lse), _ptr(key_mask), _c.c_int(B), _c.c_int(N),
                                 _c.c_int(H), _c.c_int(64), _stream())
        _chk(rc, "attn_fwd")
    return out


def attn_bwd(qkv, out, dout, lse, key_mask, dqkv, delta, B, N, H):
    lib = load_library()
    for t, nm in ((qkv, "qkv"), (out, "out"), (dout, "dout"), (dqkv, "dqkv")):
        _need(t, torch.bfloat16, nm)
    D = H * 64
    if qkv.numel() != B * N * 3 * D or dqkv.numel() != qkv.numel() or dout.numel() != B * N * D \
            or delta.numel() != B * H * N or lse.numel() != B * H * N:
        raise ValueError("attn_bwd: buffer sizes do not match (B,N,H)")
    with _Timed("attn_bwd (attn_bwd_dq_kernel + attn_bwd_dkv_kernel)", 10.0 * N * N * 64 * B * H, "flop"):
        rc = lib.medmoe_attn_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(key_mask), _ptr(dqkv), _ptr(delta),
                                 _c.c_int(B), _c.c_int(N), _c.c_int(H), _c.c_int(64), _stream())
        _chk(rc, "attn_bwd")
    return dqkv


# ---------------------------------------------------------------------------------------------
# generic caller for the remaining entry points: sig chars  p=pointer(tensor|None) i=int l=int64 f=float d=double
# ---------------------------------------------------------------------------------------------
_SIGS = {
    "patchify": "ppiiiiii", "patchify_ld": "ppiiiiiii", "init_tokens": "pppiii", "pos_cls_grad": "pppiii",
    "text_embed_ln": "ppppppppiiiif", "text_aggregate": "ppppippppiii",
    "mean_tokens": "ppiiiii", "broadcast_tokens": "ppiiiiif",
    "router_fwd": "pppppppppiiiii", "router_bwd": "pppppppfpppiiii",
    "sgemm": "pppiiilllllff", "dispatch": "piiiiippppppip",
    "scale_attn_fwd": "pppppippiii", "combine_fwd": "ppppiiii",
    "scale_attn_bwd": "ppppppppppiipppppiii", "stage_grad_add": "pppiiiii",
    "ce_strided": "ppiilliffip", "soft_xent_strided": "pppiillffffip", "hardneg_strided": "ppiillffip", "rownorm": "ppii", "cos_scale": "pppiif", "cos_scale_bwd": "ppppppiif",
    "add_rowscaled": "pppii", "words_prep": "pppiiii", "unpad_cast": "ppiiii",
    "local_pair": "pppppppppppppiiiiifffi", "local_scores": "pppppiiiii", "local_pair2": "ppppppppppiiiifff", "scale_blocks": "pppiiii",
    "words_prep_ragged": "pppiiiippl", "local_scores_ragged": "pppppiiiiipiill",
    "local_pair2_ragged": "pppppppppiiiifffpiill", "scale_blocks_ragged": "pppiiipl",
    "local_pair3": "ppppppppppppliiiifffpiilllip", "local_pair3_wgrad": "ppppppppppppliiiifffpiilllipp", "local_scores_t": "pppppiiiiipiilll", "gemm_tn_cols": "pipipiiiiilllil", "gemm_tn_gram": "piplipiiiill",
    "local_gen_fwd_a": "pppiiiiiifl", "local_gen_cos": "pppppppiiiiiffl", "local_gen_dwctx": "ppppppppiiiiiffl",
    "local_gen_bwd_s": "ppppiiiiiifl", "unpad_cast2": "pppiiii",
    "quant_rows_e4m3": "pipppippii", "quant_weights_e4m3": "ppppiii", "gemm_fp8_grouped": "ppppppippppiiillli",
    "lerp_tokens_fwd": "ppiiii", "lerp_tokens_bwd": "pppiiii", "lerp_tokens_bwd2": "ppppiiii",
    "text_pack": "pppppii", "segment_map": "pippppiiii", "text_aggregate_bwd": "ppppiii", "text_embed_ln_bwd": "pppppppppppiiiif", "text_embed_ln_packed": "ppppppppiiiifpp", "text_aggregate_packed": "ppppipppppiii",
    "layernorm_fwd_rows": "ppppppiifip", "attn_fwd_varlen": "ppppiiii",
    "win_attn_fwd": "ppppiiiiii", "win_attn_bwd": "ppppppiiiiii", "patch_merge": "ppiiiii", "drop_path": "ppppil", "patchify_ld": "ppiiiiiii",
    "sumsq": "plp", "sumsq_det": "plpp", "adam_step": "pppppldddddipff", "cast_bf16": "ppl", "transpose_many": "pppii",
}


_CTYPES = {"p": _vp, "i": _c.c_int, "l": _c.c_longlong, "d": _c.c_double, "f": _c.c_float}
_FN = {}


def _fn(name: str):
    """The library entry with its argument types declared once: ctypes then converts plain Python ints / floats itself."""
    f = _FN.get(name)
    if f is None:
        f = getattr(load_library(), "medmoe_" + name)
        f.argtypes = [_CTYPES[ch] for ch in _SIGS[name]] + [_vp]
        f.restype = _c.c_int
        _FN[name] = f
    return f


def call(name: str, *args):
    """Launch medmoe_<name> on the current stream.  Tensors must already be validated by the caller."""
    sig = _SIGS[name]
    if len(args) != len(sig):
        raise TypeError(f"medmoe_{name}: expected {len(sig)} arguments, got {len(args)}")
    cargs = []
    for ch, a in zip(sig, args):
        if ch == "p":
            if a is None:
                cargs.append(None)
            else:
                if not a.is_cuda:
                    _require_gpu(a, "medmoe_" + name)
                cargs.append(a.data_ptr())
        elif ch == "i" or ch == "l":
            cargs.append(int(a))
        else:
            cargs.append(float(a))
    if PROFILE is None:
        rc = _fn(name)(*cargs, _stream_handle())
        if rc != 0:
            _chk(rc, name)
        return
    cost = _COSTS.get(name)
    label, work, unit = cost(args) if cost is not None else (name + "_kernel", None, None)
    with _Timed(label, work, unit):
        _chk(_fn(name)(*cargs, _stream_handle()), name)


def _cost_scores(a):        # (ctx, words, cap_lens, X, lse, B, Bc, P, T, Do, members, n_c, ntt, cbase, ld, bs): B*P region rows x n_c captions of 16*ntt words
    return "scores512_kernel<NTT, true> (score GEMM + word softmax)", 2.0 * a[5] * a[7] * a[11] * 16 * a[12] * a[9], "flop"


def _cost_pair3(a):         # (X, dS, AT, UT, lse, gm, wn, caps, gsim, sim, att, stats, srows, B, Bc, P, T, t1, t2, eps, members, n_c, ntt, cbase, ld, bs, HWq, d2)
    elems = float(a[13]) * a[26] * a[21] * 16 * a[22]           # (image, region column, caption word row) of the class
    if a[1] is None:
        return "local_pair3_kernel<196, NTT, false> (forward)", 4.0 * elems, "byte"      # reads log-probabilities, writes A
    return "local_pair3_kernel<196, NTT, true> (backward)", 6.0 * elems, "byte"          # reads lp + A, writes dS


def _cost_tn_cols(a):       # (g, ldg, x, ldx, dw, ldw, M, Nn, Kk, n_groups, ...)
    return "gemm_tn4w_kernel<false, COLG, false> (local-loss dC)", 2.0 * a[6] * a[7] * a[8] * max(1, a[9]), "flop"


def _cost_tn_gram(a):       # (AT, ld, d2, srows, 1, out, ldo, Kp, HWq, B, bs, ostride)
    return "gemm_tn4w_kernel<false, COLG, SCALE> (weighted Gram)", 2.0 * a[7] * a[8] * a[8] * a[9], "flop"


_COSTS = {
    "local_scores_t": _cost_scores, "local_pair3": _cost_pair3, "local_pair3_wgrad": _cost_pair3, "gemm_tn_cols": _cost_tn_cols, "gemm_tn_gram": _cost_tn_gram,
    "adam_step": lambda a: ("adam_kernel", 34.0 * a[5], "byte"),                                   # p, g, m, v read; p, m, v, bf16 copy written
    "scale_attn_bwd": lambda a: ("scale_attn_bwd_kernel", 2.0 * a[17] * (4 * (2 * a[18] + 2 * a[19]) + 2 * a[18]), "byte"),    # G, dG, H1, dH1 x 4 scales + eout, d_img_l rows
    "scale_attn_fwd": lambda a: ("scale_attn_fwd_kernel", 2.0 * a[8] * (4 * (a[9] + a[10]) + a[9]), "byte"),
    "layernorm_fwd_rows": lambda a: ("layernorm_fwd_kernel", 4.0 * (ROWS_HINT or a[6]) * a[7], "byte"),
}


def local_fast_path(HW: int, T: int) -> bool:
    return bool(load_library().medmoe_local_fast_path(_c.c_int(HW), _c.c_int(T)))


def local_pair3_chunks(n: int):
    _chk(load_library().medmoe_local_pair3_chunks(_c.c_int(n)), "local_pair3_chunks")


def local_pair3_supported(HW: int, T: int) -> bool:
    return bool(load_library().medmoe_local_pair3_supported(_c.c_int(HW), _c.c_int(T)))


def local_geometry(HW: int, T: int):
    lib = load_library()
    a, b, c = _c.c_int(0), _c.c_int(0), _c.c_int(0)
    rc = lib.medmoe_local_geometry(_c.c_int(HW), _c.c_int(T), _c.byref(a), _c.byref(b), _c.byref(c))
    _chk(rc, "local_geometry")
    return a.value, b.value, c.value
