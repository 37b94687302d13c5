// Swin-T tower pieces that the ViT path does not have (the reference's image encoder is HF SwinModel
// 'microsoft/swin-tiny-patch4-window7-224', reference swin.py:119-149 - transformers' modeling_swin.py is the algorithm):
//   win_attn_fwd / win_attn_bwd : (shifted-)window multi-head attention, 7x7 windows = 49 tokens padded to 64, head dim 32,
//                                 relative position bias + shift mask, on token-major qkv [B*H*W, 3C] - the cyclic shift, the window
//                                 partition and their inverses are row arithmetic inside the kernel (modeling_swin.py window_partition /
//                                 torch.roll / get_attn_mask), nothing is copied
//   patch_merge_gather / _scatter: [B, H, W, C] -> [B, H/2, W/2, 4C] in the (0,0), (1,0), (0,1), (1,1) order of SwinPatchMerging
// One wave owns one (image, window, head): Q, K rows are 16-byte loads straight into MFMA fragments (head dim 32 = one k-step of
// v_mfma_f32_16x16x32_bf16); S^T = K Q^T puts a query on a lane and its 64 keys in 16 accumulator registers + 4 lane groups, so the
// softmax is in-lane + two shuffles, and P^T feeds O^T = V^T P^T as the B operand without leaving the registers (accumulator-as-
// operand with a permuted key order); V^T (keys along k) is gathered from a 4-KB LDS image of V.
#include "common.h"

struct WinArgs {
  const bf16_t* qkv; const bf16_t* dout; bf16_t* out; bf16_t* dqkv;
  const float* bias; float* dbias; float* lse;
  int B, H, W, C, heads, shift, n_units;
  float scale;
};

__device__ __forceinline__ int div7(int n) { return (n * 37) >> 8; }      // exact for 0 <= n < 64

// window-local token n (0..48; clamped) of window (wy, wx) -> row of the token-major activation, and its shift-mask region id
__device__ __forceinline__ void win_token(const WinArgs& p, int b, int wy, int wx, int n, int& row, int& id) {
  n = min(n, 48);
  const int iy = div7(n), ix = n - 7 * iy;
  const int Y = wy * 7 + iy, X = wx * 7 + ix;
  int y = Y + p.shift, x = X + p.shift;
  if (y >= p.H) y -= p.H;
  if (x >= p.W) x -= p.W;
  row = (b * p.H + y) * p.W + x;
  const int ry = (Y < p.H - 7) ? 0 : (Y < p.H - p.shift ? 1 : 2), rx = (X < p.W - 7) ? 0 : (X < p.W - p.shift ? 1 : 2);
  id = ry * 3 + rx;
}

// 8 bf16 from the four accumulator registers of two key tiles -> the B fragment of k-step jj: element e <-> key 32 jj + 16 (e >> 2) + 4 g + (e & 3)
__device__ __forceinline__ bf16x8_t pack_acc2(const f32x4_t& a, const f32x4_t& b) {
  const uint4 v = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3]));
  return __builtin_bit_cast(bf16x8_t, v);
}
// A fragment of M^T for k-step jj from an LDS image M[64][32] (bf16): lane (fr, g) of row tile dt holds M[key(g, e)][16 dt + fr] in the key order above
__device__ __forceinline__ bf16x8_t gather_t(const bf16_t* img, int dt, int jj, int fr, int g) {
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = 2 * q, e1 = 2 * q + 1;
    const int k0 = 32 * jj + 16 * (e0 >> 2) + 4 * g + (e0 & 3), k1 = 32 * jj + 16 * (e1 >> 2) + 4 * g + (e1 & 3);
    w[q] = (uint32_t)img[k0 * 32 + 16 * dt + fr] | ((uint32_t)img[k1 * 32 + 16 * dt + fr] << 16);
  }
  return __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3]));
}

struct WinUnit { int b, wy, wx, h; };
__device__ __forceinline__ WinUnit win_decode(const WinArgs& p, int u) {
  WinUnit w;
  const int nwx = p.W / 7, nwy = p.H / 7;
  w.h = u % p.heads; u /= p.heads;
  w.wx = u % nwx; u /= nwx;
  w.wy = u % nwy; w.b = u / nwy;
  return w;
}

// S^T tiles [key tile j][query tile i] -> scores with scale, bias and shift mask (in place)
template <bool SHIFT>
__device__ __forceinline__ void win_scores(const WinArgs& p, const WinUnit& un, f32x4_t (&acc)[4][4], const int (&qid)[4], int fr, int g) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int kid[4] = {0, 0, 0, 0};
    if (SHIFT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { int row; win_token(p, un.b, un.wy, un.wx, 16 * j + 4 * g + r, row, kid[r]); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 bv = *(const float4*)(p.bias + (((long long)un.h * 64 + 16 * i + fr) * 64 + 16 * j + 4 * g));
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = acc[j][i][r] * p.scale + bb[r];
        if (SHIFT && kid[r] != qid[i]) s -= 100.f;
        acc[j][i][r] = s;
      }
    }
  }
}

template <bool SHIFT>
__global__ __launch_bounds__(256) void win_attn_fwd_kernel(WinArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t vsm[4][64 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int u = blockIdx.x * 4 + wave;
  if (u >= p.n_units) return;
  const WinUnit un = win_decode(p, u);
  const int C3 = 3 * p.C;
  int row[4], qid[4];
  bf16x8_t qf[4], kf[4];
  bf16_t* vimg = vsm[wave];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    win_token(p, un.b, un.wy, un.wx, 16 * t + fr, row[t], qid[t]);
    const bool ok = 16 * t + fr < 49;
    const bf16_t* src = p.qkv + (long long)row[t] * C3 + un.h * 32 + 8 * g;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    qf[t] = __builtin_bit_cast(bf16x8_t, ok ? *(const uint4*)src : z);
    kf[t] = __builtin_bit_cast(bf16x8_t, ok ? *(const uint4*)(src + p.C) : z);
    *(uint4*)(vimg + (16 * t + fr) * 32 + 8 * g) = ok ? *(const uint4*)(src + 2 * p.C) : z;
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], qf[i], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  win_scores<SHIFT>(p, un, acc, qid, fr, g);
  // softmax over the keys of each query column (16 values in the lane x 4 lane groups)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[j][i][r]);
    m = fmaxf(m, __shfl_xor(m, 16, 64)); m = fmaxf(m, __shfl_xor(m, 32, 64));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = __expf(acc[j][i][r] - m); acc[j][i][r] = e; s += e; }
    s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
    const float inv = 1.f / s;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][i][r] *= inv;
    if (g == 0 && p.lse) p.lse[(long long)u * 64 + 16 * i + fr] = m + __logf(s);
  }
  // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const bf16x8_t v0 = gather_t(vimg, dt, 0, fr, g), v1 = gather_t(vimg, dt, 1, fr, g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4_t o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, pack_acc2(acc[0][i], acc[1][i]), (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, pack_acc2(acc[2][i], acc[3][i]), o, 0, 0, 0);
      if (16 * i + fr < 49)
        *(uint2*)(p.out + (long long)row[i] * p.C + un.h * 32 + 16 * dt + 4 * g) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
    }
  }
}

// Backward of one (window, head): recomputes the probabilities in BOTH orientations (a query on a lane for the row sums and dQ, a key
// on a lane for dK and dV: each then feeds its product as an accumulator-held B operand); K^T, Q^T, dO^T come from LDS images.
template <bool SHIFT>
__global__ __launch_bounds__(256) void win_attn_bwd_kernel(WinArgs p) {
  __shared__ __attribute__((aligned(16))) bf16_t sm[4][3][64 * 32];      // per wave: K, Q, dO images [token][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, g = lane >> 4;
  const int u = blockIdx.x * 4 + wave;
  if (u >= p.n_units) return;
  const WinUnit un = win_decode(p, u);
  const int C3 = 3 * p.C;
  int row[4], qid[4];
  bf16x8_t qf[4], kf[4], vf[4], gf[4];
  bf16_t* kimg = sm[wave][0]; bf16_t* qimg = sm[wave][1]; bf16_t* gimg = sm[wave][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    win_token(p, un.b, un.wy, un.wx, 16 * t + fr, row[t], qid[t]);
    const bool ok = 16 * t + fr < 49;
    const bf16_t* src = p.qkv + (long long)row[t] * C3 + un.h * 32 + 8 * g;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const uint4 q4 = ok ? *(const uint4*)src : z, k4 = ok ? *(const uint4*)(src + p.C) : z;
    const uint4 g4 = ok ? *(const uint4*)(p.dout + (long long)row[t] * p.C + un.h * 32 + 8 * g) : z;
    qf[t] = __builtin_bit_cast(bf16x8_t, q4); kf[t] = __builtin_bit_cast(bf16x8_t, k4);
    vf[t] = __builtin_bit_cast(bf16x8_t, ok ? *(const uint4*)(src + 2 * p.C) : z);
    gf[t] = __builtin_bit_cast(bf16x8_t, g4);
    *(uint4*)(kimg + (16 * t + fr) * 32 + 8 * g) = k4;
    *(uint4*)(qimg + (16 * t + fr) * 32 + 8 * g) = q4;
    *(uint4*)(gimg + (16 * t + fr) * 32 + 8 * g) = g4;
  }
  float lse[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lse[i] = p.lse[(long long)u * 64 + 16 * i + fr];
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  // ---- orientation T: a query on a lane (S^T[key][query]) ----
  f32x4_t acc[4][4], dp[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], qf[i], zero, 0, 0, 0);
      dp[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[j], gf[i], zero, 0, 0, 0);      // dP^T[key][query] = sum_d V[key][d] dO[query][d]
    }
  win_scores<SHIFT>(p, un, acc, qid, fr, g);
  float delta[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float pr = __expf(acc[j][i][r] - lse[i]); acc[j][i][r] = pr; d += pr * dp[j][i][r]; }
    d += __shfl_xor(d, 16, 64); d += __shfl_xor(d, 32, 64);
    delta[i] = d;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[j][i][r] *= dp[j][i][r] - d;                                        // dS^T
      }
    // d bias of this (window, head): its own [64][64] slab (the caller sums the slabs over images and windows - atomics into the 49 x 49
    // table of a head, 59 M of them per layer at batch 128, were 3/4 of this kernel's time)
    if (p.dbias) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(float4*)(p.dbias + ((long long)u * 64 + 16 * i + fr) * 64 + 16 * j + 4 * g) = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
    }
  }
  // dQ^T[d][query] = scale sum_key K^T[d][key] dS^T[key][query]
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const bf16x8_t k0 = gather_t(kimg, dt, 0, fr, g), k1 = gather_t(kimg, dt, 1, fr, g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4_t o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, pack_acc2(acc[0][i], acc[1][i]), zero, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, pack_acc2(acc[2][i], acc[3][i]), o, 0, 0, 0);
      if (16 * i + fr < 49)
        *(uint2*)(p.dqkv + (long long)row[i] * C3 + un.h * 32 + 16 * dt + 4 * g) =
            make_uint2(pack2bf(o[0] * p.scale, o[1] * p.scale), pack2bf(o[2] * p.scale, o[3] * p.scale));
    }
  }
  // ---- orientation N: a key on a lane (S[query][key]): P and dS again, as B operands [k = query][n = key] ----
  // per-lane ids / lse / delta of the 16 queries it now holds as rows: from the lanes that hold them as columns
  f32x4_t pn[4][4], dn[4][4];                                 // [query tile i][key tile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pn[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], kf[j], zero, 0, 0, 0);       // S[query][key]: rows = queries 4g+r, col = key fr
      dn[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[i], vf[j], zero, 0, 0, 0);       // dP[query][key]
    }
  int kid[4] = {0, 0, 0, 0};
  if (SHIFT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) kid[j] = qid[j];              // lane fr of tile j: the token's id, as a key now
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ql = 4 * g + r;                                // query 16 i + ql lives on lane ql (any group) in orientation T
      const float lq = __shfl(lse[i], ql, 64), dq = __shfl(delta[i], ql, 64);
      const int iq = SHIFT ? __shfl(qid[i], ql, 64) : 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = pn[i][j][r] * p.scale + p.bias[((long long)un.h * 64 + 16 * i + ql) * 64 + 16 * j + fr];
        if (SHIFT && kid[j] != iq) s -= 100.f;
        const float pr = __expf(s - lq);
        pn[i][j][r] = pr;
        dn[i][j][r] = pr * (dn[i][j][r] - dq);
      }
    }
  }
  // dV^T[d][key] = sum_query dO^T[d][query] P[query][key];  dK^T[d][key] = scale sum_query Q^T[d][query] dS[query][key]
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const bf16x8_t g0 = gather_t(gimg, dt, 0, fr, g), g1 = gather_t(gimg, dt, 1, fr, g);
    const bf16x8_t q0 = gather_t(qimg, dt, 0, fr, g), q1 = gather_t(qimg, dt, 1, fr, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4_t ov = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g0, pack_acc2(pn[0][j], pn[1][j]), zero, 0, 0, 0);
      ov = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g1, pack_acc2(pn[2][j], pn[3][j]), ov, 0, 0, 0);
      f32x4_t ok_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0, pack_acc2(dn[0][j], dn[1][j]), zero, 0, 0, 0);
      ok_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1, pack_acc2(dn[2][j], dn[3][j]), ok_, 0, 0, 0);
      if (16 * j + fr < 49) {
        bf16_t* dst = p.dqkv + (long long)row[j] * C3 + un.h * 32 + 16 * dt + 4 * g;
        *(uint2*)(dst + p.C) = make_uint2(pack2bf(ok_[0] * p.scale, ok_[1] * p.scale), pack2bf(ok_[2] * p.scale, ok_[3] * p.scale));
        *(uint2*)(dst + 2 * p.C) = make_uint2(pack2bf(ov[0], ov[1]), pack2bf(ov[2], ov[3]));
      }
    }
  }
}

static int win_check(const void* qkv, int B, int H, int W, int C, int heads, int shift) {
  if (!qkv) return MM_ERR_ARG;
  if (B <= 0 || H < 7 || W < 7 || (H % 7) || (W % 7) || heads <= 0 || C != heads * 32 || shift < 0 || shift >= 7) return MM_ERR_SHAPE;
  if (shift && (H == 7 || W == 7)) return MM_ERR_SHAPE;        // a single window is never shifted (modeling_swin.py set_shift_and_window_size)
  if ((long long)B * H * W * 3 * C >= (1ll << 31)) return MM_ERR_SHAPE;
  return MM_OK;
}

// qkv [B*H*W, 3C] bf16 (q | k | v, head h at columns h*32), bias [heads][64][64] fp32 (relative position bias of the 49 x 49 token pairs,
// key columns >= 49 hold -30000, everything else 0), out [B*H*W, C] bf16, lse [B * (H/7) * (W/7) * heads][64] fp32 (for the backward).
extern "C" int medmoe_win_attn_fwd(const void* qkv, const float* bias, void* out, float* lse, int B, int H, int W, int C, int heads,
                                   int shift, hipStream_t stream) {
  const int rc = win_check(qkv, B, H, W, C, heads, shift);
  if (rc != MM_OK) return rc;
  if (!bias || !out) return MM_ERR_ARG;
  WinArgs p = {};
  p.qkv = (const bf16_t*)qkv; p.out = (bf16_t*)out; p.bias = bias; p.lse = lse;
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.shift = shift; p.n_units = B * (H / 7) * (W / 7) * heads;
  p.scale = 0.17677669529663687f;                               // 1 / sqrt(32)
  const dim3 grid((p.n_units + 3) / 4), block(256);
  if (shift) hipLaunchKernelGGL(win_attn_fwd_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(win_attn_fwd_kernel<false>, grid, block, 0, stream, p);
  return mm_check_launch();
}

// dqkv [B*H*W, 3C] bf16 is fully written; dbias (may be NULL): [B * (H/7) * (W/7) * heads][64][64] fp32, one slab of dS per (image, window,
// head) (rows = queries, columns = keys; entries beyond 49 are zero or padding) - summed over images and windows it is the bias gradient.
extern "C" int medmoe_win_attn_bwd(const void* qkv, const float* bias, const void* dout, const float* lse, void* dqkv, float* dbias, int B,
                                   int H, int W, int C, int heads, int shift, hipStream_t stream) {
  const int rc = win_check(qkv, B, H, W, C, heads, shift);
  if (rc != MM_OK) return rc;
  if (!bias || !dout || !lse || !dqkv) return MM_ERR_ARG;
  WinArgs p = {};
  p.qkv = (const bf16_t*)qkv; p.dout = (const bf16_t*)dout; p.dqkv = (bf16_t*)dqkv; p.bias = bias; p.dbias = dbias; p.lse = (float*)lse;
  p.B = B; p.H = H; p.W = W; p.C = C; p.heads = heads; p.shift = shift; p.n_units = B * (H / 7) * (W / 7) * heads;
  p.scale = 0.17677669529663687f;
  const dim3 grid((p.n_units + 3) / 4), block(256);
  if (shift) hipLaunchKernelGGL(win_attn_bwd_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(win_attn_bwd_kernel<false>, grid, block, 0, stream, p);
  return mm_check_launch();
}

// SwinPatchMerging's concat (modeling_swin.py: input_feature_0 = x[:, 0::2, 0::2], _1 = x[:, 1::2, 0::2], _2 = x[:, 0::2, 1::2], _3 = x[:, 1::2, 1::2]):
// y[b, Y, X, q*C + c] = x[b, 2Y + (q & 1), 2X + (q >> 1), c].  SCATTER = the same map read the other way (the gradient).
template <bool SCATTER>
__global__ __launch_bounds__(256) void patch_merge_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int B, int H, int W, int C) {
  const int c8 = C / 8;
  const long long total = (long long)B * (H / 2) * (W / 2) * 4 * c8;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ch = i % c8;
    long long r = i / c8;
    const int q = r % 4; r /= 4;
    const int X = r % (W / 2); r /= (W / 2);
    const int Y = r % (H / 2);
    const int b = r / (H / 2);
    const long long big = (((long long)b * H + 2 * Y + (q & 1)) * W + 2 * X + (q >> 1)) * C + ch * 8;
    const long long merged = ((((long long)b * (H / 2) + Y) * (W / 2) + X) * 4 + q) * C + ch * 8;
    if (SCATTER) *(uint4*)(dst + big) = *(const uint4*)(src + merged);
    else *(uint4*)(dst + merged) = *(const uint4*)(src + big);
  }
}

extern "C" int medmoe_patch_merge(const void* src, void* dst, int B, int H, int W, int C, int scatter, hipStream_t stream) {
  if (!src || !dst) return MM_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || (H % 2) || (W % 2) || (C % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)B * (H / 2) * (W / 2) * 4 * (C / 8);
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  if (scatter) hipLaunchKernelGGL(patch_merge_kernel<true>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, B, H, W, C);
  else hipLaunchKernelGGL(patch_merge_kernel<false>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, B, H, W, C);
  return mm_check_launch();
}

// Stochastic depth (modeling_swin.py SwinDropPath on the attention branch of a SwinLayer): out[b, l, :] = res[b, l, :] + scale[b] * branch[b, l, :]
// with scale[b] = mask_b / keep_prob; res == nullptr: out = scale[b] * branch (the gradient of the branch).
__global__ __launch_bounds__(256) void drop_path_kernel(const bf16_t* __restrict__ branch, const bf16_t* __restrict__ res,
                                                        const float* __restrict__ scale, bf16_t* __restrict__ out, long long per_sample8,
                                                        long long total8) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
    const float sc = scale[i / per_sample8];
    const uint4 bv = *(const uint4*)(branch + i * 8);
    const uint32_t bw[4] = {bv.x, bv.y, bv.z, bv.w};
    uint32_t ow[4];
    if (res) {
      const uint4 rv = *(const uint4*)(res + i * 8);
      const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ow[q] = pack2bf(__uint_as_float(rw[q] << 16) + sc * __uint_as_float(bw[q] << 16),
                        __uint_as_float(rw[q] & 0xffff0000u) + sc * __uint_as_float(bw[q] & 0xffff0000u));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) ow[q] = pack2bf(sc * __uint_as_float(bw[q] << 16), sc * __uint_as_float(bw[q] & 0xffff0000u));
    }
    *(uint4*)(out + i * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

extern "C" int medmoe_drop_path(const void* branch, const void* res, const float* scale, void* out, int B, long long per_sample,
                                hipStream_t stream) {
  if (!branch || !scale || !out) return MM_ERR_ARG;
  if (B <= 0 || per_sample <= 0 || (per_sample % 8)) return MM_ERR_SHAPE;
  const long long total8 = (long long)B * per_sample / 8;
  const int grid = (int)min((total8 + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(drop_path_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)branch, (const bf16_t*)res, scale, (bf16_t*)out,
                     per_sample / 8, total8);
  return mm_check_launch();
}
