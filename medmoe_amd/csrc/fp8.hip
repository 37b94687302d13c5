// fp8 (OCP e4m3fn) expert weights on the CDNA4 fp8 MFMA (BASELINE.json configs[4]).
//
// The expert projections of reference swin.py:18-30 (Conv1d k=1 = Linear) run as
//     Y[m, n] = epi( sa[m] * sb[g][n] * sum_k Aq[m, k] * Bq[g][n, k]  (+ bias[g][n]) )
// with Aq = the activation rows quantised to e4m3 with ONE dynamic scale per row (sa = amax / 448), Bq = the expert's weight
// quantised with one scale per output channel (sb), fp32 accumulation in v_mfma_f32_16x16x32_fp8_fp8.  The fp32 master weights
// stay in the flat parameter buffer (Adam updates them; medmoe_quant_weights_e4m3 re-derives Bq, its transpose and sb after
// every step).  Backward: dgrad uses the TRANSPOSED fp8 weights with the output-channel scales folded into the incoming
// gradient rows before those are quantised (medmoe_quant_rows_e4m3 with a per-group column scale); wgrad stays bf16
// (gemm_tn on the saved bf16 activations) into the fp32 master gradient = straight-through estimator.
//
//   medmoe_quant_rows_e4m3     bf16 rows (optionally gathered, optionally x per-group column scale) -> e4m3 rows + row scales
//   medmoe_quant_weights_e4m3  fp32 [G][N][K] -> e4m3 [G][N][K], e4m3 transposed [G][K][N], scales [G][N]
//   medmoe_gemm_fp8_grouped    the grouped NT product above on 128-row tiles of the dispatch table (moe.hip), 128x128 tile per
//                              workgroup, k-step 64, register-staged double buffer; epilogues: bias + ReLU, (+ residual) x ReLU'(aux)
#include "common.h"

typedef long i64_t;

__device__ __forceinline__ uint32_t cvt4_e4m3(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);      // bytes 0, 1 (round to nearest even, saturating)
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);       // bytes 2, 3
  return (uint32_t)w;
}

// one wave per row; K % 8 == 0.  group_of_row: expert of a row = slot_expert[row / rows_per_slot] (column scales are per expert)
__global__ __launch_bounds__(256) void quant_rows_kernel(const bf16_t* __restrict__ x, int ldx, const int* __restrict__ rowmap,
                                                         const float* __restrict__ colscale, const int* __restrict__ slot_expert,
                                                         int rows_per_slot, uint8_t* __restrict__ q, float* __restrict__ s, int M, int K) {
  const int lane = threadIdx.x & 63;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
    const bf16_t* xr = x + (long long)(rowmap ? rowmap[row] : row) * ldx;
    const float* cs = colscale ? colscale + (long long)(slot_expert ? slot_expert[row / rows_per_slot] : 0) * K : nullptr;
    float amax = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
      const uint4 v = *(const uint4*)(xr + c);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
        if (cs) { lo *= cs[c + 2 * e]; hi *= cs[c + 2 * e + 1]; }
        amax = fmaxf(amax, fmaxf(fabsf(lo), fabsf(hi)));
      }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) s[row] = sc;
    for (int c = lane * 8; c < K; c += 512) {
      const uint4 v = *(const uint4*)(xr + c);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      float f[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(w[e] << 16); f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        if (cs) { f[2 * e] *= cs[c + 2 * e]; f[2 * e + 1] *= cs[c + 2 * e + 1]; }
      }
      uint2 o;
      o.x = cvt4_e4m3(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
      o.y = cvt4_e4m3(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
      *(uint2*)(q + (long long)row * K + c) = o;
    }
  }
}

extern "C" int medmoe_quant_rows_e4m3(const void* x, int ldx, const int* rowmap, const float* colscale, const int* slot_expert,
                                      int rows_per_slot, void* q, float* s, int M, int K, hipStream_t stream) {
  if (!x || !q || !s) return MM_ERR_ARG;
  if (M <= 0 || K <= 0 || (K % 8) || (ldx % 8) || (slot_expert && rows_per_slot <= 0)) return MM_ERR_SHAPE;
  const int grid = min((M + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(quant_rows_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, rowmap, colscale, slot_expert,
                     rows_per_slot, (uint8_t*)q, s, M, K);
  return mm_check_launch();
}

// one wave per weight row (g, n): scale = amax_k |w| / 448; q[g][n][k], qT[g][k][n]
__global__ __launch_bounds__(256) void quant_weights_kernel(const float* __restrict__ w, uint8_t* __restrict__ q, uint8_t* __restrict__ qT,
                                                            float* __restrict__ s, int G, int N, int K) {
  const int lane = threadIdx.x & 63;
  const int rows = G * N;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const float* wr = w + (long long)row * K;
    float amax = 0.f;
    for (int c = lane; c < K; c += 64) amax = fmaxf(amax, fabsf(wr[c]));
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) s[row] = sc;
    const int g = row / N, n = row - g * N;
    for (int c = lane * 4; c < K; c += 256) {
      const float4 v = *(const float4*)(wr + c);
      const uint32_t o = cvt4_e4m3(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
      *(uint32_t*)(q + (long long)row * K + c) = o;
      if (qT) {
        uint8_t* t = qT + ((long long)g * K + c) * N + n;
        t[0] = (uint8_t)o; t[N] = (uint8_t)(o >> 8); t[2ll * N] = (uint8_t)(o >> 16); t[3ll * N] = (uint8_t)(o >> 24);
      }
    }
  }
}

extern "C" int medmoe_quant_weights_e4m3(const float* w, void* q, void* qT, float* s, int G, int N, int K, hipStream_t stream) {
  if (!w || !q || !s) return MM_ERR_ARG;
  if (G <= 0 || N <= 0 || K <= 0 || (K % 4)) return MM_ERR_SHAPE;
  const int grid = min((G * N + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(quant_weights_kernel, dim3(grid), dim3(256), 0, stream, w, (uint8_t*)q, (uint8_t*)qT, s, G, N, K);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// grouped fp8 GEMM on the 128-row tile table of medmoe_dispatch: tiles[t] = {group, m0, m_end, -}
// ---------------------------------------------------------------------------------------------
#define F8_BM 128
#define F8_BN 128
#define F8_BK 64
#define F8_LDS_ROW 80            // 64 data bytes + 16 pad: 16-B aligned rows whose 8-B fragment reads spread over the banks
#define F8_EPI_RELU 1            // relu(acc + bias)
#define F8_EPI_MUL_DRELU 2       // (acc + residual) * (aux > 0)

struct GemmF8Args {
  const uint8_t* A; const float* sa; const uint8_t* B; const float* sb; const float* bias;
  bf16_t* C; const bf16_t* residual; const bf16_t* aux; const int* tiles; const int* tile_count;
  int N, K, ldc, epi;
  long long strideB, strideSb, strideBias;
};

__global__ __launch_bounds__(256) void gemm_fp8_grouped_kernel(GemmF8Args p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (F8_BM + F8_BN) * F8_LDS_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int wm = wid & 1, wn = wid >> 1;                       // 2 x 2 waves of 64 x 64
  const int tiles_n = (p.N + F8_BN - 1) / F8_BN;
  const int n_tiles = p.tile_count[0];
  const int id = blockIdx.x;
  const int ti = id / tiles_n, tn_ = id - ti * tiles_n;
  if (ti >= n_tiles) return;
  const int group = p.tiles[ti * 4], m0 = p.tiles[ti * 4 + 1], m_end = p.tiles[ti * 4 + 2];
  const int n0 = tn_ * F8_BN;
  const uint8_t* Bg = p.B + (long long)group * p.strideB;

  // staging: 128 rows x 64 B per operand and k-step = 512 pieces of 16 B, two per thread and operand
  const uint8_t* srcA[2]; const uint8_t* srcB[2];
  int dst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 2, c = idx & 3;
    srcA[i] = p.A + (long long)min(m0 + row, m_end - 1) * p.K + c * 16;     // rows past the group's end: clamped copies, never stored
    srcB[i] = Bg + (long long)min(n0 + row, p.N - 1) * p.K + c * 16;
    dst[i] = row * F8_LDS_ROW + c * 16;
  }
  auto bufA = [&](int b) { return smem + b * (F8_BM + F8_BN) * F8_LDS_ROW; };
  auto bufB = [&](int b) { return smem + b * (F8_BM + F8_BN) * F8_LDS_ROW + F8_BM * F8_LDS_ROW; };
  uint4 ra[2], rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { ra[i] = *(const uint4*)(srcA[i] + k0); rb[i] = *(const uint4*)(srcB[i] + k0); }
  };
  auto lstore = [&](int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { *(uint4*)(bufA(b) + dst[i]) = ra[i]; *(uint4*)(bufB(b) + dst[i]) = rb[i]; }
  };
  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nk = p.K / F8_BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * F8_BK);
    const char* sA = bufA(cur) + (wm * 64 + fr) * F8_LDS_ROW + g * 8;
    const char* sB = bufB(cur) + (wn * 64 + fr) * F8_LDS_ROW + g * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i64_t af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *(const i64_t*)(sA + t * 16 * F8_LDS_ROW + ks * 32);
        bf[t] = *(const i64_t*)(sB + t * 16 * F8_LDS_ROW + ks * 32);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)      // D[i = n][j = m]: lane holds row m = fr of the tile, columns n = 4g + r
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bf[tn], af[tm], acc[tm][tn], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }
  // epilogue
  const float* sbg = p.sb ? p.sb + (long long)group * p.strideSb : nullptr;
  const float* bg = p.bias ? p.bias + (long long)group * p.strideBias : nullptr;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m0 + wm * 64 + tm * 16 + fr;
    if (m >= m_end) continue;
    const float sam = p.sa[m];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wn * 64 + tn * 16 + g * 4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[tm][tn][r] * sam * (sbg ? sbg[n + r] : 1.f);
        if (bg) v[r] += bg[n + r];
      }
      const long long o = (long long)m * p.ldc + n;
      if (p.epi == F8_EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (p.epi == F8_EPI_MUL_DRELU) {
        const uint2 rs = p.residual ? *(const uint2*)(p.residual + o) : make_uint2(0u, 0u);
        const uint2 ax = *(const uint2*)(p.aux + o);
        const float res[4] = {__uint_as_float(rs.x << 16), __uint_as_float(rs.x & 0xffff0000u), __uint_as_float(rs.y << 16), __uint_as_float(rs.y & 0xffff0000u)};
        const float au[4] = {__uint_as_float(ax.x << 16), __uint_as_float(ax.x & 0xffff0000u), __uint_as_float(ax.y << 16), __uint_as_float(ax.y & 0xffff0000u)};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = au[r] > 0.f ? v[r] + res[r] : 0.f;
      }
      uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
      *(uint2*)(p.C + o) = pk;
    }
  }
}

extern "C" int medmoe_gemm_fp8_grouped(const void* Aq, const float* sa, const void* Bq, const float* sb, const float* bias, void* C, int ldc,
                                       const void* residual, const void* aux, const int* tiles, const int* tile_count, int max_tiles,
                                       int N, int K, long long strideB, long long strideSb, long long strideBias, int epi,
                                       hipStream_t stream) {
  if (!Aq || !sa || !Bq || !C || !tiles || !tile_count || max_tiles <= 0) return MM_ERR_ARG;
  if (N <= 0 || K <= 0 || (K % F8_BK) || (N % 4) || (ldc % 4)) return MM_ERR_SHAPE;
  if (epi < 0 || epi > F8_EPI_MUL_DRELU || (epi == F8_EPI_MUL_DRELU && !aux)) return MM_ERR_ARG;
  GemmF8Args p;
  p.A = (const uint8_t*)Aq; p.sa = sa; p.B = (const uint8_t*)Bq; p.sb = sb; p.bias = bias; p.C = (bf16_t*)C;
  p.residual = (const bf16_t*)residual; p.aux = (const bf16_t*)aux; p.tiles = tiles; p.tile_count = tile_count;
  p.N = N; p.K = K; p.ldc = ldc; p.epi = epi; p.strideB = strideB; p.strideSb = strideSb; p.strideBias = strideBias;
  const int tiles_n = (N + F8_BN - 1) / F8_BN;
  hipLaunchKernelGGL(gemm_fp8_grouped_kernel, dim3(max_tiles * tiles_n), dim3(256), 0, stream, p);
  return mm_check_launch();
}
