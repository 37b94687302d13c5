// LayerNorm forward / backward (HBM-bound; statistics in fp32 = Fp32LayerNorm,
// reference normalizations.py:8-19).  One wave per row, 16-byte bf16 loads, values kept
// in registers between the mean and the variance pass.
#include "common.h"

#define LN_MAX_CHUNKS 4   // D <= 64 lanes * 8 * 4 = 2048
int g_ln_one_row = 0;     // medmoe_set_option(16, 1): one row per wave whatever the width (A/B runs)

// LPR lanes per row: 64 (one wave per row) or, for narrow rows (D <= 8 * LPR: the Swin-T stages of 96 / 192 channels), 16 / 32 - a wave then
// normalises 4 / 2 rows at once and the two reductions run over LPR lanes (at D = 96 the one-row form used 12 of 64 lanes)
template <int LPR>
__device__ __forceinline__ float lpr_sum(float v) {
#pragma unroll
  for (int m = LPR / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);      // wave_sum's order: the same bits as the one-row form
  return v;
}

template <typename OutT, int NCH, int LPR = 64>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, OutT* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps, const int* __restrict__ rows_dev) {
  static_assert(LPR == 64 || NCH == 1, "several rows per wave only for rows of one chunk slot");
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane / LPR, cl = lane % LPR;
  const int nchunk = D >> 3;
  if (rows_dev) rows = min(rows, *rows_dev);              // packed variable-length batches: the row count lives on the device
  for (int r0 = (blockIdx.x * 4 + wid) * RPW; r0 < rows; r0 += gridDim.x * 4 * RPW) {
    const int row = r0 + sub;
    const bool rv = row < rows;                           // wave-uniform trip count: every lane takes part in the reductions
    const bf16_t* xr = x + (long long)(rv ? row : r0) * D;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = cl + i * LPR;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
      if (c < nchunk) {
        const uint4 raw = *(const uint4*)(xr + c * 8);
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][2 * e] = __uint_as_float(w[e] << 16);
          v[i][2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
          s += v[i][2 * e] + v[i][2 * e + 1];
        }
      }
    }
    const float mean = lpr_sum<LPR>(s) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (cl + i * LPR < nchunk)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    const float rstd = rsqrtf(lpr_sum<LPR>(sq) / (float)D + eps);
    if (cl == 0 && rv) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = cl + i * LPR;
      if (c < nchunk && rv) {
        const float4 g0 = *(const float4*)(gamma + c * 8), g1 = *(const float4*)(gamma + c * 8 + 4);
        const float4 b0 = *(const float4*)(beta + c * 8), b1 = *(const float4*)(beta + c * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
        if constexpr (sizeof(OutT) == 2) {
          uint4 pk;
          pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]);
          pk.z = pack2bf(o[4], o[5]); pk.w = pack2bf(o[6], o[7]);
          *(uint4*)((bf16_t*)y + (long long)row * D + c * 8) = pk;
        } else {
          float* yr = (float*)y + (long long)row * D + c * 8;
          *(float4*)yr = make_float4(o[0], o[1], o[2], o[3]);
          *(float4*)(yr + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    }
  }
}

static int layernorm_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int D,
                              float eps, int out_f32, const int* rows_dev, hipStream_t stream);
extern "C" int medmoe_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                    float* mean, float* rstd, int rows, int D, float eps,
                                    int out_f32, hipStream_t stream) {
  return layernorm_fwd_impl(x, gamma, beta, y, mean, rstd, rows, D, eps, out_f32, nullptr, stream);
}
// the first min(rows, *rows_dev) rows only (rows_dev: device int)
extern "C" int medmoe_layernorm_fwd_rows(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows,
                                         int D, float eps, int out_f32, const int* rows_dev, hipStream_t stream) {
  if (!rows_dev) return MM_ERR_ARG;
  return layernorm_fwd_impl(x, gamma, beta, y, mean, rstd, rows, D, eps, out_f32, rows_dev, stream);
}
static int layernorm_fwd_impl(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int rows, int D,
                              float eps, int out_f32, const int* rows_dev, hipStream_t stream) {
  if (!x || !gamma || !beta || !y) return MM_ERR_ARG;
  if (rows <= 0 || D <= 0 || (D % 8) || D > 64 * 8 * LN_MAX_CHUNKS) return MM_ERR_SHAPE;
  const int lpr = (g_ln_one_row & 1) ? 64 : D <= 128 ? 16 : D <= 256 ? 32 : 64;   // lanes per row
  const int grid = min((rows + 4 * (64 / lpr) - 1) / (4 * (64 / lpr)), 256 * 8);
  const int nch = (D / 8 + 63) / 64;
#define LN_FWD(T, N) hipLaunchKernelGGL((layernorm_fwd_kernel<T, N>), dim3(grid), dim3(256), 0, stream, \
                                        (const bf16_t*)x, gamma, beta, (T*)y, mean, rstd, rows, D, eps, rows_dev)
#define LN_FWD_L(T, L) hipLaunchKernelGGL((layernorm_fwd_kernel<T, 1, L>), dim3(grid), dim3(256), 0, stream, \
                                          (const bf16_t*)x, gamma, beta, (T*)y, mean, rstd, rows, D, eps, rows_dev)
#define LN_FWD_N(T) do { if (lpr == 16) LN_FWD_L(T, 16); else if (lpr == 32) LN_FWD_L(T, 32); else if (nch == 1) LN_FWD(T, 1); \
                         else if (nch == 2) LN_FWD(T, 2); else if (nch == 3) LN_FWD(T, 3); else LN_FWD(T, 4); } while (0)
  if (out_f32) LN_FWD_N(float); else LN_FWD_N(bf16_t);
  return mm_check_launch();
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) [+ add];  dgamma += dy*xhat; dbeta += dy
// LNB_WAVES waves per workgroup: the dgamma / dbeta partials of a workgroup's waves meet in LDS and leave as ONE atomic per column and
// workgroup.  With 4-wave workgroups (1024 of them) every launch ended in 1.57 M atomics onto the same 1536 addresses - serialised at the
// memory side, a third of the launch at 25216 rows; 16 waves x 512 workgroups: 0.79 M.
constexpr int LNB_WAVES = 16;
template <int NCH, int LPR = 64>
__global__ __launch_bounds__(LNB_WAVES * 64) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const bf16_t* __restrict__ add,
                                                            bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int rows, int D) {
  static_assert(LPR == 64 || NCH == 1, "several rows per wave only for rows of one chunk slot");
  constexpr int RPW = 64 / LPR;
  __shared__ float red[2][LNB_WAVES][512];   // [dgamma|dbeta][wave][lane*8+e] for one chunk slot at a time
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane / LPR, cl = lane % LPR;
  const int nchunk = D >> 3;
  float ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }

  for (int r0 = (blockIdx.x * LNB_WAVES + wid) * RPW; r0 < rows; r0 += gridDim.x * LNB_WAVES * RPW) {
    const int row = min(r0 + sub, rows - 1);
    const bool rv = r0 + sub < rows;                      // wave-uniform trip count; a lane past the last row contributes zeros
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][8], dg[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = cl + i * LPR;
#pragma unroll
      for (int e = 0; e < 8; ++e) { xh[i][e] = 0.f; dg[i][e] = 0.f; }
      if (c < nchunk && rv) {
        const uint4 rx = *(const uint4*)(x + (long long)row * D + c * 8);
        const uint4 rd = *(const uint4*)(dy + (long long)row * D + c * 8);
        const float4 g0 = *(const float4*)(gamma + c * 8), g1 = *(const float4*)(gamma + c * 8 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const uint32_t wx[4] = {rx.x, rx.y, rx.z, rx.w}, wd[4] = {rd.x, rd.y, rd.z, rd.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = __uint_as_float(wx[e] << 16), x1 = __uint_as_float(wx[e] & 0xffff0000u);
          const float d0 = __uint_as_float(wd[e] << 16), d1 = __uint_as_float(wd[e] & 0xffff0000u);
          xh[i][2 * e] = (x0 - mu) * rs; xh[i][2 * e + 1] = (x1 - mu) * rs;
          ab[i][2 * e] += d0; ab[i][2 * e + 1] += d1;
          ag[i][2 * e] += d0 * xh[i][2 * e]; ag[i][2 * e + 1] += d1 * xh[i][2 * e + 1];
          dg[i][2 * e] = d0 * gg[2 * e]; dg[i][2 * e + 1] = d1 * gg[2 * e + 1];
          s1 += dg[i][2 * e] + dg[i][2 * e + 1];
          s2 += dg[i][2 * e] * xh[i][2 * e] + dg[i][2 * e + 1] * xh[i][2 * e + 1];
        }
      }
    }
    const float m1 = lpr_sum<LPR>(s1) / (float)D, m2 = lpr_sum<LPR>(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = cl + i * LPR;
      if (c < nchunk && rv) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rs * (dg[i][e] - m1 - xh[i][e] * m2);
        if (add) {
          const uint4 ra = *(const uint4*)(add + (long long)row * D + c * 8);
          const uint32_t wa[4] = {ra.x, ra.y, ra.z, ra.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[2 * e] += __uint_as_float(wa[e] << 16);
            o[2 * e + 1] += __uint_as_float(wa[e] & 0xffff0000u);
          }
        }
        uint4 pk;
        pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]);
        pk.z = pack2bf(o[4], o[5]); pk.w = pack2bf(o[6], o[7]);
        *(uint4*)(dx + (long long)row * D + c * 8) = pk;
      }
    }
  }
  if (!dgamma) return;
  if constexpr (LPR < 64) {             // the rows of a wave meet first: lane cl of every row group holds the same columns
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1)
#pragma unroll
      for (int e = 0; e < 8; ++e) { ag[0][e] += __shfl_xor(ag[0][e], m, 64); ab[0][e] += __shfl_xor(ab[0][e], m, 64); }
    if (sub != 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { ag[0][e] = 0.f; ab[0][e] = 0.f; }         // lanes >= LPR: columns past D in the block reduction below
    }
  }
  // block reduce the per-wave partials, one atomic per (block, column)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (i * 64 >= nchunk) break;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][wid][lane * 8 + e] = ag[i][e]; red[1][wid][lane * 8 + e] = ab[i][e]; }
    __syncthreads();
    {                                   // 1024 threads: (dgamma | dbeta) x 512 columns of this chunk slot
      const int which = threadIdx.x >> 9, j = threadIdx.x & 511, col = i * 512 + j;
      if (col < D) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_WAVES; ++w) v += red[which][w][j];
        atomicAdd((which ? dbeta : dgamma) + col, v);
      }
    }
  }
}

extern "C" int medmoe_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                    const float* gamma, const void* add, void* dx, float* dgamma,
                                    float* dbeta, int rows, int D, hipStream_t stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx) return MM_ERR_ARG;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return MM_ERR_ARG;
  if (rows <= 0 || D <= 0 || (D % 8) || D > 64 * 8 * LN_MAX_CHUNKS) return MM_ERR_SHAPE;
  const int lpr = (g_ln_one_row & 2) ? 64 : D <= 128 ? 16 : D <= 256 ? 32 : 64;
  const int rpb = LNB_WAVES * (64 / lpr);
  const int grid = min((rows + rpb - 1) / rpb, 256 * 2);
  const int nch = (D / 8 + 63) / 64;
#define LN_BWD(N) hipLaunchKernelGGL((layernorm_bwd_kernel<N>), dim3(grid), dim3(LNB_WAVES * 64), 0, stream, (const bf16_t*)dy, \
                                     (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)add, (bf16_t*)dx, dgamma, dbeta, rows, D)
#define LN_BWD_L(L) hipLaunchKernelGGL((layernorm_bwd_kernel<1, L>), dim3(grid), dim3(LNB_WAVES * 64), 0, stream, (const bf16_t*)dy, \
                                       (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)add, (bf16_t*)dx, dgamma, dbeta, rows, D)
  if (lpr == 16) LN_BWD_L(16); else if (lpr == 32) LN_BWD_L(32); else if (nch == 1) LN_BWD(1); else if (nch == 2) LN_BWD(2); else if (nch == 3) LN_BWD(3); else LN_BWD(4);
  return mm_check_launch();
}
