// LayerNorm forward / backward (HBM-bound; statistics in fp32 = Fp32LayerNorm,
// reference normalizations.py:8-19).  One wave per row, 16-byte bf16 loads, values kept
// in registers between the mean and the variance pass.
#include "common.h"

#define LN_MAX_CHUNKS 4   // D <= 64 lanes * 8 * 4 = 2048

template <typename OutT, int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, OutT* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps, const int* __restrict__ rows_dev) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nchunk = D >> 3;
  if (rows_dev) rows = min(rows, *rows_dev);              // packed variable-length batches: the row count lives on the device
  for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
    const bf16_t* xr = x + (long long)row * D;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
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
    const float mean = wave_sum(s) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (lane + i * 64 < nchunk)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
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
  const int grid = min((rows + 3) / 4, 256 * 8);
  const int nch = (D / 8 + 63) / 64;
#define LN_FWD(T, N) hipLaunchKernelGGL((layernorm_fwd_kernel<T, N>), dim3(grid), dim3(256), 0, stream, \
                                        (const bf16_t*)x, gamma, beta, (T*)y, mean, rstd, rows, D, eps, rows_dev)
#define LN_FWD_N(T) do { if (nch == 1) LN_FWD(T, 1); else if (nch == 2) LN_FWD(T, 2); else if (nch == 3) LN_FWD(T, 3); else LN_FWD(T, 4); } while (0)
  if (out_f32) LN_FWD_N(float); else LN_FWD_N(bf16_t);
  return mm_check_launch();
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) [+ add];  dgamma += dy*xhat; dbeta += dy
// LNB_WAVES waves per workgroup: the dgamma / dbeta partials of a workgroup's waves meet in LDS and leave as ONE atomic per column and
// workgroup.  With 4-wave workgroups (1024 of them) every launch ended in 1.57 M atomics onto the same 1536 addresses - serialised at the
// memory side, a third of the launch at 25216 rows; 16 waves x 512 workgroups: 0.79 M.
constexpr int LNB_WAVES = 16;
template <int NCH>
__global__ __launch_bounds__(LNB_WAVES * 64) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const bf16_t* __restrict__ add,
                                                            bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int rows, int D) {
  __shared__ float red[2][LNB_WAVES][512];   // [dgamma|dbeta][wave][lane*8+e] for one chunk slot at a time
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nchunk = D >> 3;
  float ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }

  for (int row = blockIdx.x * LNB_WAVES + wid; row < rows; row += gridDim.x * LNB_WAVES) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NCH][8], dg[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
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
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nchunk) {
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
  const int grid = min((rows + LNB_WAVES - 1) / LNB_WAVES, 256 * 2);
  const int nch = (D / 8 + 63) / 64;
#define LN_BWD(N) hipLaunchKernelGGL((layernorm_bwd_kernel<N>), dim3(grid), dim3(LNB_WAVES * 64), 0, stream, (const bf16_t*)dy, \
                                     (const bf16_t*)x, mean, rstd, gamma, (const bf16_t*)add, (bf16_t*)dx, dgamma, dbeta, rows, D)
  if (nch == 1) LN_BWD(1); else if (nch == 2) LN_BWD(2); else if (nch == 3) LN_BWD(3); else LN_BWD(4);
  return mm_check_launch();
}
