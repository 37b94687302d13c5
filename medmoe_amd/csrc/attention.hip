// Multi-head self-attention core for short sequences (N <= 272 keys, head_dim 64):
// softmax(Q K^T / sqrt(hd) [+ key mask]) V, forward and backward, one workgroup per
// (batch, head).  Semantics = F.scaled_dot_product_attention as called by the reference
// (multi_head_attention.py:62-78); q/k/v are read straight out of the fused QKV projection
// buffer [B*N, 3D] (chunk(3) + view + transpose of :62-69 become pointer arithmetic).
//
// The whole key range of one (b,h) fits in LDS, so there is no online-softmax loop: a wave
// owns 16 query rows, holds the full 16 x NK score block in MFMA accumulators, and reduces
// the row max / row sum with two wave shuffles.  Score tiles are computed TRANSPOSED
// (S^T = K Q^T) so the accumulator registers are already the B operand of the P.V product
// (sum over the accumulator's row index needs no lane movement; cdna guide section 3).
#include "common.h"

#define HD 64

__device__ __forceinline__ int tpos(int r) {   // position of key/query r in a [d][pos] image
  const int o = r & 31;
  return (r & ~31) + (((o & 15) >> 2) << 3) + ((o >> 4) << 2) + (o & 3);
}

// [rows][64] bf16 -> LDS rows of 128 B with chunk ^= row&7 (conflict-free ds_read_b128 fragments).
// CNT = n_pad*8/256 loads per thread, ALL issued before the first LDS write (a load->store loop exposes
// one full memory latency per iteration: 14 serialized round trips per workgroup in the first version).
template <int CNT>
__device__ __forceinline__ void fill_rowmajor(char* lds, const bf16_t* src, long long row_stride, int n_valid, int tid) {
  uint4 v[CNT];
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 3, c = idx & 7;
    v[i] = *(const uint4*)(src + (long long)min(row, n_valid - 1) * row_stride + c * 8);
  }
#pragma unroll
  for (int i = 0; i < CNT; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 3, c = idx & 7;
    *(uint4*)(lds + row * 128 + ((c ^ (row & 7)) << 4)) = v[i];
  }
}

// [rows][64] bf16 -> LDS [64 d][stride bytes] with element r at tpos(r); pad positions zeroed first
__device__ __forceinline__ void fill_transposed(char* lds, const bf16_t* src, long long row_stride, int n_valid,
                                                int stride_bytes, int tid) {
  for (int i = tid; i < (64 * stride_bytes) / 16; i += 256) *(uint4*)(lds + i * 16) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int idx = tid; idx < n_valid * 8; idx += 256) {
    const int row = idx >> 3, c = idx & 7;
    const uint4 v = *(const uint4*)(src + (long long)row * row_stride + c * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    const int pos = tpos(row);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      *(bf16_t*)(lds + (c * 8 + 2 * e) * stride_bytes + pos * 2) = (bf16_t)(w[e] & 0xffff);
      *(bf16_t*)(lds + (c * 8 + 2 * e + 1) * stride_bytes + pos * 2) = (bf16_t)(w[e] >> 16);
    }
  }
}

// MFMA operand whose contraction index is the ROW of a row-major [rows][64] LDS image (V for P.V, K for
// dS.K, dO / Q for the dV / dK products): two ds_read_b64_tr_b16 (cdna guide T10) straight from the
// chunk^(row&7)-swizzled image - conflict-free, no transposed copy, EXEC must be all ones.
// Elements 0..3 = rows row0 + 4*(lane>>4) + {0..3}, elements 4..7 = the same rows + 16; column = col0 + (lane&15).
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int row0, int col0, int lane) {
  const int q = (lane & 15) >> 2, pp = lane & 3;
  const int ra = row0 + 4 * (lane >> 4) + q, rb = ra + 16;
  const int col = col0 + 4 * pp;
  const int ch = col >> 3, within = (col & 7) << 1;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4_t*)(img + ra * 128 + ((ch ^ (ra & 7)) << 4) + within));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4_t*)(img + rb * 128 + ((ch ^ (rb & 7)) << 4) + within));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8_t load_frag_global(const bf16_t* base, long long row_stride, int row, int chunk) {
  const uint4 v = *(const uint4*)(base + (long long)row * row_stride + chunk * 8);
  return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t pack_frag(const f32x4_t& a, const f32x4_t& b) {
  uint4 v;
  v.x = pack2bf(a[0], a[1]); v.y = pack2bf(a[2], a[3]);
  v.z = pack2bf(b[0], b[1]); v.w = pack2bf(b[2], b[3]);
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int NKT>
struct AttnGeom {
  static constexpr int NKP = NKT * 16;            // padded rows
  static constexpr int KS = (NKT + 1) / 2;        // 32-wide contraction steps over keys / queries
  static constexpr int RMROWS = KS * 32;          // rows of a row-major image (zero/duplicate padded to whole 32-row k-steps)
  static constexpr int RM_BYTES = RMROWS * 128;
  static constexpr int FILL = RMROWS * 8 / 256;   // 16-B loads per thread to fill one row-major image
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
#ifdef ATTN_TIMING
__device__ unsigned long long g_attn_timing[8];       // fill, scores, softmax, PV + store, count (tools/attn_timing.hip)
#define ATTN_T(...) __VA_ARGS__
#else
#define ATTN_T(...)
#endif

template <int NKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       float* __restrict__ lse, const unsigned char* __restrict__ key_mask,
                                                       int N, int H, float scale) {
  using G = AttnGeom<NKT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * G::RM_BYTES + G::NKP * 4];
  char* sK = smem;
  char* sV = smem + G::RM_BYTES;
  float* sMask = (float*)(smem + 2 * G::RM_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  const bf16_t* base = qkv + (long long)b * N * rs + h * HD;

  ATTN_T(long long t0 = clock64(), ts = 0, tx = 0, tp = 0;)
  fill_rowmajor<G::FILL>(sK, base + D, rs, N, tid);
  fill_rowmajor<G::FILL>(sV, base + 2 * D, rs, N, tid);
  for (int k = tid; k < G::NKP; k += 256)
    sMask[k] = (k < N && (!key_mask || key_mask[(long long)b * N + k])) ? 0.f : -INFINITY;
  __syncthreads();

  ATTN_T(const long long t1 = clock64();)
  const int fr = lane & 15, g = lane >> 4;
  const int nqb = (N + 15) >> 4;
  bf16x8_t qf[2], qn[2];
  {
    const int qc0 = min(wid * 16 + fr, N - 1);
    qf[0] = load_frag_global(base, rs, qc0, g);
    qf[1] = load_frag_global(base, rs, qc0, 4 + g);
  }
  for (int qb = wid; qb < nqb; qb += 4) {
    const int q = qb * 16 + fr;
    {                                     // next block's Q fragments: in flight under this block's MFMAs
      const int qcn = min((qb + 4) * 16 + fr, N - 1);
      qn[0] = load_frag_global(base, rs, qcn, g);
      qn[1] = load_frag_global(base, rs, qcn, 4 + g);
    }
    ATTN_T(const long long a0 = clock64();)
    f32x4_t s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      const char* kr = sK + (kt * 16 + fr) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(kr + (((ks * 4 + g) ^ (fr & 7)) << 4));
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
      }
    }
    ATTN_T(__builtin_amdgcn_sched_barrier(0); const long long a1 = clock64(); __builtin_amdgcn_sched_barrier(0); ts += a1 - a0;)
    // softmax in the exp2 domain on register PAIRS (v_pk_fma / v_pk_add / v_pk_mul: two scores per instruction):
    // t = s * (scale * log2 e) + mask ; p = 2^(t - max t) ; the row's natural-log LSE = (max t + log2 sum) * ln 2
    const float c2 = scale * 1.44269504088896f;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const float4 mk = *(const float4*)(sMask + kt * 16 + g * 4);
      const f32x2_t t0 = (f32x2_t){s[kt][0], s[kt][1]} * c2 + (f32x2_t){mk.x, mk.y};
      const f32x2_t t1 = (f32x2_t){s[kt][2], s[kt][3]} * c2 + (f32x2_t){mk.z, mk.w};
      s[kt][0] = t0[0]; s[kt][1] = t0[1]; s[kt][2] = t1[0]; s[kt][3] = t1[1];
      m = fmaxf(fmaxf(m, t0[0]), fmaxf(t0[1], fmaxf(t1[0], t1[1])));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float ms = (m == -INFINITY) ? 0.f : m;        // a fully masked row: 2^(-inf - 0) = 0 instead of nan
    f32x2_t l2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const f32x2_t d0 = (f32x2_t){s[kt][0], s[kt][1]} - ms, d1 = (f32x2_t){s[kt][2], s[kt][3]} - ms;
      const f32x2_t e0 = {__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
      const f32x2_t e1 = {__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
      s[kt][0] = e0[0]; s[kt][1] = e0[1]; s[kt][2] = e1[0]; s[kt][3] = e1[1];
      l2 += e0 + e1;
    }
    float l = l2[0] + l2[1];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (g == 0 && q < N) lse[((long long)b * H + h) * N + q] = (m + __log2f(l)) * 0.693147180559945f;
    ATTN_T(__builtin_amdgcn_sched_barrier(0); const long long a2 = clock64(); __builtin_amdgcn_sched_barrier(0); tx += a2 - a1;)
    bf16x8_t pf[G::KS];
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < G::KS; ++t) {
      f32x4_t a = s[2 * t] * inv;
      f32x4_t c = (2 * t + 1 < NKT) ? s[(2 * t + 1 < NKT) ? 2 * t + 1 : 0] * inv : zero4;
      pf[t] = pack_frag(a, c);
    }
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      f32x4_t o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < G::KS; ++t) {
        const bf16x8_t vf = tr_frag(sV, t * 32, nd * 16, lane);
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[t], o, 0, 0, 0);
      }
      if (q < N) {
        uint2 pk; pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]);
        *(uint2*)(out + ((long long)b * N + q) * D + h * HD + nd * 16 + g * 4) = pk;
      }
    }
    qf[0] = qn[0]; qf[1] = qn[1];
    ATTN_T(__builtin_amdgcn_sched_barrier(0); tp += clock64() - a2;)
  }
  ATTN_T(if (lane == 0 && wid == 0) { atomicAdd(&g_attn_timing[0], (unsigned long long)(t1 - t0)); atomicAdd(&g_attn_timing[1], (unsigned long long)ts);
           atomicAdd(&g_attn_timing[2], (unsigned long long)tx); atomicAdd(&g_attn_timing[3], (unsigned long long)tp);
           atomicAdd(&g_attn_timing[4], (unsigned long long)(clock64() - t0)); atomicAdd(&g_attn_timing[5], 1ull); })
}

// ------------------------------------------------------------------------------------------
// backward pass 1: dQ (wave owns 16 queries, loops all keys);  also writes delta = rowsum(dO*O)
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          const unsigned char* __restrict__ key_mask,
                                                          bf16_t* __restrict__ dqkv, float* __restrict__ delta,
                                                          int N, int H, float scale) {
  using G = AttnGeom<NKT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * G::RM_BYTES + G::NKP * 4];
  char* sK = smem;
  char* sV = smem + G::RM_BYTES;
  float* sMask = (float*)(smem + 2 * G::RM_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  const bf16_t* base = qkv + (long long)b * N * rs + h * HD;
  const bf16_t* obase = o + (long long)b * N * D + h * HD;
  const bf16_t* dobase = dout + (long long)b * N * D + h * HD;

  fill_rowmajor<G::FILL>(sK, base + D, rs, N, tid);
  fill_rowmajor<G::FILL>(sV, base + 2 * D, rs, N, tid);
  for (int k = tid; k < G::NKP; k += 256)
    sMask[k] = (k < N && (!key_mask || key_mask[(long long)b * N + k])) ? 0.f : -INFINITY;
  __syncthreads();

  const int fr = lane & 15, g = lane >> 4;
  const int nqb = (N + 15) >> 4;
  for (int qb = wid; qb < nqb; qb += 4) {
    const int q = qb * 16 + fr;
    const int qc = min(q, N - 1);
    bf16x8_t qf[2], dof[2], of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = load_frag_global(base, rs, qc, ks * 4 + g);
      dof[ks] = load_frag_global(dobase, D, qc, ks * 4 + g);
      of[ks] = load_frag_global(obase, D, qc, ks * 4 + g);
    }
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[ks][e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float L = lse[((long long)b * H + h) * N + qc];
    if (g == 0 && q < N) delta[((long long)b * H + h) * N + q] = dl;

    f32x4_t acc[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) acc[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < G::KS; ++t) {
      f32x4_t ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kt = 2 * t + u;
        ds[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (kt < NKT) {
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          const char* kr = sK + (kt * 16 + fr) * 128;
          const char* vr = sV + (kt * 16 + fr) * 128;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int off = ((ks * 4 + g) ^ (fr & 7)) << 4;
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(kr + off), qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(vr + off), dof[ks], dp, 0, 0, 0);
          }
          const float4 mk = *(const float4*)(sMask + kt * 16 + g * 4);
          const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(s[r] * scale + mkv[r] - L);
            ds[u][r] = p * (dp[r] - dl) * scale;
          }
        }
      }
      const bf16x8_t dsf = pack_frag(ds[0], ds[1]);
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        const bf16x8_t ktf = tr_frag(sK, t * 32, nd * 16, lane);
        acc[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf, acc[nd], 0, 0, 0);
      }
    }
    if (q < N) {
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        uint2 pk; pk.x = pack2bf(acc[nd][0], acc[nd][1]); pk.y = pack2bf(acc[nd][2], acc[nd][3]);
        *(uint2*)(dqkv + ((long long)b * N + q) * rs + h * HD + nd * 16 + g * 4) = pk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward pass 2: dK, dV (wave owns 16 keys, loops all queries)
// ------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           const unsigned char* __restrict__ key_mask,
                                                           bf16_t* __restrict__ dqkv, int N, int H, float scale) {
  using G = AttnGeom<NKT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * G::RM_BYTES + G::NKP * 12];
  char* sQ = smem;
  char* sDO = smem + G::RM_BYTES;
  float* sMask = (float*)(smem + 2 * G::RM_BYTES);
  float* sLse = sMask + G::NKP;
  float* sDelta = sLse + G::NKP;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  const bf16_t* base = qkv + (long long)b * N * rs + h * HD;
  const bf16_t* dobase = dout + (long long)b * N * D + h * HD;

  fill_rowmajor<G::FILL>(sQ, base, rs, N, tid);
  fill_rowmajor<G::FILL>(sDO, dobase, D, N, tid);
  for (int k = tid; k < G::NKP; k += 256) {
    const bool valid = k < N;
    sMask[k] = (valid && (!key_mask || key_mask[(long long)b * N + k])) ? 0.f : -INFINITY;
    sLse[k] = valid ? lse[((long long)b * H + h) * N + k] : INFINITY;    // padded query rows -> p = 0
    sDelta[k] = valid ? delta[((long long)b * H + h) * N + k] : 0.f;
  }
  __syncthreads();

  const int fr = lane & 15, g = lane >> 4;
  const int nkb = (N + 15) >> 4;
  for (int kb = wid; kb < nkb; kb += 4) {
    const int key = kb * 16 + fr;
    const int kc = min(key, N - 1);
    bf16x8_t kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = load_frag_global(base + D, rs, kc, ks * 4 + g);
      vf[ks] = load_frag_global(base + 2 * D, rs, kc, ks * 4 + g);
    }
    const float mk = sMask[kb * 16 + fr];
    f32x4_t dk[4], dv[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) { dk[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

#pragma unroll 1
    for (int t = 0; t < G::KS; ++t) {
      f32x4_t pp[2], dss[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * t + u;
        if (qt < NKT) {
          // S[q][key] tile: rows q = qt*16 + 4g + r (registers), column key = lane&15
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          const char* qr = sQ + (qt * 16 + fr) * 128;
          const char* dr = sDO + (qt * 16 + fr) * 128;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int off = ((ks * 4 + g) ^ (fr & 7)) << 4;
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(qr + off), kf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)(dr + off), vf[ks], dp, 0, 0, 0);
          }
          const float4 L4 = *(const float4*)(sLse + qt * 16 + g * 4);
          const float4 D4 = *(const float4*)(sDelta + qt * 16 + g * 4);
          const float Lv[4] = {L4.x, L4.y, L4.z, L4.w}, Dv[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __expf(s[r] * scale + mk - Lv[r]);
            pp[u][r] = p;
            dss[u][r] = p * (dp[r] - Dv[r]) * scale;
          }
        } else {
          pp[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          dss[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
      }
      const bf16x8_t pf = pack_frag(pp[0], pp[1]);
      const bf16x8_t dsf = pack_frag(dss[0], dss[1]);
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        const bf16x8_t dot = tr_frag(sDO, t * 32, nd * 16, lane);
        const bf16x8_t qt_ = tr_frag(sQ, t * 32, nd * 16, lane);
        dv[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf, dv[nd], 0, 0, 0);
        dk[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_, dsf, dk[nd], 0, 0, 0);
      }
    }
    if (key < N) {
      bf16_t* row = dqkv + ((long long)b * N + key) * rs + h * HD;
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        uint2 pk; pk.x = pack2bf(dk[nd][0], dk[nd][1]); pk.y = pack2bf(dk[nd][2], dk[nd][3]);
        *(uint2*)(row + D + nd * 16 + g * 4) = pk;
        pk.x = pack2bf(dv[nd][0], dv[nd][1]); pk.y = pack2bf(dv[nd][2], dv[nd][3]);
        *(uint2*)(row + 2 * D + nd * 16 + g * 4) = pk;
      }
    }
  }
}

static int pick_nkt(int N) {
  if (N <= 80) return 5;
  if (N <= 208) return 13;
  if (N <= 272) return 17;
  return 0;
}

extern "C" int medmoe_attn_fwd(const void* qkv, void* out, float* lse, const unsigned char* key_mask,
                               int B, int N, int H, int head_dim, hipStream_t stream) {
  if (!qkv || !out || !lse) return MM_ERR_ARG;
  if (head_dim != HD || B <= 0 || H <= 0 || N <= 0) return MM_ERR_SHAPE;
  const int nkt = pick_nkt(N);
  if (!nkt) return MM_ERR_SHAPE;
  const float scale = 0.125f;
#define AF(K) hipLaunchKernelGGL((attn_fwd_kernel<K>), dim3(B * H), dim3(256), 0, stream, (const bf16_t*)qkv, \
                                 (bf16_t*)out, lse, key_mask, N, H, scale)
  if (nkt == 5) AF(5); else if (nkt == 13) AF(13); else AF(17);
  return mm_check_launch();
}

extern "C" int medmoe_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                               const unsigned char* key_mask, void* dqkv, float* delta, int B, int N,
                               int H, int head_dim, hipStream_t stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || !delta) return MM_ERR_ARG;
  if (head_dim != HD || B <= 0 || H <= 0 || N <= 0) return MM_ERR_SHAPE;
  const int nkt = pick_nkt(N);
  if (!nkt) return MM_ERR_SHAPE;
  const float scale = 0.125f;
#define ABQ(K) hipLaunchKernelGGL((attn_bwd_dq_kernel<K>), dim3(B * H), dim3(256), 0, stream, (const bf16_t*)qkv, \
                                  (const bf16_t*)out, (const bf16_t*)dout, lse, key_mask, (bf16_t*)dqkv, delta, N, H, scale)
#define ABK(K) hipLaunchKernelGGL((attn_bwd_dkv_kernel<K>), dim3(B * H), dim3(256), 0, stream, (const bf16_t*)qkv, \
                                  (const bf16_t*)dout, lse, delta, key_mask, (bf16_t*)dqkv, N, H, scale)
  if (nkt == 5) { ABQ(5); ABK(5); } else if (nkt == 13) { ABQ(13); ABK(13); } else { ABQ(17); ABK(17); }
  return mm_check_launch();
}
