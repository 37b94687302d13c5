// Multi-head self-attention core (head_dim 64).  Two kernel families: RESIDENT (below; N <= 272 keys, the whole key range of a
// (b, h) in LDS) and STREAMING (further down; any N, the default).  What follows describes the resident family:
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
// RT = threads per workgroup of the RESIDENT kernels (template parameter): eight waves share one K / V image, so two workgroups per CU put FOUR waves
// on every SIMD (the kernels are latency chains - LDS read -> MFMA -> cross-lane reduction -> MFMA - and with the two waves per
// SIMD of the 256-thread form the CU sat idle most of the time: 10.9 us per workgroup for ~3 us of issue work).
// (RT = 256 for <= 80 keys - five query blocks would leave three of eight waves idle -, 512 up to 272 keys, 1024 for the 577 tokens
// of ViT-L/14 at 336 px, whose K and V images fill the LDS of a CU with one workgroup of sixteen waves.)
template <int CNT, int RT>
__device__ __forceinline__ void fill_rowmajor(char* lds, const bf16_t* src, long long row_stride, int n_valid, int tid) {
  // CNT = total 16-B pieces of the image (rows * 8); every thread issues ALL its loads before the first LDS write
  constexpr int PER = (CNT + RT - 1) / RT;
  uint4 v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = min(tid + i * RT, CNT - 1);
    const int row = idx >> 3, c = idx & 7;
    v[i] = *(const uint4*)(src + (long long)min(row, n_valid - 1) * row_stride + c * 8);
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = tid + i * RT;
    const int row = idx >> 3, c = idx & 7;
    if (idx < CNT) *(uint4*)(lds + row * 128 + ((c ^ (row & 7)) << 4)) = v[i];
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
  static constexpr int FILL = RMROWS * 8;         // 16-B pieces of one row-major image
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

// Forward, resident K / V.  A wave owns 16 query rows and walks the keys in CHUNKS of 8 tiles (128 keys) with an online
// softmax (running row maximum / sum in the exp2 domain), so only 8 score tiles are live at a time: the kernel fits 128
// registers and eight waves per workgroup x two workgroups per CU put four waves on every SIMD (the kernel is a latency chain:
// LDS read -> MFMA -> cross-lane reduction -> exp -> MFMA; with two waves per SIMD the CU idled most of the time).
// MASKED = false (image tower: no key mask): the only invalid keys are the clamped copies of key N-1 that pad the last
// tiles; they cannot raise the row maximum, so the maximum is taken over the raw scores (v_max3, no mask adds), the exponent
// is ONE fma per pair (scale and -max folded), padded keys are zeroed only in the tiles that hold any, and the 1 / rowsum goes
// onto the 16 output values instead of the probabilities.
template <int NKT, bool MASKED, int RT>
__global__ __launch_bounds__(RT, RT == 256 ? 2 : 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         float* __restrict__ lse, const unsigned char* __restrict__ key_mask,
                                                         int Nmax, int H, float scale, const int* __restrict__ seq_off) {
  using G = AttnGeom<NKT>;
  constexpr int CH = 8;                                   // key tiles per chunk (even: P.V contracts 32 keys per MFMA)
  constexpr int NCH = (NKT + CH - 1) / CH;
  __shared__ __attribute__((aligned(16))) char smem[2 * G::RM_BYTES + G::NKP * 4];
  char* sK = smem;
  char* sV = smem + G::RM_BYTES;
  float* sMask = (float*)(smem + 2 * G::RM_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  // seq_off (variable-length batches, text_pack): sequence b = rows seq_off[b] .. seq_off[b+1] of the packed qkv / out, every key valid
  const long long row0 = seq_off ? (long long)seq_off[b] : (long long)b * Nmax;
  const int N = seq_off ? max(1, min(seq_off[b + 1] - seq_off[b], Nmax)) : Nmax;
  const bf16_t* base = qkv + row0 * rs + h * HD;

  fill_rowmajor<G::FILL, RT>(sK, base + D, rs, N, tid);
  fill_rowmajor<G::FILL, RT>(sV, base + 2 * D, rs, N, tid);
  for (int k = tid; k < G::NKP; k += RT)
    sMask[k] = (k < N && (!key_mask || key_mask[(long long)b * Nmax + k])) ? 0.f : -INFINITY;
  __syncthreads();

  const int fr = lane & 15, g = lane >> 4;
  const int nqb = (N + 15) >> 4;
  const float c2 = scale * 1.44269504088896f;
  bf16x8_t qf[2], qn[2];
  {
    const int qc0 = min(wid * 16 + fr, N - 1);
    qf[0] = load_frag_global(base, rs, qc0, g);
    qf[1] = load_frag_global(base, rs, qc0, 4 + g);
  }
  for (int qb = wid; qb < nqb; qb += RT / 64) {
    const int q = qb * 16 + fr;
    {                                     // next block's Q fragments: in flight under this block's MFMAs
      const int qcn = min((qb + RT / 64) * 16 + fr, N - 1);
      qn[0] = load_frag_global(base, rs, qcn, g);
      qn[1] = load_frag_global(base, rs, qcn, 4 + g);
    }
    float m = -INFINITY, lp = 0.f;        // running maximum (exp2 domain) and this lane's partial row sum
    f32x4_t o[4];
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) o[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {                       // NOT unrolled: the chunks' score tiles must not be live together
      const int kt0 = c * CH;
      f32x4_t s[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        s[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (kt0 + u < NKT) {
          const char* kr = sK + ((kt0 + u) * 16 + fr) * 128;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t kf = *(const bf16x8_t*)(kr + (((ks * 4 + g) ^ (fr & 7)) << 4));
            s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[u], 0, 0, 0);
          }
        }
      }
      float alpha;
      f32x2_t l2 = {0.f, 0.f};
      if constexpr (MASKED) {
        float bm = -INFINITY;
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (kt0 + u < NKT) {
            const float4 mk = *(const float4*)(sMask + (kt0 + u) * 16 + g * 4);
            const f32x2_t t0 = (f32x2_t){s[u][0], s[u][1]} * c2 + (f32x2_t){mk.x, mk.y};
            const f32x2_t t1 = (f32x2_t){s[u][2], s[u][3]} * c2 + (f32x2_t){mk.z, mk.w};
            s[u][0] = t0[0]; s[u][1] = t0[1]; s[u][2] = t1[0]; s[u][3] = t1[1];
            bm = fmaxf(fmaxf(bm, fmaxf(t0[0], t0[1])), fmaxf(t1[0], t1[1]));
          }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        const float ms = (mn == -INFINITY) ? 0.f : mn;    // nothing but masked keys so far: 2^(-inf - 0) = 0 instead of nan
        alpha = __builtin_amdgcn_exp2f(m - ms);
        m = mn;
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (kt0 + u < NKT) {
            const f32x2_t d0 = (f32x2_t){s[u][0], s[u][1]} - ms, d1 = (f32x2_t){s[u][2], s[u][3]} - ms;
            const f32x2_t e0 = {__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
            const f32x2_t e1 = {__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
            s[u][0] = e0[0]; s[u][1] = e0[1]; s[u][2] = e1[0]; s[u][3] = e1[1];
            l2 += e0 + e1;
          }
      } else {
        float bm = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int u = 1; u < CH; ++u)
          if (kt0 + u < NKT) bm = fmaxf(fmaxf(bm, fmaxf(s[u][0], s[u][1])), fmaxf(s[u][2], s[u][3]));     // -> v_max3_f32
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm * c2);
        alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        const float nm = -mn;
#pragma unroll
        for (int u = 0; u < CH; ++u)
          if (kt0 + u < NKT) {
            const f32x2_t d0 = (f32x2_t){s[u][0], s[u][1]} * c2 + nm, d1 = (f32x2_t){s[u][2], s[u][3]} * c2 + nm;
            f32x2_t e0 = {__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
            f32x2_t e1 = {__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
            if ((kt0 + u) * 16 + 16 > N) {                // a tile with padded keys (block-uniform branch)
              const int k0 = (kt0 + u) * 16 + g * 4;
              e0[0] = k0 < N ? e0[0] : 0.f; e0[1] = k0 + 1 < N ? e0[1] : 0.f;
              e1[0] = k0 + 2 < N ? e1[0] : 0.f; e1[1] = k0 + 3 < N ? e1[1] : 0.f;
            }
            s[u][0] = e0[0]; s[u][1] = e0[1]; s[u][2] = e1[0]; s[u][3] = e1[1];
            l2 += e0 + e1;
          }
      }
      lp = lp * alpha + l2[0] + l2[1];
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        if (c > 0) o[nd] *= alpha;
#pragma unroll
        for (int t = 0; t < CH / 2; ++t)
          if (kt0 + 2 * t < NKT) {                        // UNNORMALISED probabilities (<= 1); a missing odd tile holds zeros
            const bf16x8_t pf = pack_frag(s[2 * t], s[2 * t + 1]);
            const bf16x8_t vf = tr_frag(sV, (kt0 + 2 * t) * 16, nd * 16, lane);
            o[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[nd], 0, 0, 0);
          }
      }
    }
    float l = lp;
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    if (q < N) {
      if (g == 0) lse[((long long)b * H + h) * Nmax + q] = (m + __log2f(l)) * 0.693147180559945f;
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        const f32x4_t v = o[nd] * inv;
        uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
        *(uint2*)(out + (row0 + q) * D + h * HD + nd * 16 + g * 4) = pk;
      }
    }
    qf[0] = qn[0]; qf[1] = qn[1];
  }
}

// ------------------------------------------------------------------------------------------
// backward pass 1: dQ (wave owns 16 queries, loops all keys);  also writes delta = rowsum(dO*O)
// ------------------------------------------------------------------------------------------
template <int NKT, bool MASKED, int RT>
__global__ __launch_bounds__(RT, 4) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
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

  fill_rowmajor<G::FILL, RT>(sK, base + D, rs, N, tid);
  fill_rowmajor<G::FILL, RT>(sV, base + 2 * D, rs, N, tid);
  for (int k = tid; k < G::NKP; k += RT)
    sMask[k] = (k < N && (!key_mask || key_mask[(long long)b * N + k])) ? 0.f : -INFINITY;
  __syncthreads();

  const int fr = lane & 15, g = lane >> 4;
  const int nqb = (N + 15) >> 4;
  bf16x8_t qf[2], dof[2], of[2], qn[2], don[2], on[2];
  {
    const int qc0 = min(wid * 16 + fr, N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = load_frag_global(base, rs, qc0, ks * 4 + g);
      dof[ks] = load_frag_global(dobase, D, qc0, ks * 4 + g);
      of[ks] = load_frag_global(obase, D, qc0, ks * 4 + g);
    }
  }
  for (int qb = wid; qb < nqb; qb += RT / 64) {
    const int q = qb * 16 + fr;
    const int qc = min(q, N - 1);
    {                                     // the next block's Q / dO / O fragments: in flight under this block's MFMAs (as the forward kernel does)
      const int qcn = min((qb + RT / 64) * 16 + fr, N - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qn[ks] = load_frag_global(base, rs, qcn, ks * 4 + g);
        don[ks] = load_frag_global(dobase, D, qcn, ks * 4 + g);
        on[ks] = load_frag_global(obase, D, qcn, ks * 4 + g);
      }
    }
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[ks][e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    const float c2 = scale * 1.44269504088896f;
    const float nL2 = -lse[((long long)b * H + h) * N + qc] * 1.44269504088896f;     // p = 2^(s c2 - L log2 e)
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
          // dS / scale = p (dP - delta): the softmax scale is applied once, to the 16 accumulated dQ values
          if (MASKED || kt * 16 + 16 > N) {               // mask adds only where a key can be invalid (block-uniform branch)
            const float4 mk = *(const float4*)(sMask + kt * 16 + g * 4);
            const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[u][r] = __builtin_amdgcn_exp2f(s[r] * c2 + (mkv[r] + nL2)) * (dp[r] - dl);
          } else {
            const f32x2_t t0 = (f32x2_t){s[0], s[1]} * c2 + nL2, t1 = (f32x2_t){s[2], s[3]} * c2 + nL2;
            const f32x2_t g0 = (f32x2_t){dp[0], dp[1]} - dl, g1 = (f32x2_t){dp[2], dp[3]} - dl;
            const f32x2_t d0 = (f32x2_t){__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} * g0;
            const f32x2_t d1 = (f32x2_t){__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} * g1;
            ds[u] = (f32x4_t){d0[0], d0[1], d1[0], d1[1]};
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
        const f32x4_t v = acc[nd] * scale;
        uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
        *(uint2*)(dqkv + ((long long)b * N + q) * rs + h * HD + nd * 16 + g * 4) = pk;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { qf[ks] = qn[ks]; dof[ks] = don[ks]; of[ks] = on[ks]; }
  }
}

// ------------------------------------------------------------------------------------------
// backward pass 2: dK, dV (wave owns 16 keys, loops all queries)
// ------------------------------------------------------------------------------------------
template <int NKT, int RT>
__global__ __launch_bounds__(RT, 4) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
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

  fill_rowmajor<G::FILL, RT>(sQ, base, rs, N, tid);
  fill_rowmajor<G::FILL, RT>(sDO, dobase, D, N, tid);
  for (int k = tid; k < G::NKP; k += RT) {
    const bool valid = k < N;
    sMask[k] = (valid && (!key_mask || key_mask[(long long)b * N + k])) ? 0.f : -INFINITY;
    sLse[k] = valid ? lse[((long long)b * H + h) * N + k] * 1.44269504088896f : INFINITY;    // exp2 domain; padded query rows -> p = 0
    sDelta[k] = valid ? delta[((long long)b * H + h) * N + k] : 0.f;
  }
  __syncthreads();

  const int fr = lane & 15, g = lane >> 4;
  const int nkb = (N + 15) >> 4;
  bf16x8_t kf[2], vf[2], kn[2], vn[2];
  {
    const int kc0 = min(wid * 16 + fr, N - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[ks] = load_frag_global(base + D, rs, kc0, ks * 4 + g);
      vf[ks] = load_frag_global(base + 2 * D, rs, kc0, ks * 4 + g);
    }
  }
  for (int kb = wid; kb < nkb; kb += RT / 64) {
    const int key = kb * 16 + fr;
    {                                     // the next key block's K / V fragments: in flight under this block's MFMAs
      const int kcn = min((kb + RT / 64) * 16 + fr, N - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kn[ks] = load_frag_global(base + D, rs, kcn, ks * 4 + g);
        vn[ks] = load_frag_global(base + 2 * D, rs, kcn, ks * 4 + g);
      }
    }
    const float mk = sMask[kb * 16 + fr];
    const float c2 = scale * 1.44269504088896f;
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
          // p = 2^(s c2 + mask - L log2 e) on pairs; dS / scale = p (dP - delta): the scale goes onto the 16 dK values at the end
          const f32x2_t t0 = (f32x2_t){s[0], s[1]} * c2 + (mk - (f32x2_t){Lv[0], Lv[1]});
          const f32x2_t t1 = (f32x2_t){s[2], s[3]} * c2 + (mk - (f32x2_t){Lv[2], Lv[3]});
          const f32x2_t p0 = {__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])};
          const f32x2_t p1 = {__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])};
          const f32x2_t d0 = p0 * ((f32x2_t){dp[0], dp[1]} - (f32x2_t){Dv[0], Dv[1]});
          const f32x2_t d1 = p1 * ((f32x2_t){dp[2], dp[3]} - (f32x2_t){Dv[2], Dv[3]});
          pp[u] = (f32x4_t){p0[0], p0[1], p1[0], p1[1]};
          dss[u] = (f32x4_t){d0[0], d0[1], d1[0], d1[1]};
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
        const f32x4_t kv = dk[nd] * scale;
        uint2 pk; pk.x = pack2bf(kv[0], kv[1]); pk.y = pack2bf(kv[2], kv[3]);
        *(uint2*)(row + D + nd * 16 + g * 4) = pk;
        pk.x = pack2bf(dv[nd][0], dv[nd][1]); pk.y = pack2bf(dv[nd][2], dv[nd][3]);
        *(uint2*)(row + 2 * D + nd * 16 + g * 4) = pk;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
  }
}

// ==========================================================================================
// STREAMING kernels (any N): flash-attention organisation.  A workgroup owns 64 rows of one (b, h) - four waves of 16 -
// and walks the other sequence dimension in blocks of 64 rows that are staged through LDS, double-buffered: the global
// loads of block i+1 are in flight while block i is computed, one barrier per block.  The forward keeps a running row
// maximum / sum (online softmax) in the exp2 domain; the two backward kernels recompute P from the saved row LSE.
// Against the resident kernels above: 33 KB of LDS and < 128 registers instead of 56 KB / 246, so four workgroups share a
// CU (16 waves: one wave's softmax VALU work runs under another's MFMAs), K/V never have to fit LDS (N = 577 of
// ViT-L/14 at 336 px), and the work items are 64-row chunks (13 query tiles of N = 197 -> 4 + 4 + 4 + 1 waves instead of
// one workgroup whose waves take 4, 3, 3, 3 tiles).  The workgroups of one (b, h) are neighbours on one XCD (xcd_remap),
// so its K / V are fetched from HBM once and re-read from that XCD's L2.
// ==========================================================================================
#define SB 64                      // rows per streamed block
#define SB_BYTES (SB * 128)        // one [64][64] bf16 row-major image

__device__ __forceinline__ void stage_load(uint4 (&v)[2], const bf16_t* src, long long row_stride, int row0, int n_valid, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 3, c = idx & 7;
    v[i] = *(const uint4*)(src + (long long)min(row0 + row, n_valid - 1) * row_stride + c * 8);
  }
}
__device__ __forceinline__ void stage_store(char* lds, const uint4 (&v)[2], int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 3, c = idx & 7;
    *(uint4*)(lds + row * 128 + ((c ^ (row & 7)) << 4)) = v[i];
  }
}
// A-operand fragment of a row-major swizzled [64][64] image: row = 16-row tile `t16` * 16 + (lane & 15), d-chunk ks*4 + (lane >> 4)
__device__ __forceinline__ bf16x8_t rm_frag(const char* img, int t16, int ks, int fr, int g) {
  return *(const bf16x8_t*)(img + (t16 * 16 + fr) * 128 + (((ks * 4 + g) ^ (fr & 7)) << 4));
}

template <bool MASKED>
__global__ __launch_bounds__(256) void attn_fwd_stream_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              float* __restrict__ lse, const unsigned char* __restrict__ key_mask,
                                                              int N, int H, int QB, float scale) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SB_BYTES + 2 * SB * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = id / QB, qc = id - bh * QB;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  const bf16_t* base = qkv + (long long)b * N * rs + h * HD;
  const int q0 = qc * 64 + wid * 16;
  const bool active = q0 < N;                         // wave-uniform: a wave without queries only helps staging
  const int q = q0 + fr;
  bf16x8_t qf[2];
  {
    const int qr = min(q, N - 1);
    qf[0] = load_frag_global(base, rs, qr, g);
    qf[1] = load_frag_global(base, rs, qr, 4 + g);
  }
  const int nkb = (N + SB - 1) / SB;
  auto bufK = [&](int i) { return smem + i * 2 * SB_BYTES; };
  auto bufV = [&](int i) { return smem + i * 2 * SB_BYTES + SB_BYTES; };
  float* sMaskAll = (float*)(smem + 4 * SB_BYTES);
  auto mask_of = [&](int key) -> float { return (key < N && (!key_mask || key_mask[(long long)b * N + key])) ? 0.f : -INFINITY; };
  uint4 kr[2], vr[2];
  stage_load(kr, base + D, rs, 0, N, tid);
  stage_load(vr, base + 2 * D, rs, 0, N, tid);
  stage_store(bufK(0), kr, tid); stage_store(bufV(0), vr, tid);
  if (tid < SB) sMaskAll[tid] = mask_of(tid);
  __syncthreads();

  const float c2 = scale * 1.44269504088896f;
  float m = -INFINITY, lp = 0.f;
  f32x4_t o[4];
#pragma unroll
  for (int nd = 0; nd < 4; ++nd) o[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nkb;
    float mnext = 0.f;
    if (more) {
      stage_load(kr, base + D, rs, (kb + 1) * SB, N, tid);
      stage_load(vr, base + 2 * D, rs, (kb + 1) * SB, N, tid);
      if (tid < SB) mnext = mask_of((kb + 1) * SB + tid);
    }
    if (active) {
      const char* sK = bufK(cur);
      const char* sV = bufV(cur);
      const float* sM = sMaskAll + cur * SB;
      f32x4_t s[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        s[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rm_frag(sK, kt, ks, fr, g), qf[ks], s[kt], 0, 0, 0);
      }
      float alpha;
      f32x2_t l2 = {0.f, 0.f};
      if constexpr (MASKED) {
        float bm = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const float4 mk = *(const float4*)(sM + kt * 16 + g * 4);
          const f32x2_t t0 = (f32x2_t){s[kt][0], s[kt][1]} * c2 + (f32x2_t){mk.x, mk.y};
          const f32x2_t t1 = (f32x2_t){s[kt][2], s[kt][3]} * c2 + (f32x2_t){mk.z, mk.w};
          s[kt][0] = t0[0]; s[kt][1] = t0[1]; s[kt][2] = t1[0]; s[kt][3] = t1[1];
          bm = fmaxf(fmaxf(bm, fmaxf(t0[0], t0[1])), fmaxf(t1[0], t1[1]));
        }
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);
        const float ms = (mn == -INFINITY) ? 0.f : mn;      // nothing but masked keys so far: 2^(-inf - 0) = 0 instead of nan
        alpha = __builtin_amdgcn_exp2f(m - ms);
        m = mn;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const f32x2_t d0 = (f32x2_t){s[kt][0], s[kt][1]} - ms, d1 = (f32x2_t){s[kt][2], s[kt][3]} - ms;
          const f32x2_t e0 = {__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
          const f32x2_t e1 = {__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
          s[kt][0] = e0[0]; s[kt][1] = e0[1]; s[kt][2] = e1[0]; s[kt][3] = e1[1];
          l2 += e0 + e1;
        }
      } else {
        // no key mask: padded keys are clamped copies of key N-1 (they cannot raise the maximum); raw-score maximum by v_max3,
        // one fma per pair for the exponent, padded keys zeroed only in the last block
        float bm = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) bm = fmaxf(fmaxf(bm, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm * c2);
        alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        const float nm = -mn;
        const bool tail = kb * SB + SB > N;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const f32x2_t d0 = (f32x2_t){s[kt][0], s[kt][1]} * c2 + nm, d1 = (f32x2_t){s[kt][2], s[kt][3]} * c2 + nm;
          f32x2_t e0 = {__builtin_amdgcn_exp2f(d0[0]), __builtin_amdgcn_exp2f(d0[1])};
          f32x2_t e1 = {__builtin_amdgcn_exp2f(d1[0]), __builtin_amdgcn_exp2f(d1[1])};
          if (tail) {
            const int k0 = kb * SB + kt * 16 + g * 4;
            e0[0] = k0 < N ? e0[0] : 0.f; e0[1] = k0 + 1 < N ? e0[1] : 0.f;
            e1[0] = k0 + 2 < N ? e1[0] : 0.f; e1[1] = k0 + 3 < N ? e1[1] : 0.f;
          }
          s[kt][0] = e0[0]; s[kt][1] = e0[1]; s[kt][2] = e1[0]; s[kt][3] = e1[1];
          l2 += e0 + e1;
        }
      }
      lp = lp * alpha + l2[0] + l2[1];                   // per-lane partial row sum: the lane groups are added once, at the end
      bf16x8_t pf[2];
      pf[0] = pack_frag(s[0], s[1]);
      pf[1] = pack_frag(s[2], s[3]);
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        o[nd] *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t) o[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sV, t * 32, nd * 16, lane), pf[t], o[nd], 0, 0, 0);
      }
    }
    if (more) {
      stage_store(bufK(cur ^ 1), kr, tid); stage_store(bufV(cur ^ 1), vr, tid);
      if (tid < SB) sMaskAll[(cur ^ 1) * SB + tid] = mnext;
    }
    __syncthreads();
  }
  if (!active) return;
  float l = lp;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  if (q < N) {
    if (g == 0) lse[((long long)b * H + h) * N + q] = (m + __log2f(l)) * 0.693147180559945f;
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const f32x4_t v = o[nd] * inv;
      uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
      *(uint2*)(out + ((long long)b * N + q) * D + h * HD + nd * 16 + g * 4) = pk;
    }
  }
}

// backward pass 1 (streaming): dQ for 64 queries per workgroup, keys in blocks of 64; also writes delta = rowsum(dO * O)
template <bool MASKED>
__global__ __launch_bounds__(256) void attn_bwd_dq_stream_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                 const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                 const unsigned char* __restrict__ key_mask,
                                                                 bf16_t* __restrict__ dqkv, float* __restrict__ delta,
                                                                 int N, int H, int QB, float scale) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SB_BYTES + 2 * SB * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = id / QB, qc = id - bh * QB;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  const bf16_t* base = qkv + (long long)b * N * rs + h * HD;
  const bf16_t* obase = o + (long long)b * N * D + h * HD;
  const bf16_t* dobase = dout + (long long)b * N * D + h * HD;
  const int q0 = qc * 64 + wid * 16;
  const bool active = q0 < N;
  const int q = q0 + fr, qr = min(q, N - 1);
  bf16x8_t qf[2], dof[2];
  float dl = 0.f;
  {
    bf16x8_t of[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = load_frag_global(base, rs, qr, ks * 4 + g);
      dof[ks] = load_frag_global(dobase, D, qr, ks * 4 + g);
      of[ks] = load_frag_global(obase, D, qr, ks * 4 + g);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)of[ks][e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
  }
  const float c2 = scale * 1.44269504088896f;
  const float nL2 = -lse[((long long)b * H + h) * N + qr] * 1.44269504088896f;
  if (active && g == 0 && q < N) delta[((long long)b * H + h) * N + q] = dl;

  const int nkb = (N + SB - 1) / SB;
  auto bufK = [&](int i) { return smem + i * 2 * SB_BYTES; };
  auto bufV = [&](int i) { return smem + i * 2 * SB_BYTES + SB_BYTES; };
  float* sMaskAll = (float*)(smem + 4 * SB_BYTES);
  auto mask_of = [&](int key) -> float { return (key < N && (!key_mask || key_mask[(long long)b * N + key])) ? 0.f : -INFINITY; };
  uint4 kr[2], vr[2];
  stage_load(kr, base + D, rs, 0, N, tid);
  stage_load(vr, base + 2 * D, rs, 0, N, tid);
  stage_store(bufK(0), kr, tid); stage_store(bufV(0), vr, tid);
  if (tid < SB) sMaskAll[tid] = mask_of(tid);
  __syncthreads();
  f32x4_t acc[4];
#pragma unroll
  for (int nd = 0; nd < 4; ++nd) acc[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const bool more = kb + 1 < nkb;
    float mnext = 0.f;
    if (more) {
      stage_load(kr, base + D, rs, (kb + 1) * SB, N, tid);
      stage_load(vr, base + 2 * D, rs, (kb + 1) * SB, N, tid);
      if (tid < SB) mnext = mask_of((kb + 1) * SB + tid);
    }
    if (active) {
      const char* sK = bufK(cur);
      const char* sV = bufV(cur);
      const float* sM = sMaskAll + cur * SB;
      f32x4_t ds[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rm_frag(sK, kt, ks, fr, g), qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rm_frag(sV, kt, ks, fr, g), dof[ks], dp, 0, 0, 0);
        }
        // dS / scale = p (dP - delta): the softmax scale is applied once, to the accumulated dQ
        if (MASKED || kb * SB + SB > N) {
          const float4 mk = *(const float4*)(sM + kt * 16 + g * 4);
          const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[kt][r] = __builtin_amdgcn_exp2f(s[r] * c2 + (mkv[r] + nL2)) * (dp[r] - dl);
        } else {
          const f32x2_t t0 = (f32x2_t){s[0], s[1]} * c2 + nL2, t1 = (f32x2_t){s[2], s[3]} * c2 + nL2;
          const f32x2_t d0 = (f32x2_t){__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])} * ((f32x2_t){dp[0], dp[1]} - dl);
          const f32x2_t d1 = (f32x2_t){__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])} * ((f32x2_t){dp[2], dp[3]} - dl);
          ds[kt] = (f32x4_t){d0[0], d0[1], d1[0], d1[1]};
        }
      }
      const bf16x8_t dsf0 = pack_frag(ds[0], ds[1]), dsf1 = pack_frag(ds[2], ds[3]);
#pragma unroll
      for (int nd = 0; nd < 4; ++nd) {
        acc[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 0, nd * 16, lane), dsf0, acc[nd], 0, 0, 0);
        acc[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sK, 32, nd * 16, lane), dsf1, acc[nd], 0, 0, 0);
      }
    }
    if (more) {
      stage_store(bufK(cur ^ 1), kr, tid); stage_store(bufV(cur ^ 1), vr, tid);
      if (tid < SB) sMaskAll[(cur ^ 1) * SB + tid] = mnext;
    }
    __syncthreads();
  }
  if (active && q < N) {
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const f32x4_t v = acc[nd] * scale;
      uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
      *(uint2*)(dqkv + ((long long)b * N + q) * rs + h * HD + nd * 16 + g * 4) = pk;
    }
  }
}

// backward pass 2 (streaming): dK, dV for 64 keys per workgroup, queries in blocks of 64
__global__ __launch_bounds__(256) void attn_bwd_dkv_stream_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  const unsigned char* __restrict__ key_mask,
                                                                  bf16_t* __restrict__ dqkv, int N, int H, int QB, float scale) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SB_BYTES + 4 * SB * 4];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = id / QB, kc = id - bh * QB;
  const int b = bh / H, h = bh - b * H;
  const int D = H * HD;
  const long long rs = 3LL * D;
  const bf16_t* base = qkv + (long long)b * N * rs + h * HD;
  const bf16_t* dobase = dout + (long long)b * N * D + h * HD;
  const float* lrow = lse + ((long long)b * H + h) * N;
  const float* drow = delta + ((long long)b * H + h) * N;
  const int k0 = kc * 64 + wid * 16;
  const bool active = k0 < N;
  const int key = k0 + fr, kr_ = min(key, N - 1);
  bf16x8_t kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = load_frag_global(base + D, rs, kr_, ks * 4 + g);
    vf[ks] = load_frag_global(base + 2 * D, rs, kr_, ks * 4 + g);
  }
  const float mk = (key < N && (!key_mask || key_mask[(long long)b * N + key])) ? 0.f : -INFINITY;
  const float c2 = scale * 1.44269504088896f;
  const int nqb = (N + SB - 1) / SB;
  auto bufQ = [&](int i) { return smem + i * 2 * SB_BYTES; };
  auto bufD = [&](int i) { return smem + i * 2 * SB_BYTES + SB_BYTES; };
  float* sL = (float*)(smem + 4 * SB_BYTES);          // [2][64] row LSE (exp2 domain; +inf for padded queries -> p = 0)
  float* sDl = sL + 2 * SB;                            // [2][64] delta
  uint4 qr[2], dr[2];
  stage_load(qr, base, rs, 0, N, tid);
  stage_load(dr, dobase, D, 0, N, tid);
  stage_store(bufQ(0), qr, tid); stage_store(bufD(0), dr, tid);
  if (tid < SB) { sL[tid] = tid < N ? lrow[tid] * 1.44269504088896f : INFINITY; sDl[tid] = tid < N ? drow[tid] : 0.f; }
  __syncthreads();
  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int nd = 0; nd < 4; ++nd) { dk[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[nd] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
  for (int qb = 0; qb < nqb; ++qb) {
    const int cur = qb & 1;
    const bool more = qb + 1 < nqb;
    float ln = 0.f, dn = 0.f;
    if (more) {
      stage_load(qr, base, rs, (qb + 1) * SB, N, tid);
      stage_load(dr, dobase, D, (qb + 1) * SB, N, tid);
      if (tid < SB) {
        const int qq = (qb + 1) * SB + tid;
        ln = qq < N ? lrow[qq] * 1.44269504088896f : INFINITY; dn = qq < N ? drow[qq] : 0.f;
      }
    }
    if (active) {
      const char* sQ = bufQ(cur);
      const char* sDO = bufD(cur);
      const float* cL = sL + cur * SB;
      const float* cD = sDl + cur * SB;
      f32x4_t pp[4], dss[4];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        // S[q][key] tile: rows q = qt*16 + 4g + r (registers), column key = lane & 15
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rm_frag(sQ, qt, ks, fr, g), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rm_frag(sDO, qt, ks, fr, g), vf[ks], dp, 0, 0, 0);
        }
        const float4 L4 = *(const float4*)(cL + qt * 16 + g * 4);
        const float4 D4 = *(const float4*)(cD + qt * 16 + g * 4);
        const float Lv[4] = {L4.x, L4.y, L4.z, L4.w}, Dv[4] = {D4.x, D4.y, D4.z, D4.w};
        const f32x2_t t0 = (f32x2_t){s[0], s[1]} * c2 + (mk - (f32x2_t){Lv[0], Lv[1]});
        const f32x2_t t1 = (f32x2_t){s[2], s[3]} * c2 + (mk - (f32x2_t){Lv[2], Lv[3]});
        const f32x2_t p0 = {__builtin_amdgcn_exp2f(t0[0]), __builtin_amdgcn_exp2f(t0[1])};
        const f32x2_t p1 = {__builtin_amdgcn_exp2f(t1[0]), __builtin_amdgcn_exp2f(t1[1])};
        const f32x2_t d0 = p0 * ((f32x2_t){dp[0], dp[1]} - (f32x2_t){Dv[0], Dv[1]});      // dS / scale: the scale goes onto dK at the end
        const f32x2_t d1 = p1 * ((f32x2_t){dp[2], dp[3]} - (f32x2_t){Dv[2], Dv[3]});
        pp[qt] = (f32x4_t){p0[0], p0[1], p1[0], p1[1]};
        dss[qt] = (f32x4_t){d0[0], d0[1], d1[0], d1[1]};
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8_t pf = pack_frag(pp[2 * t], pp[2 * t + 1]);
        const bf16x8_t dsf = pack_frag(dss[2 * t], dss[2 * t + 1]);
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) {
          dv[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sDO, t * 32, nd * 16, lane), pf, dv[nd], 0, 0, 0);
          dk[nd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sQ, t * 32, nd * 16, lane), dsf, dk[nd], 0, 0, 0);
        }
      }
    }
    if (more) {
      stage_store(bufQ(cur ^ 1), qr, tid); stage_store(bufD(cur ^ 1), dr, tid);
      if (tid < SB) { sL[(cur ^ 1) * SB + tid] = ln; sDl[(cur ^ 1) * SB + tid] = dn; }
    }
    __syncthreads();
  }
  if (active && key < N) {
    bf16_t* row = dqkv + ((long long)b * N + key) * rs + h * HD;
#pragma unroll
    for (int nd = 0; nd < 4; ++nd) {
      const f32x4_t kv = dk[nd] * scale;
      uint2 pk; pk.x = pack2bf(kv[0], kv[1]); pk.y = pack2bf(kv[2], kv[3]);
      *(uint2*)(row + D + nd * 16 + g * 4) = pk;
      pk.x = pack2bf(dv[nd][0], dv[nd][1]); pk.y = pack2bf(dv[nd][2], dv[nd][3]);
      *(uint2*)(row + 2 * D + nd * 16 + g * 4) = pk;
    }
  }
}

// 1 (default) = resident kernels where the keys fit LDS (N <= 272), streaming beyond; 0 = streaming kernels for every N: medmoe_set_option(11, v)
int g_attn_resident = 1;
// threads per workgroup of the resident kernels at 13 key tiles (197 / 196 tokens): 512 = eight waves walk the 13 query blocks in two rounds (five waves take
// two blocks, three take one); 448 = seven waves x two blocks (one idle slot instead of three).  medmoe_set_option(15, 448 | 512)
// Measured (tools/bench_attn_rt.py, 12 heads): forward 423 -> 389 us at batch 1024 but 47 -> 51 us at batch 128; backward 1009 -> 1021 / 119 -> 129 us.
// 0 (default) = 448 for the forward launch of >= 4096 (batch, head) pairs, 512 otherwise.
int g_attn_rt13 = 0;

static int pick_nkt(int N) {
  if (N <= 80) return 5;
  if (N <= 208) return 13;
  if (N <= 272) return 17;
  if (N <= 592) return 37;         // ViT-L/14 at 336 px (577 tokens): K + V = 156 KB, one sixteen-wave workgroup per CU
  return 0;
}

template <int K, int RT>
static void launch_fwd(const void* qkv, void* out, float* lse, const unsigned char* key_mask, int B, int N, int H, float scale, hipStream_t stream) {
  if (key_mask) hipLaunchKernelGGL((attn_fwd_kernel<K, true, RT>), dim3(B * H), dim3(RT), 0, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, key_mask, N, H, scale, (const int*)nullptr);
  else hipLaunchKernelGGL((attn_fwd_kernel<K, false, RT>), dim3(B * H), dim3(RT), 0, stream, (const bf16_t*)qkv, (bf16_t*)out, lse, key_mask, N, H, scale, (const int*)nullptr);
}
template <int K, int RT>
static void launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, const unsigned char* key_mask, void* dqkv,
                       float* delta, int B, int N, int H, float scale, hipStream_t stream) {
  if (key_mask) hipLaunchKernelGGL((attn_bwd_dq_kernel<K, true, RT>), dim3(B * H), dim3(RT), 0, stream, (const bf16_t*)qkv, (const bf16_t*)out,
                                   (const bf16_t*)dout, lse, key_mask, (bf16_t*)dqkv, delta, N, H, scale);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<K, false, RT>), dim3(B * H), dim3(RT), 0, stream, (const bf16_t*)qkv, (const bf16_t*)out,
                          (const bf16_t*)dout, lse, key_mask, (bf16_t*)dqkv, delta, N, H, scale);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<K, RT>), dim3(B * H), dim3(RT), 0, stream, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta,
                     key_mask, (bf16_t*)dqkv, N, H, scale);
}

extern "C" int medmoe_attn_fwd(const void* qkv, void* out, float* lse, const unsigned char* key_mask,
                               int B, int N, int H, int head_dim, hipStream_t stream) {
  if (!qkv || !out || !lse) return MM_ERR_ARG;
  if (head_dim != HD || B <= 0 || H <= 0 || N <= 0) return MM_ERR_SHAPE;
  const int nkt = pick_nkt(N);
  const float scale = 0.125f;
  if (!g_attn_resident || !nkt) {
    const int QB = (N + SB - 1) / SB;
    if (key_mask) hipLaunchKernelGGL(attn_fwd_stream_kernel<true>, dim3(B * H * QB), dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)out,
                                     lse, key_mask, N, H, QB, scale);
    else hipLaunchKernelGGL(attn_fwd_stream_kernel<false>, dim3(B * H * QB), dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)out,
                            lse, key_mask, N, H, QB, scale);
    return mm_check_launch();
  }
  if (nkt == 5) launch_fwd<5, 256>(qkv, out, lse, key_mask, B, N, H, scale, stream);
  else if (nkt == 13 && (g_attn_rt13 == 448 || (g_attn_rt13 == 0 && B * H >= 4096))) launch_fwd<13, 448>(qkv, out, lse, key_mask, B, N, H, scale, stream);
  else if (nkt == 13) launch_fwd<13, 512>(qkv, out, lse, key_mask, B, N, H, scale, stream);
  else if (nkt == 17) launch_fwd<17, 512>(qkv, out, lse, key_mask, B, N, H, scale, stream);
  else launch_fwd<37, 1024>(qkv, out, lse, key_mask, B, N, H, scale, stream);
  return mm_check_launch();
}

// Forward over a PACKED variable-length batch (text_pack): sequence b = rows seq_off[b] .. seq_off[b+1] of qkv / out ([sum len, 3*H*64] /
// [sum len, H*64]), at most Nmax <= 80 tokens each, no padding keys inside a sequence; lse: [B][H][Nmax].
extern "C" int medmoe_attn_fwd_varlen(const void* qkv, void* out, float* lse, const int* seq_off, int B, int Nmax, int H, int head_dim,
                                      hipStream_t stream) {
  if (!qkv || !out || !lse || !seq_off) return MM_ERR_ARG;
  if (head_dim != HD || B <= 0 || H <= 0 || Nmax <= 0 || Nmax > 80) return MM_ERR_SHAPE;
  hipLaunchKernelGGL((attn_fwd_kernel<5, false, 256>), dim3(B * H), dim3(256), 0, stream, (const bf16_t*)qkv, (bf16_t*)out, lse,
                     (const unsigned char*)nullptr, Nmax, H, 0.125f, seq_off);
  return mm_check_launch();
}

extern "C" int medmoe_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                               const unsigned char* key_mask, void* dqkv, float* delta, int B, int N,
                               int H, int head_dim, hipStream_t stream) {
  if (!qkv || !out || !dout || !lse || !dqkv || !delta) return MM_ERR_ARG;
  if (head_dim != HD || B <= 0 || H <= 0 || N <= 0) return MM_ERR_SHAPE;
  const int nkt = pick_nkt(N);
  const float scale = 0.125f;
  if (!g_attn_resident || !nkt) {
    const int QB = (N + SB - 1) / SB;
    if (key_mask) hipLaunchKernelGGL(attn_bwd_dq_stream_kernel<true>, dim3(B * H * QB), dim3(256), 0, stream, (const bf16_t*)qkv,
                                     (const bf16_t*)out, (const bf16_t*)dout, lse, key_mask, (bf16_t*)dqkv, delta, N, H, QB, scale);
    else hipLaunchKernelGGL(attn_bwd_dq_stream_kernel<false>, dim3(B * H * QB), dim3(256), 0, stream, (const bf16_t*)qkv,
                            (const bf16_t*)out, (const bf16_t*)dout, lse, key_mask, (bf16_t*)dqkv, delta, N, H, QB, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_stream_kernel, dim3(B * H * QB), dim3(256), 0, stream, (const bf16_t*)qkv, (const bf16_t*)dout,
                       lse, delta, key_mask, (bf16_t*)dqkv, N, H, QB, scale);
    return mm_check_launch();
  }
  if (nkt == 5) launch_bwd<5, 256>(qkv, out, dout, lse, key_mask, dqkv, delta, B, N, H, scale, stream);
  else if (nkt == 13 && g_attn_rt13 == 448) launch_bwd<13, 448>(qkv, out, dout, lse, key_mask, dqkv, delta, B, N, H, scale, stream);
  else if (nkt == 13) launch_bwd<13, 512>(qkv, out, dout, lse, key_mask, dqkv, delta, B, N, H, scale, stream);
  else if (nkt == 17) launch_bwd<17, 512>(qkv, out, dout, lse, key_mask, dqkv, delta, B, N, H, scale, stream);
  else launch_bwd<37, 1024>(qkv, out, dout, lse, key_mask, dqkv, delta, B, N, H, scale, stream);
  return mm_check_launch();
}
