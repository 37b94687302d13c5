// Modality-specialised MoE glue kernels (reference swin.py:11-117): token mean-pool, the
// router (fixed fp32 operation order => bit-exact top-k indices against the oracle), dispatch
// tables for the grouped expert GEMMs, the fused scale-attention + combine, and their backward.
// The expert projections themselves are grouped bf16 MFMA GEMMs (gemm.hip).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// out[b,c] = (1/cnt) * sum_{t=t0}^{t0+cnt-1} x[b,t,c]   (t ascending, fp32; swin.py:137 / :112)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_tokens_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int Nt,
                                                          int D, int t0, int cnt) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, col = blockIdx.y * 512 + threadIdx.x * 2;
  if (col >= D) return;
  float a0 = 0.f, a1 = 0.f;
  for (int t = t0; t < t0 + cnt; ++t) {
    const uint32_t v = *(const uint32_t*)(x + ((long long)b * Nt + t) * D + col);
    a0 = a0 + __uint_as_float(v << 16);
    a1 = a1 + __uint_as_float(v & 0xffff0000u);
  }
  out[(long long)b * D + col] = a0 / (float)cnt;
  out[(long long)b * D + col + 1] = a1 / (float)cnt;
}

extern "C" int medmoe_mean_tokens(const void* x, float* out, int B, int Nt, int D, int t0, int cnt,
                                  hipStream_t stream) {
  if (!x || !out) return MM_ERR_ARG;
  if (B <= 0 || D <= 0 || (D % 2) || t0 < 0 || cnt <= 0 || t0 + cnt > Nt) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(B, (D + 511) / 512), dim3(256), 0, stream, (const bf16_t*)x, out, Nt, D, t0, cnt);
  return mm_check_launch();
}

// dy[b,t,:] = (t in [t0,t0+cnt)) ? g[b,:]*scale : 0     (backward of the mean-pool)
__global__ __launch_bounds__(256) void broadcast_tokens_kernel(const float* __restrict__ g, bf16_t* __restrict__ dy,
                                                               int B, int Nt, int D, int t0, int cnt, float scale) {
  const long long total = (long long)B * Nt * D / 4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const int col = e % D;
    const long long bt = e / D;
    const int t = bt % Nt, b = bt / Nt;
    uint2 o = make_uint2(0u, 0u);
    if (t >= t0 && t < t0 + cnt) {
      const float4 v = *(const float4*)(g + (long long)b * D + col);
      o.x = pack2bf(v.x * scale, v.y * scale); o.y = pack2bf(v.z * scale, v.w * scale);
    }
    *(uint2*)(dy + e) = o;
  }
}

extern "C" int medmoe_broadcast_tokens(const float* g, void* dy, int B, int Nt, int D, int t0, int cnt, float scale,
                                       hipStream_t stream) {
  if (!g || !dy) return MM_ERR_ARG;
  if (B <= 0 || D <= 0 || (D % 4) || t0 < 0 || cnt <= 0 || t0 + cnt > Nt) return MM_ERR_SHAPE;
  const long long total = (long long)B * Nt * D / 4;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(broadcast_tokens_kernel, dim3(grid), dim3(256), 0, stream, g, (bf16_t*)dy, B, Nt, D, t0, cnt, scale);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// router forward (swin.py:88-92,98-100): probs = softmax(W2 relu(W1 x + b1) + b2), top-k on the
// PROBABILITIES (first max = lowest index on ties).  Every fp32 operation is issued in the order
// of oracle.router_fixed_order (acc = bias; acc = fl(acc + fl(x_j*w_j)), j ascending; no FMA).
// One block per sample; thread j owns hidden unit j; wave 0 then does the softmax and the top-k with wave shuffles.
// ---------------------------------------------------------------------------------------------
#define ROUTER_MAX_E 64
__global__ __launch_bounds__(128) void router_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, float* __restrict__ h_out,
                                                         float* __restrict__ probs, int* __restrict__ idx,
                                                         float* __restrict__ gates, int Dv, int Hd, int E, int k) {
#pragma clang fp contract(off)       // HIP defaults to fp-contract=fast; the oracle order has no FMA
  extern __shared__ float sm[];          // x[Dv] | h[Hd] | logits[E]
  float* sx = sm; float* sh = sm + Dv; float* sl = sh + Hd;
  const int b = blockIdx.x, j = threadIdx.x;
  for (int c = j; c < Dv; c += blockDim.x) sx[c] = x[(long long)b * Dv + c];
  __syncthreads();
  for (int u = j; u < Hd; u += blockDim.x) {
    float acc = b1[u];
    const float* wr = w1 + (long long)u * Dv;
    for (int c = 0; c < Dv; ++c) { const float pr = sx[c] * wr[c]; acc = acc + pr; }
    acc = fmaxf(acc, 0.f);
    sh[u] = acc;
    h_out[(long long)b * Hd + u] = acc;
  }
  __syncthreads();
  if (j < E) {
    float acc = b2[j];
    const float* wr = w2 + (long long)j * Hd;
    for (int c = 0; c < Hd; ++c) { const float pr = sh[c] * wr[c]; acc = acc + pr; }
    sl[j] = acc;
  }
  __syncthreads();
  // softmax + top-k in ONE wave, lane e = expert e: the row maximum and the k arg-max rounds are wave-shuffle butterflies (the
  // arg-max carries (value, index) and prefers the LOWER index on equal values = torch.argmax's first maximum, swin.py:100);
  // the softmax denominator is the one reduction kept in ascending order (s = fl(s + e_q), q = 0..E-1, every lane walking the
  // same sequence through __shfl), so the probabilities - and with them the top-k - stay bit-identical to the oracle's.
  if (j < 64) {
    const int e = j;
    const float l = e < E ? sl[e] : -INFINITY;
    const float m = wave_max(l);
    const float ex = e < E ? expf(l - m) : 0.f;
    float s = 0.f;
    for (int q = 0; q < E; ++q) s = s + __shfl(ex, q, 64);
    const float pr = ex / s;
    if (e < E) probs[(long long)b * E + e] = pr;
    float v = e < E ? pr : -2.f;
    float selp[8], selsum = 0.f;
    for (int t = 0; t < k; ++t) {
      float bv = v; int bi = e;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (e == 0) idx[(long long)b * k + t] = bi;
      selp[t] = bv; selsum += bv;
      if (e == bi) v = -1.f;
    }
    for (int t = 0; t < k; ++t)
      if (e == t) gates[(long long)b * k + t] = (k == 1) ? 1.f : selp[t] / selsum;
  }
}

extern "C" int medmoe_router_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                 float* h, float* probs, int* idx, float* gates, int B, int Dv, int Hd, int E,
                                 int k, hipStream_t stream) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !h || !probs || !idx || !gates) return MM_ERR_ARG;
  if (B <= 0 || Dv <= 0 || Hd <= 0 || E <= 0 || E > ROUTER_MAX_E || E > 128 || k < 1 || k > 8 || k > E) return MM_ERR_SHAPE;
  const size_t sh = (size_t)(Dv + Hd + E) * sizeof(float);
  hipLaunchKernelGGL(router_fwd_kernel, dim3(B), dim3(128), sh, stream, x, w1, b1, w2, b2, h, probs, idx, gates, Dv, Hd, E, k);
  return mm_check_launch();
}

// router backward, per-sample part: CE on the already-softmaxed probabilities
// (medmoe_module.py:235-237) + top-k gate gradients -> dlogits, dh; also the loss value / accuracy.
__global__ __launch_bounds__(128) void router_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ h,
                                                         const float* __restrict__ w2, const int* __restrict__ idx,
                                                         const float* __restrict__ dgates, const int* __restrict__ labels,
                                                         const float* __restrict__ dprobs_ext, float ce_scale,
                                                         float* __restrict__ dlogits,
                                                         float* __restrict__ dh, float* __restrict__ loss_acc,
                                                         int B, int Hd, int E, int k) {
  __shared__ float sdl[ROUTER_MAX_E];
  const int b = blockIdx.x, j = threadIdx.x;
  if (j == 0) {
    const float* p = probs + (long long)b * E;
    float dp[ROUTER_MAX_E];
    float m = p[0];
    for (int e = 1; e < E; ++e) m = fmaxf(m, p[e]);
    float s = 0.f;
    for (int e = 0; e < E; ++e) { dp[e] = __expf(p[e] - m); s += dp[e]; }
    const int lab = labels ? labels[b] : -1;
    int am = 0;
    for (int e = 1; e < E; ++e) if (p[e] > p[am]) am = e;
    for (int e = 0; e < E; ++e) {
      const float q = dp[e] / s;
      if (e == lab) atomicAdd(loss_acc, -__logf(q) / (float)B);
      dp[e] = labels ? ce_scale * (q - (e == lab ? 1.f : 0.f)) : 0.f;
    }
    if (labels && am == lab) atomicAdd(loss_acc + 1, 1.f / (float)B);
    if (dprobs_ext)                      // gradient arriving from outside (autograd path of src/ mirror)
      for (int e = 0; e < E; ++e) dp[e] += dprobs_ext[(long long)b * E + e];
    if (dgates && k > 1) {
      float ss = 0.f;
      for (int t = 0; t < k; ++t) ss += p[idx[b * k + t]];
      float dot = 0.f;
      for (int t = 0; t < k; ++t) dot += dgates[b * k + t] * p[idx[b * k + t]];
      for (int t = 0; t < k; ++t) dp[idx[b * k + t]] += dgates[b * k + t] / ss - dot / (ss * ss);
    }
    float pd = 0.f;
    for (int e = 0; e < E; ++e) pd += p[e] * dp[e];
    for (int e = 0; e < E; ++e) { sdl[e] = p[e] * (dp[e] - pd); dlogits[(long long)b * E + e] = sdl[e]; }
  }
  __syncthreads();
  for (int u = j; u < Hd; u += blockDim.x) {
    float a = 0.f;
    for (int e = 0; e < E; ++e) a += sdl[e] * w2[(long long)e * Hd + u];
    dh[(long long)b * Hd + u] = h[(long long)b * Hd + u] > 0.f ? a : 0.f;
  }
}

extern "C" int medmoe_router_bwd(const float* probs, const float* h, const float* w2, const int* idx,
                                 const float* dgates, const int* labels, const float* dprobs_ext, float ce_scale,
                                 float* dlogits, float* dh, float* loss_acc, int B, int Hd, int E, int k,
                                 hipStream_t stream) {
  if (!probs || !h || !w2 || !idx || !dlogits || !dh || !loss_acc) return MM_ERR_ARG;
  if (B <= 0 || Hd <= 0 || E <= 0 || E > ROUTER_MAX_E || k < 1 || k > 8) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(router_bwd_kernel, dim3(B), dim3(128), 0, stream, probs, h, w2, idx, dgates, labels, dprobs_ext,
                     ce_scale, dlogits, dh, loss_acc, B, Hd, E, k);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// generic small fp32 GEMM with arbitrary strides: C[m,n] = alpha*sum_k A[m,k]*B[k,n] + beta*C
// (router wgrad/dgrad, global-loss similarity + its gradients).  64x64 tile, 4x4 per thread.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                    float* __restrict__ C, int M, int N, int K, long long sam,
                                                    long long sak, long long sbk, long long sbn, long long ldc,
                                                    float alpha, float beta) {
  __shared__ float sA[16][65], sB[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      const int kk = i & 15, r = i >> 4;
      sA[kk][r] = (m0 + r < M && k0 + kk < K) ? A[(long long)(m0 + r) * sam + (long long)(k0 + kk) * sak] : 0.f;
      sB[kk][r] = (n0 + r < N && k0 + kk < K) ? Bm[(long long)(k0 + kk) * sbk + (long long)(n0 + r) * sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sA[kk][ty * 4 + i]; b[i] = sB[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] += a[i] * b[jn];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + jn;
      if (m < M && n < N) {
        float* c = C + (long long)m * ldc + n;
        *c = alpha * acc[i][jn] + (beta != 0.f ? beta * (*c) : 0.f);
      }
    }
}

extern "C" int medmoe_sgemm(const float* A, const float* Bm, float* C, int M, int N, int K, long long sam,
                            long long sak, long long sbk, long long sbn, long long ldc, float alpha, float beta,
                            hipStream_t stream) {
  if (!A || !Bm || !C) return MM_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(sgemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, stream, A, Bm, C, M, N, K, sam,
                     sak, sbk, sbn, ldc, alpha, beta);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// dispatch: stable counting sort of the (sample, choice) pairs by expert + GEMM tile tables: tiles[0 .. max_tiles) are
// 128-row tiles (count in tile_count[0]), tiles[max_tiles .. 2*max_tiles) 256-row tiles (count in tile_count[1]).
// slot s <-> (b,j);  rows of slot s are [s*P, (s+1)*P);  expert e owns rows [row_off[e], row_off[e+1]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dispatch_kernel(const int* __restrict__ idx, int n_items, int E, int P,
                                                       int* __restrict__ slot_of, int* __restrict__ item_of_slot,
                                                       int* __restrict__ expert_of_slot, int* __restrict__ row_off,
                                                       int* __restrict__ tiles, int* __restrict__ tile_count,
                                                       int max_tiles) {
  // One workgroup, all 256 threads (the first form let thread 0 walk every item and write every tile: 308 us at 2048 items / 4704 tiles,
  // on the critical path of the MoE forward).  Stable counting sort: thread t owns the contiguous items [t * per, (t + 1) * per); its
  // per-expert counts are prefix-summed over the threads, then it hands out the slots of its items in ascending order.
  __shared__ unsigned short cnt[256][ROUTER_MAX_E];   // per (thread chunk, expert): items, then the exclusive prefix inside the expert (n_items < 65536: host check)
  __shared__ int off[ROUTER_MAX_E + 1], t128[ROUTER_MAX_E + 1], t256[ROUTER_MAX_E + 1];
  const int tid = threadIdx.x;
  const int per = (n_items + 255) / 256;
  const int lo = min(tid * per, n_items), hi = min(lo + per, n_items);
  for (int e = 0; e < E; ++e) cnt[tid][e] = 0;
  for (int i = lo; i < hi; ++i) cnt[tid][idx[i]] += 1;
  __syncthreads();
  if (tid < E) {                                  // expert tid: exclusive scan of its column over the 256 chunks
    int run = 0;
    for (int t = 0; t < 256; ++t) { const int c = cnt[t][tid]; cnt[t][tid] = (unsigned short)run; run += c; }
    t128[tid + 1] = run;                          // parked: the expert's item count
  }
  __syncthreads();
  if (tid == 0) {
    off[0] = 0;
    for (int e = 0; e < E; ++e) off[e + 1] = off[e] + t128[e + 1];
    t128[0] = 0; t256[0] = 0;
    for (int e = 0; e < E; ++e) {
      const int rows = (off[e + 1] - off[e]) * P;
      t128[e + 1] = t128[e] + (rows + 127) / 128;
      t256[e + 1] = t256[e] + (rows + 255) / 256;
    }
    tile_count[0] = min(t128[E], max_tiles);
    tile_count[1] = min(t256[E], max_tiles);
  }
  __syncthreads();
  if (tid <= E) row_off[tid] = off[tid] * P;
  for (int i = lo; i < hi; ++i) {                 // stable: ascending (b, j) inside an expert
    const int e = idx[i];
    const int s = off[e] + cnt[tid][e]++;
    slot_of[i] = s; item_of_slot[s] = i; expert_of_slot[s] = e;
  }
  // the tile tables: 128-row tiles, and the same rows as 256-row tiles (medmoe_gemm_nt_tiles256) behind the max_tiles 128-row entries
  int* tiles2 = tiles + max_tiles * 4;
  for (int e = 0; e < E; ++e) {
    const int r0 = off[e] * P, r1 = off[e + 1] * P;
    for (int t = tid; t128[e] + t < min(t128[e + 1], max_tiles); t += 256) {
      int* q = tiles + (t128[e] + t) * 4;
      q[0] = e; q[1] = r0 + t * 128; q[2] = r1; q[3] = 0;
    }
    for (int t = tid; t256[e] + t < min(t256[e + 1], max_tiles); t += 256) {
      int* q = tiles2 + (t256[e] + t) * 4;
      q[0] = e; q[1] = r0 + t * 256; q[2] = r1; q[3] = 0;
    }
  }
}

// rowmap[s*P + p] = sample(s)*Nt + 1 + p   (patch tokens of the routed sample; CLS dropped, swin.py:139)
__global__ __launch_bounds__(256) void rowmap_kernel(const int* __restrict__ item_of_slot, int k, int P, int Nt,
                                                     int R, int* __restrict__ rowmap) {
  for (int r = blockIdx.x * 256 + threadIdx.x; r < R; r += gridDim.x * 256) {
    const int s = r / P, p = r - s * P;
    rowmap[r] = (item_of_slot[s] / k) * Nt + 1 + p;
  }
}

extern "C" int medmoe_dispatch(const int* idx, int B, int k, int E, int P, int Nt, int* slot_of, int* item_of_slot,
                               int* expert_of_slot, int* row_off, int* tiles, int* tile_count, int max_tiles,
                               int* rowmap, hipStream_t stream) {
  if (!idx || !slot_of || !item_of_slot || !expert_of_slot || !row_off || !tiles || !tile_count || !rowmap) return MM_ERR_ARG;
  if (B <= 0 || k < 1 || E < 1 || E > ROUTER_MAX_E || P <= 0 || Nt < P + 1 || (long long)B * k >= 65536) return MM_ERR_SHAPE;
  const int R = B * k * P;
  if (max_tiles < (R + 127) / 128 + E) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(dispatch_kernel, dim3(1), dim3(256), 0, stream, idx, B * k, E, P, slot_of, item_of_slot,
                     expert_of_slot, row_off, tiles, tile_count, max_tiles);
  hipLaunchKernelGGL(rowmap_kernel, dim3(min((R + 255) / 256, 2048)), dim3(256), 0, stream, item_of_slot, k, P, Nt, R, rowmap);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// fused scale attention (swin.py:62-80): a_s = w2 . H1_s + b2 ; w = softmax_s(a) ; out = sum_s w_s G_s
// one wave per slot row.  G: [S][R][Do] bf16 (post-ReLU), H1: [S][R][Dh] bf16 (post-ReLU).
// ---------------------------------------------------------------------------------------------
#define SA_S 4
template <int DCH, int HCH>   // Do <= DCH*512, Dh <= HCH*512
__global__ __launch_bounds__(256) void scale_attn_fwd_kernel(const bf16_t* __restrict__ G, const bf16_t* __restrict__ H1,
                                                             const float* __restrict__ w2, const float* __restrict__ b2,
                                                             const int* __restrict__ expert_of_slot, int P,
                                                             bf16_t* __restrict__ out, float* __restrict__ wts, int R,
                                                             int Do, int Dh) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wid; r < R; r += gridDim.x * 4) {
    const int e = expert_of_slot[r / P];
    float a[SA_S];
#pragma unroll
    for (int s = 0; s < SA_S; ++s) {
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < HCH; ++i) {
        const int c = lane + i * 64;
        if (c * 8 < Dh) {
          const uint4 hv = *(const uint4*)(H1 + ((long long)s * R + r) * Dh + c * 8);
          const float4 w0 = *(const float4*)(w2 + (long long)e * Dh + c * 8), w1 = *(const float4*)(w2 + (long long)e * Dh + c * 8 + 4);
          d += __uint_as_float(hv.x << 16) * w0.x + __uint_as_float(hv.x & 0xffff0000u) * w0.y +
               __uint_as_float(hv.y << 16) * w0.z + __uint_as_float(hv.y & 0xffff0000u) * w0.w +
               __uint_as_float(hv.z << 16) * w1.x + __uint_as_float(hv.z & 0xffff0000u) * w1.y +
               __uint_as_float(hv.w << 16) * w1.z + __uint_as_float(hv.w & 0xffff0000u) * w1.w;
        }
      }
      a[s] = wave_sum(d) + b2[e];
    }
    const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < SA_S; ++s) { a[s] = __expf(a[s] - m); den += a[s]; }
#pragma unroll
    for (int s = 0; s < SA_S; ++s) a[s] /= den;
    if (lane < SA_S) wts[(long long)r * SA_S + lane] = a[lane];
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const int c = lane + i * 64;
      if (c * 8 < Do) {
        float o[8] = {};
#pragma unroll
        for (int s = 0; s < SA_S; ++s) {
          const uint4 gv = *(const uint4*)(G + ((long long)s * R + r) * Do + c * 8);
          const uint32_t w[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[2 * q] += a[s] * __uint_as_float(w[q] << 16);
            o[2 * q + 1] += a[s] * __uint_as_float(w[q] & 0xffff0000u);
          }
        }
        uint4 pk;
        pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]); pk.z = pack2bf(o[4], o[5]); pk.w = pack2bf(o[6], o[7]);
        *(uint4*)(out + (long long)r * Do + c * 8) = pk;
      }
    }
  }
}

extern "C" int medmoe_scale_attn_fwd(const void* G, const void* H1, const float* w2, const float* b2,
                                     const int* expert_of_slot, int P, void* out, float* wts, int R, int Do, int Dh,
                                     hipStream_t stream) {
  if (!G || !H1 || !w2 || !b2 || !expert_of_slot || !out || !wts) return MM_ERR_ARG;
  if (R <= 0 || P <= 0 || (R % P) || (Do % 8) || (Dh % 8) || Do > 1024 || Dh > 512) return MM_ERR_SHAPE;
  const int grid = min((R + 3) / 4, 256 * 8);
  hipLaunchKernelGGL((scale_attn_fwd_kernel<2, 1>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)G, (const bf16_t*)H1,
                     w2, b2, expert_of_slot, P, (bf16_t*)out, wts, R, Do, Dh);
  return mm_check_launch();
}

// img_l[b,p,:] = sum_j gate[b,j] * expert_out[slot_of[b,j]*P + p, :]    (swin.py:105-108; k>1 build-defined)
__global__ __launch_bounds__(256) void combine_fwd_kernel(const bf16_t* __restrict__ eo, const int* __restrict__ slot_of,
                                                          const float* __restrict__ gates, bf16_t* __restrict__ img_l,
                                                          int B, int k, int P, int Do) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rows = B * P, nch = Do >> 3;
  for (int r = blockIdx.x * 4 + wid; r < rows; r += gridDim.x * 4) {
    const int b = r / P, p = r - b * P;
    for (int c = lane; c < nch; c += 64) {
      float o[8] = {};
      for (int j = 0; j < k; ++j) {
        const float gt = gates[b * k + j];
        const uint4 v = *(const uint4*)(eo + ((long long)slot_of[b * k + j] * P + p) * Do + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[2 * q] += gt * __uint_as_float(w[q] << 16);
          o[2 * q + 1] += gt * __uint_as_float(w[q] & 0xffff0000u);
        }
      }
      uint4 pk;
      pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]); pk.z = pack2bf(o[4], o[5]); pk.w = pack2bf(o[6], o[7]);
      *(uint4*)(img_l + (long long)r * Do + c * 8) = pk;
    }
  }
}

extern "C" int medmoe_combine_fwd(const void* expert_out, const int* slot_of, const float* gates, void* img_l, int B,
                                  int k, int P, int Do, hipStream_t stream) {
  if (!expert_out || !slot_of || !gates || !img_l) return MM_ERR_ARG;
  if (B <= 0 || k < 1 || P <= 0 || (Do % 8)) return MM_ERR_SHAPE;
  const int grid = min((B * P + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(combine_fwd_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)expert_out, slot_of, gates,
                     (bf16_t*)img_l, B, k, P, Do);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// backward of combine + scale attention, one wave per slot row r = s*P + p:
//   d_out  = gate * (d_img_l[b,p,:] + d_img_g[b,:]/P)
//   dw_s   = <d_out, G_s>;  da_s = w_s (dw_s - sum_t w_t dw_t)
//   dG_s   = w_s * d_out                       (direct part; the H1 path is added by a dgrad GEMM)
//   dH1_s  = da_s * w2 * (H1_s > 0)
//   dw2[e] += sum_s da_s * H1_s ;  db2[e] += sum_s da_s ;  dgate[b,j] += <d_final, expert_out[r]>
// ---------------------------------------------------------------------------------------------
constexpr int SA_WAVES = 16;          // waves per workgroup of scale_attn_bwd_kernel
int g_sa_rows = 0;                    // medmoe_set_option(13, rows per wave): measurement only, 0 = the built-in rule
template <int DCH, int HCH>
__global__ __launch_bounds__(SA_WAVES * 64) void scale_attn_bwd_kernel(const bf16_t* __restrict__ d_img_l, const float* __restrict__ d_img_g,
                                                             const bf16_t* __restrict__ G, const bf16_t* __restrict__ H1,
                                                             const float* __restrict__ wts, const float* __restrict__ w2,
                                                             const bf16_t* __restrict__ expert_out,
                                                             const int* __restrict__ expert_of_slot,
                                                             const int* __restrict__ item_of_slot, const float* __restrict__ gates,
                                                             int k, int P, bf16_t* __restrict__ dG, bf16_t* __restrict__ dH1,
                                                             float* __restrict__ dw2, float* __restrict__ db2,
                                                             float* __restrict__ dgate, int R, int Do, int Dh, int rows_per_wave) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = blockIdx.x * SA_WAVES + wid;
  const int r_begin = wave * rows_per_wave, r_end = min(R, r_begin + rows_per_wave);
  float aw2[HCH][8];
#pragma unroll
  for (int i = 0; i < HCH; ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) aw2[i][q] = 0.f;
  float ab2 = 0.f;
  int cur_e = -1;
  // ab2 is the same in every lane (da comes out of wave sums)
  auto flush = [&]() {
    if (cur_e < 0) return;
#pragma unroll
    for (int i = 0; i < HCH; ++i) {
      const int c = lane + i * 64;
      if (c * 8 < Dh)
#pragma unroll
        for (int q = 0; q < 8; ++q) { atomicAdd(dw2 + (long long)cur_e * Dh + c * 8 + q, aw2[i][q]); aw2[i][q] = 0.f; }
    }
    if (lane == 0) atomicAdd(db2 + cur_e, ab2);
    ab2 = 0.f;
  };
  for (int r = r_begin; r < r_end; ++r) {
    const int s_ = r / P, p = r - s_ * P;
    const int e = expert_of_slot[s_];
    if (e != cur_e) { flush(); cur_e = e; }
    const int item = item_of_slot[s_];
    const int b = item / k;
    const float gt = gates[item];
    // d_final row (kept in registers), dot with the four G rows and with expert_out
    float dfin[DCH][8];
    float dws[SA_S] = {0.f, 0.f, 0.f, 0.f};
    float dg = 0.f;
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const int c = lane + i * 64;
      if (c * 8 < Do) {
        float f[8] = {};
        if (d_img_l) {
          const uint4 v = *(const uint4*)(d_img_l + ((long long)b * P + p) * Do + c * 8);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) { f[2 * q] = __uint_as_float(w[q] << 16); f[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
        }
        if (d_img_g) {
          const float4 g0 = *(const float4*)(d_img_g + (long long)b * Do + c * 8), g1 = *(const float4*)(d_img_g + (long long)b * Do + c * 8 + 4);
          const float inv = 1.f / (float)P;
          f[0] += g0.x * inv; f[1] += g0.y * inv; f[2] += g0.z * inv; f[3] += g0.w * inv;
          f[4] += g1.x * inv; f[5] += g1.y * inv; f[6] += g1.z * inv; f[7] += g1.w * inv;
        }
        if (dgate) {
          const uint4 v = *(const uint4*)(expert_out + (long long)r * Do + c * 8);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) dg += f[2 * q] * __uint_as_float(w[q] << 16) + f[2 * q + 1] * __uint_as_float(w[q] & 0xffff0000u);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) dfin[i][q] = f[q] * gt;
#pragma unroll
        for (int s = 0; s < SA_S; ++s) {
          const uint4 v = *(const uint4*)(G + ((long long)s * R + r) * Do + c * 8);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            dws[s] += dfin[i][2 * q] * __uint_as_float(w[q] << 16) + dfin[i][2 * q + 1] * __uint_as_float(w[q] & 0xffff0000u);
        }
      }
    }
    float w[SA_S], da[SA_S], wd = 0.f;
#pragma unroll
    for (int s = 0; s < SA_S; ++s) { dws[s] = wave_sum(dws[s]); w[s] = wts[(long long)r * SA_S + s]; wd += w[s] * dws[s]; }
#pragma unroll
    for (int s = 0; s < SA_S; ++s) { da[s] = w[s] * (dws[s] - wd); ab2 += da[s]; }
    if (dgate) { dg = wave_sum(dg); if (lane == 0) atomicAdd(dgate + item, dg); }
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const int c = lane + i * 64;
      if (c * 8 < Do)
#pragma unroll
        for (int s = 0; s < SA_S; ++s) {
          uint4 pk;
          pk.x = pack2bf(w[s] * dfin[i][0], w[s] * dfin[i][1]); pk.y = pack2bf(w[s] * dfin[i][2], w[s] * dfin[i][3]);
          pk.z = pack2bf(w[s] * dfin[i][4], w[s] * dfin[i][5]); pk.w = pack2bf(w[s] * dfin[i][6], w[s] * dfin[i][7]);
          *(uint4*)(dG + ((long long)s * R + r) * Do + c * 8) = pk;
        }
    }
#pragma unroll
    for (int i = 0; i < HCH; ++i) {
      const int c = lane + i * 64;
      if (c * 8 < Dh) {
        const float4 w0 = *(const float4*)(w2 + (long long)e * Dh + c * 8), w1 = *(const float4*)(w2 + (long long)e * Dh + c * 8 + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int s = 0; s < SA_S; ++s) {
          const uint4 hv = *(const uint4*)(H1 + ((long long)s * R + r) * Dh + c * 8);
          const uint32_t hw_[4] = {hv.x, hv.y, hv.z, hv.w};
          float hf[8], o[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) { hf[2 * q] = __uint_as_float(hw_[q] << 16); hf[2 * q + 1] = __uint_as_float(hw_[q] & 0xffff0000u); }
#pragma unroll
          for (int q = 0; q < 8; ++q) { o[q] = hf[q] > 0.f ? da[s] * wv[q] : 0.f; aw2[i][q] += da[s] * hf[q]; }
          uint4 pk;
          pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]); pk.z = pack2bf(o[4], o[5]); pk.w = pack2bf(o[6], o[7]);
          *(uint4*)(dH1 + ((long long)s * R + r) * Dh + c * 8) = pk;
        }
      }
    }
  }
  // Final flush.  Every wave adds Dh + 1 floats to the SAME E x Dh addresses: at 16 rows per wave that was 1.2 M atomics onto 3072
  // addresses at batch 128 (serialised at the memory side: a third of the kernel).  The waves of a workgroup own consecutive row ranges,
  // almost always of one expert: sum them in LDS first, one atomic per element and workgroup; mixed workgroups (an expert boundary
  // inside) fall back to per-wave atomics.
  __shared__ float red[SA_WAVES][HCH * 512 + 1];
  __shared__ int red_e[SA_WAVES];
  if (lane == 0) { red_e[wid] = cur_e; red[wid][HCH * 512] = ab2; }
#pragma unroll
  for (int i = 0; i < HCH; ++i)
#pragma unroll
    for (int q = 0; q < 8; ++q) red[wid][(i * 64 + lane) * 8 + q] = aw2[i][q];
  __syncthreads();
  int e0 = -1;
  bool uniform = true;
  for (int w = 0; w < SA_WAVES; ++w) {
    const int ew = red_e[w];
    if (ew < 0) continue;
    if (e0 < 0) e0 = ew; else if (ew != e0) uniform = false;
  }
  if (!uniform) { flush(); return; }
  if (e0 < 0) return;
  for (int c = threadIdx.x; c <= Dh; c += SA_WAVES * 64) {
    const int src = c < Dh ? c : HCH * 512;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SA_WAVES; ++w) v += red[w][src];
    atomicAdd(c < Dh ? dw2 + (long long)e0 * Dh + c : db2 + e0, v);
  }
}

extern "C" int medmoe_scale_attn_bwd(const void* d_img_l, const float* d_img_g, const void* G, const void* H1,
                                     const float* wts, const float* w2, const void* expert_out,
                                     const int* expert_of_slot, const int* item_of_slot, const float* gates, int k,
                                     int P, void* dG, void* dH1, float* dw2, float* db2, float* dgate, int R, int Do,
                                     int Dh, hipStream_t stream) {
  if (!G || !H1 || !wts || !w2 || !expert_of_slot || !item_of_slot || !gates || !dG || !dH1 || !dw2 || !db2) return MM_ERR_ARG;
  if (!d_img_l && !d_img_g) return MM_ERR_ARG;
  if (dgate && !expert_out) return MM_ERR_ARG;
  if (R <= 0 || P <= 0 || (R % P) || (Do % 8) || (Dh % 8) || Do > 1024 || Dh > 512) return MM_ERR_SHAPE;
  // every wave ends with 8 * Dh/8 + 1 atomics into the same E * Dh addresses of dw2 / db2: few long row ranges, but still two
  // rounds of resident waves (256 CUs x 12).  Measured at R = 401408: 16 rows 4.55 ms, 32 3.18, 64 2.89, 96 2.86, 128 3.22, 256 3.30.
  // (before the workgroup-level reduction of the final flush: 16 rows 4.55 ms, 64 2.89, 96 2.86 at R = 401408)
  // with it (tools/bench_scale_attn_bwd.py): R = 50176: 4 rows 342 us, 16 363, 32 584, 96 1423 (757 before); R = 401408: 4 rows 2207 us, 8 2236, 32 2530, 96 3405 (2721 before)
  const int rows_per_wave = g_sa_rows > 0 ? g_sa_rows : max(4, min(8, (R + 49999) / 50000));
  const int waves = (R + rows_per_wave - 1) / rows_per_wave;
  hipLaunchKernelGGL((scale_attn_bwd_kernel<2, 1>), dim3((waves + SA_WAVES - 1) / SA_WAVES), dim3(SA_WAVES * 64), 0, stream, (const bf16_t*)d_img_l,
                     d_img_g, (const bf16_t*)G, (const bf16_t*)H1, wts, w2, (const bf16_t*)expert_out, expert_of_slot,
                     item_of_slot, gates, k, P, (bf16_t*)dG, (bf16_t*)dH1, dw2, db2, dgate, R, Do, Dh, rows_per_wave);
  return mm_check_launch();
}

// dx[b,1+p,:] += sum_j dF[slot_of[b,j]*P + p, :]      (stage-feature gradient into the residual stream)
__global__ __launch_bounds__(256) void stage_grad_add_kernel(const bf16_t* __restrict__ dF, const int* __restrict__ slot_of,
                                                             bf16_t* __restrict__ dx, int B, int k, int P, int Nt, int D) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rows = B * P, nch = D >> 3;
  for (int r = blockIdx.x * 4 + wid; r < rows; r += gridDim.x * 4) {
    const int b = r / P, p = r - b * P;
    bf16_t* dst = dx + ((long long)b * Nt + 1 + p) * D;
    for (int c = lane; c < nch; c += 64) {
      const uint4 v0 = *(const uint4*)(dst + c * 8);
      const uint32_t w0[4] = {v0.x, v0.y, v0.z, v0.w};
      float o[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) { o[2 * q] = __uint_as_float(w0[q] << 16); o[2 * q + 1] = __uint_as_float(w0[q] & 0xffff0000u); }
      for (int j = 0; j < k; ++j) {
        const uint4 v = *(const uint4*)(dF + ((long long)slot_of[b * k + j] * P + p) * D + c * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { o[2 * q] += __uint_as_float(w[q] << 16); o[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u); }
      }
      uint4 pk;
      pk.x = pack2bf(o[0], o[1]); pk.y = pack2bf(o[2], o[3]); pk.z = pack2bf(o[4], o[5]); pk.w = pack2bf(o[6], o[7]);
      *(uint4*)(dst + c * 8) = pk;
    }
  }
}

extern "C" int medmoe_stage_grad_add(const void* dF, const int* slot_of, void* dx, int B, int k, int P, int Nt, int D,
                                     hipStream_t stream) {
  if (!dF || !slot_of || !dx) return MM_ERR_ARG;
  if (B <= 0 || k < 1 || P <= 0 || Nt < P + 1 || (D % 8)) return MM_ERR_SHAPE;
  const int grid = min((B * P + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(stage_grad_add_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dF, slot_of, (bf16_t*)dx, B, k, P, Nt, D);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Pyramid-geometry experts (the reference's Swin stages have 3136 / 784 / 196 / 49 tokens): F.interpolate(size = P,
// mode = 'linear', align_corners = False) along the token axis of the ReLU'd projection (swin.py:42).  For output token j:
// src = max((j + 0.5) * Pin / Pout - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, Pin - 1), y_j = (1 - w) x_i0 + w x_i1, w = src - i0.
// Backward (gather form, no atomics): dx_i = sum_j [i0(j) = i] (1 - w_j) dy_j + [i1(j) = i] w_j dy_j, optionally times ReLU'(aux_i).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_src(int j, float scale, int Pin, int& i0, int& i1, float& w) {
  const float src = fmaxf(((float)j + 0.5f) * scale - 0.5f, 0.f);
  i0 = min((int)src, Pin - 1);
  i1 = min(i0 + 1, Pin - 1);
  w = src - (float)i0;
}

__global__ __launch_bounds__(256) void lerp_tokens_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int n, int Pin, int Pout, int D) {
  const float scale = (float)Pin / (float)Pout;
  const long long total = (long long)n * Pout * (D / 8);
  for (long long z = blockIdx.x * 256LL + threadIdx.x; z < total; z += (long long)gridDim.x * 256) {
    const int c = z % (D / 8);
    const long long r = z / (D / 8);
    const int j = r % Pout, b = r / Pout;
    int i0, i1; float w;
    lerp_src(j, scale, Pin, i0, i1, w);
    const uint4 a = *(const uint4*)(x + ((long long)b * Pin + i0) * D + c * 8), q = *(const uint4*)(x + ((long long)b * Pin + i1) * D + c * 8);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, qw[4] = {q.x, q.y, q.z, q.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = (1.f - w) * __uint_as_float(aw[e] << 16) + w * __uint_as_float(qw[e] << 16);
      const float hi = (1.f - w) * __uint_as_float(aw[e] & 0xffff0000u) + w * __uint_as_float(qw[e] & 0xffff0000u);
      o[e] = pack2bf(lo, hi);
    }
    *(uint4*)(y + ((long long)b * Pout + j) * D + c * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int medmoe_lerp_tokens_fwd(const void* x, void* y, int n, int Pin, int Pout, int D, hipStream_t stream) {
  if (!x || !y) return MM_ERR_ARG;
  if (n <= 0 || Pin <= 0 || Pout <= 0 || D <= 0 || (D % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)n * Pout * (D / 8);
  hipLaunchKernelGGL(lerp_tokens_fwd_kernel, dim3((int)min((total + 255) / 256, (long long)256 * 16)), dim3(256), 0, stream,
                     (const bf16_t*)x, (bf16_t*)y, n, Pin, Pout, D);
  return mm_check_launch();
}

template <bool TWO>
__global__ __launch_bounds__(256) void lerp_tokens_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ dy2,
                                                              const bf16_t* __restrict__ aux,
                                                              bf16_t* __restrict__ dx, int n, int Pin, int Pout, int D) {
  const float scale = (float)Pin / (float)Pout, inv = (float)Pout / (float)Pin;
  const long long total = (long long)n * Pin * (D / 8);
  for (long long z = blockIdx.x * 256LL + threadIdx.x; z < total; z += (long long)gridDim.x * 256) {
    const int c = z % (D / 8);
    const long long r = z / (D / 8);
    const int i = r % Pin, b = r / Pin;
    // outputs j whose source interval touches input i: src(j) in (i - 1, i + 1) (plus the clamped head / tail)
    int jlo = max(0, (int)floorf(((float)i - 1.f + 0.5f) * inv - 0.5f) - 1);
    int jhi = min(Pout - 1, (int)ceilf(((float)i + 1.f + 0.5f) * inv - 0.5f) + 1);
    if (Pin == Pout) { jlo = i; jhi = i; }                // identity interpolation: source j = i with weight 1
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // branch-free body, four loads in flight: at 49 -> 3136 tokens a thread walks ~130 rows 1.5 KB apart, and with a `continue` per row the
    // loop ran one dependent load at a time (100 us per launch for 154 MB)
    const bf16_t* src = dy + (long long)b * Pout * D + c * 8;
    const bf16_t* src2 = TWO ? dy2 + (long long)b * Pout * D + c * 8 : nullptr;
#pragma unroll 4
    for (int j = jlo; j <= jhi; ++j) {
      int i0, i1; float w;
      lerp_src(j, scale, Pin, i0, i1, w);
      const float cw = (i0 == i ? 1.f - w : 0.f) + (i1 == i ? w : 0.f);
      const uint4 v = *(const uint4*)(src + (long long)j * D);
      const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += cw * __uint_as_float(vw[e] << 16); acc[2 * e + 1] += cw * __uint_as_float(vw[e] & 0xffff0000u); }
      if constexpr (TWO) {                                  // the gradient arrives as two summands (stored separately: no read-modify-write GEMM)
        const uint4 u = *(const uint4*)(src2 + (long long)j * D);
        const uint32_t uw[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += cw * __uint_as_float(uw[e] << 16); acc[2 * e + 1] += cw * __uint_as_float(uw[e] & 0xffff0000u); }
      }
    }
    const long long o = ((long long)b * Pin + i) * D + c * 8;
    if (aux) {
      const uint4 a = *(const uint4*)(aux + o);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!(__uint_as_float(aw[e] << 16) > 0.f)) acc[2 * e] = 0.f;
        if (!(__uint_as_float(aw[e] & 0xffff0000u) > 0.f)) acc[2 * e + 1] = 0.f;
      }
    }
    *(uint4*)(dx + o) = make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
  }
}

extern "C" int medmoe_lerp_tokens_bwd(const void* dy, const void* relu_aux, void* dx, int n, int Pin, int Pout, int D, hipStream_t stream) {
  if (!dy || !dx) return MM_ERR_ARG;
  if (n <= 0 || Pin <= 0 || Pout <= 0 || D <= 0 || (D % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)n * Pin * (D / 8);
  hipLaunchKernelGGL(lerp_tokens_bwd_kernel<false>, dim3((int)min((total + 255) / 256, (long long)256 * 16)), dim3(256), 0, stream,
                     (const bf16_t*)dy, (const bf16_t*)nullptr, (const bf16_t*)relu_aux, (bf16_t*)dx, n, Pin, Pout, D);
  return mm_check_launch();
}

// the same for a gradient stored as TWO summands dy + dy2 (the pyramid experts: the scale-attention's direct term and the hidden layer's
// dgrad, kept apart so that the dgrad GEMM needs no residual read); Pin == Pout is the identity interpolation: dx = (dy + dy2) ReLU'(aux)
extern "C" int medmoe_lerp_tokens_bwd2(const void* dy, const void* dy2, const void* relu_aux, void* dx, int n, int Pin, int Pout, int D,
                                       hipStream_t stream) {
  if (!dy || !dy2 || !dx) return MM_ERR_ARG;
  if (n <= 0 || Pin <= 0 || Pout <= 0 || D <= 0 || (D % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)n * Pin * (D / 8);
  hipLaunchKernelGGL(lerp_tokens_bwd_kernel<true>, dim3((int)min((total + 255) / 256, (long long)256 * 16)), dim3(256), 0, stream,
                     (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)relu_aux, (bf16_t*)dx, n, Pin, Pout, D);
  return mm_check_launch();
}
