// Contrastive-loss kernels.
//   ce_strided      : cross-entropy over the rows (or, with swapped strides, the columns) of a
//                     similarity matrix + its gradient  (losses.py:789-794, :1017-1021, :582-584)
//   gloria_global_* : cosine-similarity scaling and its backward chain (losses.py:775-794)
//   local_pair      : the GLoRIA local word<->region attention loss for one (image, caption) pair
//                     per workgroup, forward and (recomputing) backward (losses.py:979-1012,
//                     attention_fn :698-736, cosine_similarity :690-695)
#include "common.h"

// ---------------------------------------------------------------------------------------------
// CE over "rows" of X with generic strides.  row r: x[c] = X[r*rs + c*cs] * xscale.
// loss_acc += w * (lse - x[label]);  dX[r*rs + c*cs] (+)= w * xscale * (softmax - onehot)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_strided_kernel(const float* __restrict__ X, float* __restrict__ dX, int rows,
                                                         int cols, long long rs, long long cs, int label_off,
                                                         float xscale, float w, int accumulate,
                                                         float* __restrict__ loss_acc) {
  __shared__ float red[4];
  __shared__ float bc;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = X + (long long)r * rs;
  float m = -INFINITY;
  for (int c = tid; c < cols; c += 256) m = fmaxf(m, x[(long long)c * cs] * xscale);
  m = wave_max(m);
  if (lane == 0) red[wid] = m;
  __syncthreads();
  if (tid == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  m = bc;
  float s = 0.f;
  for (int c = tid; c < cols; c += 256) s += __expf(x[(long long)c * cs] * xscale - m);
  s = wave_sum(s);
  __syncthreads();
  if (lane == 0) red[wid] = s;
  __syncthreads();
  if (tid == 0) bc = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  s = bc;
  const int lab = label_off + r;
  if (tid == 0 && loss_acc) atomicAdd(loss_acc, w * (m + __logf(s) - x[(long long)lab * cs] * xscale));
  if (dX) {
    float* d = dX + (long long)r * rs;
    for (int c = tid; c < cols; c += 256) {
      const float g = w * xscale * (__expf(x[(long long)c * cs] * xscale - m) / s - (c == lab ? 1.f : 0.f));
      if (accumulate) d[(long long)c * cs] += g; else d[(long long)c * cs] = g;
    }
  }
}

extern "C" int medmoe_ce_strided(const float* X, float* dX, int rows, int cols, long long rs, long long cs,
                                 int label_off, float xscale, float w, int accumulate, float* loss_acc,
                                 hipStream_t stream) {
  if (!X) return MM_ERR_ARG;
  if (rows <= 0 || cols <= 0 || label_off < 0 || label_off + rows > cols) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(ce_strided_kernel, dim3(rows), dim3(256), 0, stream, X, dX, rows, cols, rs, cs, label_off, xscale,
                     w, accumulate, loss_acc);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Soft-GLoRIA head (losses.py:861-883 global, :1180-1208 local): for row r of the similarity matrix, with the frozen text model's
// caption-to-caption scores soft[r][:] deciding the sets P = {c : soft > t1} (positives) and N = {c : soft <= t2} (negatives),
//   loss_r = 1/|P| * sum_{j in P} softXEnt(onehot_0, [x_j, x_N]),  softXEnt = -log_softmax(.)[0] / (1 + |N|)   (losses.py:796-803 divides
// by the LENGTH of the 1-D logits vector), loss = mean_r loss_r.  Same calling convention as ce_strided: x[c] = X[r*rs + c*cs] * xscale,
// loss_acc += w * loss_r, dX (+)= w * xscale * d loss_r / dx.  A row without positives adds nothing (the reference divides by zero there;
// the diagonal soft score is 1, so it does not happen with t1 < 1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block256_sum(float v, float* red, float* bc) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) *bc = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return *bc;
}

__global__ __launch_bounds__(256) void soft_xent_strided_kernel(const float* __restrict__ X, float* __restrict__ dX,
                                                                const float* __restrict__ soft, int rows, int cols, long long rs,
                                                                long long cs, float xscale, float t1, float t2, float w,
                                                                int accumulate, float* __restrict__ loss_acc) {
  __shared__ float red[4];
  __shared__ float bc;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = X + (long long)r * rs;
  const float* sf = soft + (long long)r * cols;
  float m = -INFINITY, np = 0.f, nn = 0.f;
  for (int c = tid; c < cols; c += 256) {
    const float s = sf[c];
    const bool pos = s > t1, neg = s <= t2;
    np += pos ? 1.f : 0.f;
    nn += neg ? 1.f : 0.f;
    if (pos || neg) m = fmaxf(m, x[(long long)c * cs] * xscale);
  }
  m = wave_max(m);
  if (lane == 0) red[wid] = m;
  __syncthreads();
  if (tid == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  m = bc;
  np = block256_sum(np, red, &bc);
  nn = block256_sum(nn, red, &bc);
  float* d = dX ? dX + (long long)r * rs : nullptr;
  if (np == 0.f) {                                      // uniform over the block
    if (d && !accumulate) for (int c = tid; c < cols; c += 256) d[(long long)c * cs] = 0.f;
    return;
  }
  float z = 0.f;
  for (int c = tid; c < cols; c += 256) if (sf[c] <= t2) z += __expf(x[(long long)c * cs] * xscale - m);
  z = block256_sum(z, red, &bc);
  float l = 0.f, wsum = 0.f;
  for (int c = tid; c < cols; c += 256) if (sf[c] > t1) {
    const float xv = x[(long long)c * cs] * xscale, e = __expf(xv - m);
    l += __logf(e + z) + m - xv;
    wsum += 1.f / (e + z);
  }
  l = block256_sum(l, red, &bc);
  wsum = block256_sum(wsum, red, &bc);
  const float cw = w / (np * (1.f + nn));
  if (tid == 0 && loss_acc) atomicAdd(loss_acc, cw * l);
  if (d) {
    for (int c = tid; c < cols; c += 256) {
      const float s = sf[c], e = __expf(x[(long long)c * cs] * xscale - m);
      float g = 0.f;
      if (s > t1) g += e / (e + z) - 1.f;
      if (s <= t2) g += e * wsum;
      g *= cw * xscale;
      if (accumulate) d[(long long)c * cs] += g; else d[(long long)c * cs] = g;
    }
  }
}

extern "C" int medmoe_soft_xent_strided(const float* X, float* dX, const float* soft, int rows, int cols, long long rs, long long cs,
                                        float xscale, float t1, float t2, float w, int accumulate, float* loss_acc,
                                        hipStream_t stream) {
  if (!X || !soft) return MM_ERR_ARG;
  if (rows <= 0 || cols <= 0) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(soft_xent_strided_kernel, dim3(rows), dim3(256), 0, stream, X, dX, soft, rows, cols, rs, cs, xscale, t1, t2, w,
                     accumulate, loss_acc);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// HardNegativeContrastiveLoss head (losses.py:885-927, nmax = 1): for row r of the cosine matrix the hardest negative is the largest
// entry after the diagonal was replaced by its negative (scores - 2 diag(diag), :903); loss_acc += w * relu(hardest + margin - x[r][r]),
// dX (+)= the sub-gradient (+w at the hardest entry, -w on the diagonal; the lowest index wins a tie).  Same stride convention as
// ce_strided: rows (rs, cs = ld, 1) give the image -> caption term (sorted_img, :906), columns the caption -> image term (sorted_cap).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hardneg_strided_kernel(const float* __restrict__ X, float* __restrict__ dX, int rows, int cols,
                                                              long long rs, long long cs, float margin, float w, int accumulate,
                                                              float* __restrict__ loss_acc) {
  __shared__ float redv[4];
  __shared__ int redi[4];
  __shared__ float bv;
  __shared__ int bi;
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* x = X + (long long)r * rs;
  const float diag = x[(long long)r * cs];
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = tid; c < cols; c += 256) {
    const float v = (c == r) ? -diag : x[(long long)c * cs];
    if (v > m) { m = v; mi = c; }                         // ascending c per thread: the first maximum stays
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  if (lane == 0) { redv[wid] = m; redi[wid] = mi; }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < 4; ++q)
      if (redv[q] > m || (redv[q] == m && redi[q] < mi)) { m = redv[q]; mi = redi[q]; }
    bv = m; bi = mi;
  }
  __syncthreads();
  m = bv; mi = bi;
  const float l = m + margin - diag;
  const bool on = l > 0.f;
  if (tid == 0 && loss_acc && on) atomicAdd(loss_acc, w * l);
  if (dX) {
    float* d = dX + (long long)r * rs;
    for (int c = tid; c < cols; c += 256) {
      float g = 0.f;
      if (on) {
        if (c == mi) g += (c == r) ? -w : w;              // the hardest entry (the negated diagonal if nothing beats it)
        if (c == r) g -= w;
      }
      if (accumulate) d[(long long)c * cs] += g; else d[(long long)c * cs] = g;
    }
  }
}

extern "C" int medmoe_hardneg_strided(const float* X, float* dX, int rows, int cols, long long rs, long long cs, float margin, float w,
                                      int accumulate, float* loss_acc, hipStream_t stream) {
  if (!X) return MM_ERR_ARG;
  if (rows <= 0 || cols <= 0 || rows > cols) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(hardneg_strided_kernel, dim3(rows), dim3(256), 0, stream, X, dX, rows, cols, rs, cs, margin, w, accumulate, loss_acc);
  return mm_check_launch();
}

// row L2 norms of a fp32 [rows, D] matrix (one wave per row)
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ n, int rows, int D) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = x[(long long)row * D + c]; s += v * v; }
  s = wave_sum(s);
  if (lane == 0) n[row] = sqrtf(s);
}

extern "C" int medmoe_rownorm(const float* x, float* n, int rows, int D, hipStream_t stream) {
  if (!x || !n || rows <= 0 || D <= 0) return MM_ERR_ARG;
  hipLaunchKernelGGL(rownorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, n, rows, D);
  return mm_check_launch();
}

// S[i][j] = S[i][j] / max(na[i]*nb[j], eps)          (cosine, losses.py:781-785; temp3 applied in CE)
__global__ __launch_bounds__(256) void cos_scale_kernel(float* __restrict__ S, const float* __restrict__ na,
                                                        const float* __restrict__ nb, int M, int N, float eps) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < (long long)M * N; i += (long long)gridDim.x * 256) {
    const int r = i / N, c = i - (long long)r * N;
    S[i] = S[i] / fmaxf(na[r] * nb[c], eps);
  }
}

extern "C" int medmoe_cos_scale(float* S, const float* na, const float* nb, int M, int N, float eps, hipStream_t stream) {
  if (!S || !na || !nb || M <= 0 || N <= 0) return MM_ERR_ARG;
  const int grid = (int)min(((long long)M * N + 255) / 256, (long long)2048);
  hipLaunchKernelGGL(cos_scale_kernel, dim3(grid), dim3(256), 0, stream, S, na, nb, M, N, eps);
  return mm_check_launch();
}

// backward of the cosine scaling: given dC = dL/dcos and cos (both [M,N]):
//   dM[i][j] = dC/(na nb)   (0 where the eps clamp is active)      -> written over dC
//   ca[i] = -sum_j dC*cos / na[i]^2 ;  cb[j] = -sum_i dC*cos / nb[j]^2
// one block per row i; cb accumulated with atomics (M*N is tiny here)
__global__ __launch_bounds__(256) void cos_scale_bwd_kernel(float* __restrict__ dC, const float* __restrict__ C,
                                                            const float* __restrict__ na, const float* __restrict__ nb,
                                                            float* __restrict__ ca, float* __restrict__ cb, int M, int N,
                                                            float eps) {
  __shared__ float red[4];
  const int i = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int j = tid; j < N; j += 256) {
    const long long o = (long long)i * N + j;
    const float den = na[i] * nb[j];
    const bool live = den >= eps;
    const float g = dC[o], t = live ? g * C[o] : 0.f;
    acc += t;
    if (cb) atomicAdd(cb + j, -t / (nb[j] * nb[j]));
    dC[o] = live ? g / den : g / eps;
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) ca[i] = -(red[0] + red[1] + red[2] + red[3]) / (na[i] * na[i]);
}

extern "C" int medmoe_cos_scale_bwd(float* dC, const float* C, const float* na, const float* nb, float* ca, float* cb,
                                    int M, int N, float eps, hipStream_t stream) {
  if (!dC || !C || !na || !nb || !ca || M <= 0 || N <= 0) return MM_ERR_ARG;
  hipLaunchKernelGGL(cos_scale_bwd_kernel, dim3(M), dim3(256), 0, stream, dC, C, na, nb, ca, cb, M, N, eps);
  return mm_check_launch();
}

// dst[r,:] += coef[r] * src[r,:]
__global__ __launch_bounds__(256) void add_rowscaled_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                            const float* __restrict__ coef, int rows, int D) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < (long long)rows * D; i += (long long)gridDim.x * 256)
    dst[i] += coef[i / D] * src[i];
}

extern "C" int medmoe_add_rowscaled(float* dst, const float* src, const float* coef, int rows, int D, hipStream_t stream) {
  if (!dst || !src || !coef || rows <= 0 || D <= 0) return MM_ERR_ARG;
  const int grid = (int)min(((long long)rows * D + 255) / 256, (long long)2048);
  hipLaunchKernelGGL(add_rowscaled_kernel, dim3(grid), dim3(256), 0, stream, dst, src, coef, rows, D);
  return mm_check_launch();
}

// word norms + transposed copy for the local loss: wn[i,t] = ||words[i,t,:]||;
// wT[d][col(i) + t] = words[i,t,d] (zero for t >= T).  Uniform layout (col_of_cap == nullptr): col(i) = i*Tp, every
// caption Tp columns wide, row length Bc*Tp.  Ragged layout: caption i starts at col_of_cap[i] and is tp_of_cap[i]
// columns wide (its length class, a multiple of 16), row length ldw.
__global__ __launch_bounds__(256) void words_prep_kernel(const bf16_t* __restrict__ words, float* __restrict__ wn,
                                                         bf16_t* __restrict__ wT, int Bc, int T, int Tp, int D,
                                                         const int* __restrict__ col_of_cap, const int* __restrict__ tp_of_cap,
                                                         long long ldw) {
  const int lane = threadIdx.x & 63;
  const int rows = Bc * Tp;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
    const int i = r / Tp, t = r - i * Tp;
    const bool has_col = !col_of_cap || t < tp_of_cap[i];
    const long long col = col_of_cap ? (long long)col_of_cap[i] + t : r;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) {
      const bf16_t v = t < T ? words[((long long)i * T + t) * D + c] : (bf16_t)0;
      const float f = bf2f(v);
      s += f * f;
      if (wT && has_col) wT[(long long)c * ldw + col] = v;
    }
    s = wave_sum(s);
    if (lane == 0 && t < T) wn[i * T + t] = sqrtf(s);
  }
}

extern "C" int medmoe_words_prep(const void* words, float* wn, void* wT, int Bc, int T, int Tp, int D, hipStream_t stream) {
  if (!words || !wn || Bc <= 0 || T <= 0 || Tp < T || D <= 0) return MM_ERR_ARG;
  const int grid = min((Bc * Tp + 3) / 4, 2048);
  hipLaunchKernelGGL(words_prep_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)words, wn, (bf16_t*)wT, Bc, T, Tp, D,
                     (const int*)nullptr, (const int*)nullptr, (long long)Bc * Tp);
  return mm_check_launch();
}

// wn[i][t] = |words[i][t][:]| only (wave per word, 16-byte loads): the transposed local loss takes the words row-major and needs no wT
__global__ __launch_bounds__(256) void words_norm_kernel(const bf16_t* __restrict__ words, float* __restrict__ wn, int rows, int D) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
      const uint4 v = *(const uint4*)(words + (long long)r * D + c);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float a = __uint_as_float(w[e] << 16), b = __uint_as_float(w[e] & 0xffff0000u); s += a * a + b * b; }
    }
    s = wave_sum(s);
    if (lane == 0) wn[r] = sqrtf(s);
  }
}

extern "C" int medmoe_words_prep_ragged(const void* words, float* wn, void* wT, int Bc, int T, int Tp, int D,
                                        const int* col_of_cap, const int* tp_of_cap, long long ldw, hipStream_t stream) {
  if (!words || !wn || !col_of_cap || !tp_of_cap || Bc <= 0 || T <= 0 || Tp < T || D <= 0 || ldw <= 0) return MM_ERR_ARG;
  if (!wT && (D % 8) == 0) {            // norms only (wT null): the element-wise transposing store of the general form is 95 % of its time
    const int rows = Bc * T;
    hipLaunchKernelGGL(words_norm_kernel, dim3(min((rows + 3) / 4, 2048)), dim3(256), 0, stream, (const bf16_t*)words, wn, rows, D);
    return mm_check_launch();
  }
  const int grid = min((Bc * Tp + 3) / 4, 2048);
  hipLaunchKernelGGL(words_prep_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)words, wn, (bf16_t*)wT, Bc, T, Tp, D,
                     col_of_cap, tp_of_cap, ldw);
  return mm_check_launch();
}

// dst bf16 [B][HW][D] = src f32 [B][HWp][D] rows < HW   (un-pad + cast of the local-loss ctx gradient)
__global__ __launch_bounds__(256) void unpad_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B,
                                                         int HW, int HWp, int D) {
  const long long total = (long long)B * HW * D / 4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const int col = e % D;
    const long long bh = e / D;
    const int hw = bh % HW, b = bh / HW;
    const float4 v = *(const float4*)(src + ((long long)b * HWp + hw) * D + col);
    uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w);
    *(uint2*)(dst + e) = o;
  }
}

extern "C" int medmoe_unpad_cast(const float* src, void* dst, int B, int HW, int HWp, int D, hipStream_t stream) {
  if (!src || !dst || B <= 0 || HW <= 0 || HWp < HW || (D % 4)) return MM_ERR_ARG;
  const long long total = (long long)B * HW * D / 4;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(unpad_cast_kernel, dim3(grid), dim3(256), 0, stream, src, (bf16_t*)dst, B, HW, HWp, D);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// GLoRIA local loss, one workgroup per (image b, caption i).
//   S   = ctx_b W_i^T                       [HW x T]   (MFMA, K = D)
//   A1  = softmax over words t (row-wise)   ; A = softmax over regions hw of temp1*A1 (column-wise)
//   num_t = sum_hw A S  (= <w_t, wctx_t>)   ; n2_t = a_t^T Gm a_t with Gm = ctx ctx^T (= ||wctx_t||^2)
//   cos_t = num_t / max(|w_t| sqrt(n2_t), eps) ; sim[b,i] = log sum_t exp(temp2 cos_t)
// Backward (BWD): recomputes the above and emits, for the three follow-up GEMMs,
//   dS [B*HWp, Bc*Tp], A (same shape) and U = 2 dn2_t A  (all bf16; padded rows/cols are zero).
// Accumulator tiles have region rows in registers and the word column on the lane, so A is
// directly the B operand of Y = Gm.A (no transpose), via an LDS image [t][tpos(hw)].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lpos(int r) {
  const int o = r & 31;
  return (r & ~31) + (((o & 15) >> 2) << 3) + ((o >> 4) << 2) + (o & 3);
}

template <int NHT, int NTT, bool BWD, bool PRECOMP>
__global__ __launch_bounds__(256) void local_pair_kernel(const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ words,
                                                         const bf16_t* __restrict__ gmp, const float* __restrict__ wnorm,
                                                         const int* __restrict__ cap_lens, const float* __restrict__ gsim,
                                                         float* __restrict__ sim, bf16_t* __restrict__ dS_out,
                                                         bf16_t* __restrict__ A_out, bf16_t* __restrict__ U_out,
                                                         float* __restrict__ att_out, const bf16_t* __restrict__ a1_pre,
                                                         const float* __restrict__ lse_pre, int B, int Bc, int HW, int T, int D,
                                                         float temp1, float temp2, float eps) {
  constexpr int MH = (NHT + 3) / 4;
  constexpr int HWP = NHT * 16, TP = NTT * 16;
  constexpr int KS2 = (NHT + 1) / 2;
  constexpr int TS = KS2 * 64 + 16;                 // bytes per word row of the A image
  constexpr int GW = KS2 * 32;                      // Gm row length in lpos() positions
  constexpr int ROWS = HWP + TP;
  constexpr int STAGE = ROWS * 128;
  constexpr int IMG = TP * TS;
  constexpr int OUTB = HWP * TP * 2;
  constexpr int REGION = (2 * STAGE > IMG + OUTB) ? 2 * STAGE : IMG + OUTB;
  constexpr int NI = (ROWS * 8 + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[REGION + (4 * TP + 8 * TP + 8) * 4];
  float* red = (float*)(smem + REGION);             // [4][TP]
  float* vnum = red + 4 * TP;                       // [TP] each:
  float* vn2 = vnum + TP;
  float* ve = vn2 + TP;
  float* vdnum = ve + TP;
  float* vdn2 = vdnum + TP;
  float* vcinv = vdn2 + TP;
  float* vca = vcinv + TP;
  float* vcos = vca + TP;
  float* scal = vcos + TP;                          // [0] = sum_t e_t

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = pid / Bc, i = pid - b * Bc;
  const int cap = min(cap_lens[i], T);

  // ---------------- GEMM1: S = ctx_b . W_i^T ----------------
  f32x4_t acc[MH][NTT];
  if constexpr (!PRECOMP) {
  const bf16_t* src[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int q = u * 256 + tid;
    const int row = min(q >> 3, ROWS - 1);
    const int c = (q & 7) ^ (row & 7);
    if (row < HWP) src[u] = ctx + ((long long)b * HW + min(row, HW - 1)) * D + c * 8;
    else src[u] = words + ((long long)i * T + min(row - HWP, T - 1)) * D + c * 8;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int q = u * 256 + tid;
      if (q < ROWS * 8)
        __builtin_amdgcn_global_load_lds(GLB_PTR(src[u] + k0), LDS_PTR(smem + buf * STAGE + (u * 256 + wid * 64) * 16), 16, 0, 0);
    }
  };
#pragma unroll
  for (int mh = 0; mh < MH; ++mh)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc[mh][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const char* sb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + g) ^ (fr & 7)) << 4;
      bf16x8_t wf[NTT];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) wf[tt] = *(const bf16x8_t*)(sb + (HWP + tt * 16 + fr) * 128 + coff);
#pragma unroll
      for (int mh = 0; mh < MH; ++mh) {
        const int ht = wid + 4 * mh;
        if (ht < NHT) {
          const bf16x8_t cf = *(const bf16x8_t*)(sb + (ht * 16 + fr) * 128 + coff);
#pragma unroll
          for (int tt = 0; tt < NTT; ++tt)
            acc[mh][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf, wf[tt], acc[mh][tt], 0, 0, 0);
        }
      }
    }
  };
  const int nk = D / 64;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * 64);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  }   // !PRECOMP

  // ---------------- row softmax over words (losses.py:716) ----------------
  bool tmask[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) tmask[tt] = (tt * 16 + fr) < cap;
  float rmax[MH][4], rinv[MH][4];
  if constexpr (PRECOMP) {
    // word-softmax as fp16 log-probabilities and the row log-sum-exp were produced by local_scores: S = logp + lse.
    // Tile [HWP][TP] comes through LDS in 16-B pieces; each lane then picks its accumulator elements.
    char* tl = smem;
    const long long ldp = (long long)Bc * TP;
    for (int z = tid; z < HWP * (TP / 8); z += 256) {
      const int row = z / (TP / 8), ch = z - row * (TP / 8);
      *(uint4*)(tl + row * TP * 2 + ch * 16) = *(const uint4*)(a1_pre + ((long long)b * HWP + row) * ldp + (long long)i * TP + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int mh = 0; mh < MH; ++mh) {
      const int ht = wid + 4 * mh;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int hw = ht * 16 + g * 4 + r;
        const bool ok = ht < NHT && hw < HW;
        const float L = ok ? lse_pre[((long long)b * Bc + i) * HWP + hw] : 0.f;
        rmax[mh][r] = L; rinv[mh][r] = 1.f;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          const float lp = ok ? h2f(*(const uint16_t*)(tl + (min(hw, HWP - 1) * TP + tt * 16 + fr) * 2)) : LOGP_MIN;
          acc[mh][tt][r] = (lp > 0.5f * LOGP_MIN) ? lp + L : -1e20f;     // masked: finite, it is later multiplied by A = 0
        }
      }
    }
    __syncthreads();
  } else {
#pragma unroll
  for (int mh = 0; mh < MH; ++mh)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float m = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) if (tmask[tt]) m = fmaxf(m, acc[mh][tt][r]);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      float s = 0.f;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) if (tmask[tt]) s += __expf(acc[mh][tt][r] - m);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      rmax[mh][r] = m; rinv[mh][r] = 1.f / s;
    }
  }
  auto a1_of = [&](int mh, int tt, int r) -> float {     // softmax over t
    return tmask[tt] ? __expf(acc[mh][tt][r] - rmax[mh][r]) * rinv[mh][r] : 0.f;
  };
  auto hvalid = [&](int mh, int r) -> bool { const int ht = wid + 4 * mh; return ht < NHT && (ht * 16 + g * 4 + r) < HW; };
  // column reduction helper: per-lane partial[tt] -> total over all hw (all waves), result in every lane
  auto col_reduce = [&](float (&part)[NTT], float* dst) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float v = part[tt];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) red[wid * TP + tt * 16 + fr] = v;
    }
    __syncthreads();
    if (tid < TP) dst[tid] = red[tid] + red[TP + tid] + red[2 * TP + tid] + red[3 * TP + tid];
    __syncthreads();
  };

  // ---------------- column softmax over regions (losses.py:724-725) ----------------
  {
    float part[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      part[tt] = 0.f;
#pragma unroll
      for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (hvalid(mh, r) && tmask[tt]) part[tt] += __expf(temp1 * a1_of(mh, tt, r));
    }
    col_reduce(part, vcinv);      // vcinv holds the column SUM for now
  }
  float cinv[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) cinv[tt] = tmask[tt] ? 1.f / vcinv[tt * 16 + fr] : 0.f;
  // A (bf16-rounded, packed) kept in registers + LDS image [t][lpos(hw)] for Y = Gm.A
  uint2 apk[MH][NTT];
  for (int z = tid; z < IMG / 16; z += 256) *(uint4*)(smem + z * 16) = make_uint4(0, 0, 0, 0);
  __syncthreads();
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    const int ht = wid + 4 * mh;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float a[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        a[r] = (hvalid(mh, r) && tmask[tt]) ? __expf(temp1 * a1_of(mh, tt, r)) * cinv[tt] : 0.f;
      apk[mh][tt].x = pack2bf(a[0], a[1]); apk[mh][tt].y = pack2bf(a[2], a[3]);
      if (ht < NHT) *(uint2*)(smem + (tt * 16 + fr) * TS + lpos(ht * 16 + g * 4) * 2) = apk[mh][tt];
    }
  }
  __syncthreads();
  auto a_of = [&](int mh, int tt, int r) -> float {
    const uint32_t w = (r < 2) ? apk[mh][tt].x : apk[mh][tt].y;
    return (r & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
  };

  // ---------------- GEMM2: Y = Gm . A   (Gm columns pre-permuted by lpos) ----------------
  f32x4_t yacc[MH][NTT];
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    const int ht = wid + 4 * mh;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) yacc[mh][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (ht < NHT) {
      const bf16_t* grow = gmp + ((long long)b * HWP + ht * 16 + fr) * GW;
#pragma unroll
      for (int s = 0; s < KS2; ++s) {
        const bf16x8_t gf = __builtin_bit_cast(bf16x8_t, *(const uint4*)(grow + s * 32 + g * 8));
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          const bf16x8_t af = *(const bf16x8_t*)(smem + (tt * 16 + fr) * TS + s * 64 + g * 16);
          yacc[mh][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, af, yacc[mh][tt], 0, 0, 0);
        }
      }
    }
  }
  // ---------------- num_t, n2_t ----------------
  {
    float pn[NTT], p2[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      pn[tt] = 0.f; p2[tt] = 0.f;
#pragma unroll
      for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = a_of(mh, tt, r);
          pn[tt] += a * acc[mh][tt][r];
          p2[tt] += a * yacc[mh][tt][r];
        }
    }
    col_reduce(pn, vnum);
    col_reduce(p2, vn2);
  }
  if (tid < TP) {
    float e = 0.f, c = 0.f;
    if (tid < cap) {
      const float nw = wnorm[i * T + tid];
      const float den = fmaxf(nw * sqrtf(fmaxf(vn2[tid], 0.f)), eps);
      c = vnum[tid] / den;
      e = __expf(temp2 * c);
    }
    ve[tid] = e; vcos[tid] = c;
  }
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int t = 0; t < cap; ++t) s += ve[t];
    scal[0] = s;
    if (sim) sim[(long long)b * Bc + i] = __logf(s);
  }
  if (!BWD) {
    if (att_out && b == i) {         // attention map of the matching pair (losses.py:993-995): [T][HW]
#pragma unroll
      for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int hw = (wid + 4 * mh) * 16 + g * 4 + r, t = tt * 16 + fr;
            if (hvalid(mh, r) && t < T) att_out[((long long)i * T + t) * HW + hw] = a_of(mh, tt, r);
          }
    }
    return;
  }
  __syncthreads();
  // ---------------- backward ----------------
  if (tid < TP) {
    float dn = 0.f, d2 = 0.f;
    if (tid < cap) {
      const float gs = gsim ? gsim[(long long)b * Bc + i] : 1.f;   // null: emit UNSCALED gradients (single-pass mode)
      const float dcos = gs * temp2 * ve[tid] / scal[0];
      const float nw = wnorm[i * T + tid];
      const float n2 = fmaxf(vn2[tid], 0.f);
      const float den = nw * sqrtf(n2);
      if (den >= eps) { dn = dcos / den; d2 = -dcos * vcos[tid] / fmaxf(n2, 1e-30f); }   // d2 = 2*dn2
      else dn = dcos / eps;
    }
    vdnum[tid] = dn; vdn2[tid] = d2;
  }
  __syncthreads();
  float dnum[NTT], dn2x2[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) { dnum[tt] = vdnum[tt * 16 + fr]; dn2x2[tt] = vdn2[tt * 16 + fr]; }
  // cA[t] = sum_hw A * dA,  dA = dnum*S + 2 dn2 * Y
  {
    float part[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      part[tt] = 0.f;
#pragma unroll
      for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          part[tt] += a_of(mh, tt, r) * (dnum[tt] * acc[mh][tt][r] + dn2x2[tt] * yacc[mh][tt][r]);
    }
    col_reduce(part, vca);
  }
  float ca[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) ca[tt] = vca[tt * 16 + fr];
  // dS = dnum*A + A1*(dA1 - rowdot),  dA1 = temp1*A*(dA - cA),  rowdot = sum_t A1*dA1
  char* outb = smem + IMG;
  const long long ldo = (long long)Bc * TP;
  auto copy_out = [&](bf16_t* dst) {
    __syncthreads();
    for (int z = tid; z < HWP * (TP / 8); z += 256) {
      const int row = z / (TP / 8), ch = z - row * (TP / 8);
      *(uint4*)(dst + ((long long)b * HWP + row) * ldo + (long long)i * TP + ch * 8) = *(const uint4*)(outb + row * TP * 2 + ch * 16);
    }
    __syncthreads();
  };
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    const int ht = wid + 4 * mh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a1[NTT], da1[NTT];
      float rd = 0.f;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        a1[tt] = a1_of(mh, tt, r);
        const float a = a_of(mh, tt, r);
        const float dA = dnum[tt] * acc[mh][tt][r] + dn2x2[tt] * yacc[mh][tt][r];
        da1[tt] = temp1 * a * (dA - ca[tt]);
        rd += a1[tt] * da1[tt];
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) rd += __shfl_xor(rd, o, 64);
      if (ht < NHT) {
        const int row = ht * 16 + g * 4 + r;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          const float ds = dnum[tt] * a_of(mh, tt, r) + a1[tt] * (da1[tt] - rd);
          *(bf16_t*)(outb + (row * TP + tt * 16 + fr) * 2) = f2bf(ds);
        }
      }
    }
  }
  copy_out(dS_out);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int mh = 0; mh < MH; ++mh) {
      const int ht = wid + 4 * mh;
      if (ht < NHT)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = a_of(mh, tt, r);
            *(bf16_t*)(outb + ((ht * 16 + g * 4 + r) * TP + tt * 16 + fr) * 2) = f2bf(pass == 0 ? a : a * dn2x2[tt]);
          }
    }
    copy_out(pass == 0 ? A_out : U_out);
  }
}

extern "C" int medmoe_local_pair(const void* ctx, const void* words, const void* gmp, const float* wnorm,
                                 const int* cap_lens, const float* gsim, float* sim, void* dS, void* A, void* U,
                                 float* att, const void* a1_pre, const float* lse_pre, int B, int Bc, int HW, int T,
                                 int D, float temp1, float temp2, float eps, int backward, hipStream_t stream) {
  if (!gmp || !wnorm || !cap_lens) return MM_ERR_ARG;
  if (a1_pre ? !lse_pre : (!ctx || !words)) return MM_ERR_ARG;
  if (backward ? (!dS || !A || !U || (!gsim && !sim)) : !sim) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || T <= 0 || D <= 0 || (D % 64)) return MM_ERR_SHAPE;
  const int nht = (HW + 15) / 16, ntt = (T + 15) / 16;
#define LP2(H_, T_, BW_, PC_)                                                                                        \
  hipLaunchKernelGGL((local_pair_kernel<H_, T_, BW_, PC_>), dim3(B * Bc), dim3(256), 0, stream, (const bf16_t*)ctx,   \
                     (const bf16_t*)words, (const bf16_t*)gmp, wnorm, cap_lens, gsim, sim, (bf16_t*)dS, (bf16_t*)A,   \
                     (bf16_t*)U, att, (const bf16_t*)a1_pre, lse_pre, B, Bc, HW, T, D, temp1, temp2, eps)
#define LP(H_, T_)                                                        \
  do {                                                                    \
    if (backward) { if (a1_pre) LP2(H_, T_, true, true); else LP2(H_, T_, true, false); }   \
    else { if (a1_pre) LP2(H_, T_, false, true); else LP2(H_, T_, false, false); }          \
  } while (0)
  if (nht == 4 && ntt == 1) LP(4, 1);
  else if (nht == 13 && ntt == 2) LP(13, 2);
  else if (nht == 13 && ntt == 5) LP(13, 5);
  else return MM_ERR_SHAPE;
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// local_scores: S = ctx_all . W_all^T as ONE tiled MFMA GEMM over all (image, caption) pairs with the
// word-softmax (losses.py:716) fused in the epilogue: writes A1 = softmax_t(S) (bf16) and the row
// log-sum-exp (fp32), so the per-pair kernel no longer streams ctx and words (443 KB per pair) and
// S = log(A1) + lse is recoverable.  Block tile 128 (regions) x 160 (two captions x 80 words), K-step 64,
// 4 waves = 2 (region halves) x 2 (captions): a wave holds whole softmax rows (80 words) in registers.
// Rows m = b*HW + hw of ctx map to output rows b*HWP + hw.
// ---------------------------------------------------------------------------------------------
template <int NTT>
__global__ __launch_bounds__(256) void local_scores_kernel(const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ words,
                                                           const int* __restrict__ cap_lens, bf16_t* __restrict__ a1_out,
                                                           float* __restrict__ lse_out, int M, int HW, int HWP, int Bc,
                                                           int T, int D, const int* __restrict__ cap_list, int n_cap,
                                                           long long col_base, long long ldp) {
  // cap_list == nullptr: the n_cap = Bc captions in order, caption j at columns j*TP (uniform layout).  Otherwise
  // the n_cap captions of ONE length class (all <= TP words): caption cap_list[j] at columns col_base + j*TP.
  constexpr int TP = NTT * 16;
  constexpr int BNW = 2 * TP;                   // two captions per tile
  constexpr int ROWS = 128 + BNW;
  constexpr int STAGE = ROWS * 128;
  constexpr int NI = (ROWS * 8 + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wc = wid & 1;
  const int fr = lane & 15, g = lane >> 4;
  const int n_tiles_n = (n_cap + 1) / 2;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = id / n_tiles_n, tile_n = id - tile_m * n_tiles_n;
  const int m0 = tile_m * 128;
  const int cap_j = min(tile_n * 2 + wc, n_cap - 1);        // this wave's caption slot (clamped; stores predicated)
  const int cap_i = cap_list ? cap_list[cap_j] : cap_j;
  const bool cap_ok = tile_n * 2 + wc < n_cap;
  const int cap = min(min(cap_lens[cap_i], T), TP);

  const bf16_t* src[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int q = u * 256 + tid;
    const int row = min(q >> 3, ROWS - 1);
    const int c = (q & 7) ^ (row & 7);
    if (row < 128) src[u] = ctx + (long long)min(m0 + row, M - 1) * D + c * 8;
    else {
      const int n = row - 128, cj = min(tile_n * 2 + n / TP, n_cap - 1), t = min(n % TP, T - 1);
      const int ci = cap_list ? cap_list[cj] : cj;
      src[u] = words + ((long long)ci * T + t) * D + c * 8;
    }
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int q = u * 256 + tid;
      if (q < ROWS * 8)
        __builtin_amdgcn_global_load_lds(GLB_PTR(src[u] + k0), LDS_PTR(smem + buf * STAGE + (u * 256 + wid * 64) * 16), 16, 0, 0);
    }
  };
  f32x4_t acc[4][NTT];     // lane: region row = fr (+16 tm), words 4g..4g+3 (+16 tn)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < NTT; ++c) acc[a][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {
    const char* sb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + g) ^ (fr & 7)) << 4;
      bf16x8_t cf[4];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) cf[tm] = *(const bf16x8_t*)(sb + (wm * 64 + tm * 16 + fr) * 128 + coff);
#pragma unroll
      for (int tn = 0; tn < NTT; ++tn) {
        const bf16x8_t wf = *(const bf16x8_t*)(sb + (128 + wc * TP + tn * 16 + fr) * 128 + coff);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, cf[tm], acc[tm][tn], 0, 0, 0);
      }
    }
  };
  const int nk = D / 64;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * 64);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  // word softmax per region row: the row's words are (tn, r) in this lane and the 4 lane groups g
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    float mx = -INFINITY;
#pragma unroll
    for (int tn = 0; tn < NTT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (tn * 16 + g * 4 + r < cap) mx = fmaxf(mx, acc[tm][tn][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sm = 0.f;
#pragma unroll
    for (int tn = 0; tn < NTT; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sm += (tn * 16 + g * 4 + r < cap) ? __expf(acc[tm][tn][r] - mx) : 0.f;
      }
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    const float lse = mx + __logf(sm);
    const int m = m0 + wm * 64 + tm * 16 + fr;
    if (m < M && cap_ok) {
      const long long prow = (long long)(m / HW) * HWP + (m % HW);
      if (g == 0) lse_out[((long long)(m / HW) * Bc + cap_i) * HWP + (m % HW)] = lse;     // [image][caption][region]
#pragma unroll
      for (int tn = 0; tn < NTT; ++tn) {
        float lp[4];                                   // fp16 log-probabilities S - lse (masked words: LOGP_MIN)
#pragma unroll
        for (int r = 0; r < 4; ++r) lp[r] = (tn * 16 + g * 4 + r < cap) ? fmaxf(acc[tm][tn][r] - lse, LOGP_MIN) : LOGP_MIN;
        uint2 o;
        o.x = pack2h(lp[0], lp[1]);
        o.y = pack2h(lp[2], lp[3]);
        *(uint2*)(a1_out + prow * ldp + col_base + (long long)cap_j * TP + tn * 16 + g * 4) = o;
      }
    }
  }
}

extern "C" int medmoe_local_scores(const void* ctx, const void* words, const int* cap_lens, void* a1, float* lse, int B,
                                   int Bc, int HW, int T, int D, hipStream_t stream) {
  if (!ctx || !words || !cap_lens || !a1 || !lse) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || T <= 0 || D <= 0 || (D % 64)) return MM_ERR_SHAPE;
  const int nht = (HW + 15) / 16, ntt = (T + 15) / 16;
  const int M = B * HW, HWP = nht * 16;
  const int grid = ((M + 127) / 128) * ((Bc + 1) / 2);
#define LSC(T_) hipLaunchKernelGGL((local_scores_kernel<T_>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)ctx, \
                                   (const bf16_t*)words, cap_lens, (bf16_t*)a1, lse, M, HW, HWP, Bc, T, D, \
                                   (const int*)nullptr, Bc, 0ll, (long long)Bc * (T_ * 16))
  if (ntt == 1) LSC(1); else if (ntt == 2) LSC(2); else if (ntt == 5) LSC(5); else return MM_ERR_SHAPE;
#undef LSC
  return mm_check_launch();
}

// One caption LENGTH CLASS of the ragged layout: the n_cap captions cap_list[0..n_cap) all have <= 16*ntt words and
// occupy columns col_base + j*16*ntt of rows that are ldp wide.
extern "C" int medmoe_local_scores_ragged(const void* ctx, const void* words, const int* cap_lens, void* a1, float* lse, int B,
                                          int Bc, int HW, int T, int D, const int* cap_list, int n_cap, int ntt,
                                          long long col_base, long long ldp, hipStream_t stream) {
  if (!ctx || !words || !cap_lens || !a1 || !lse || !cap_list) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || T <= 0 || D <= 0 || (D % 64) || n_cap <= 0 || n_cap > Bc) return MM_ERR_SHAPE;
  if (ntt < 1 || ntt > 5 || col_base < 0 || (col_base % 16) || col_base + (long long)n_cap * ntt * 16 > ldp) return MM_ERR_SHAPE;
  if (mm_launch_scores512(ctx, words, cap_lens, a1, lse, B, Bc, HW, T, D, cap_list, n_cap, ntt, col_base, ldp, stream)) return mm_check_launch();
  const int nht = (HW + 15) / 16;
  const int M = B * HW, HWP = nht * 16;
  const int grid = ((M + 127) / 128) * ((n_cap + 1) / 2);
#define LSC(T_) hipLaunchKernelGGL((local_scores_kernel<T_>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)ctx, \
                                   (const bf16_t*)words, cap_lens, (bf16_t*)a1, lse, M, HW, HWP, Bc, T, D, cap_list, n_cap, \
                                   col_base, ldp)
  switch (ntt) { case 1: LSC(1); break; case 2: LSC(2); break; case 3: LSC(3); break; case 4: LSC(4); break; default: LSC(5); break; }
#undef LSC
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// local_pair2: lean per-(image b, caption i) kernel on top of local_scores.  Input tile = fp16 log-probabilities
// lp = S - lse of the word-softmax [HWP x TP] + row LSE; everything after the word-softmax of losses.py:979-1012:
//   A1 = exp(lp) ; A = softmax_hw(temp1*A1) ; num_t = sum A*S (S = lp + lse) ; n2_t = a_t^T Gm a_t ; cos, sim ;
//   and the gradients w.r.t. S (dS), plus A and U = 2 dn2_t A for the Gram-matrix gradient.
// The first version of this kernel was VALU-bound (10k instructions per wave for 140 MFMAs); this one
// keeps ~25 VALU + 2 transcendental ops per element: masks are multiplicative, A1 stays in LDS (its
// tile is overwritten in place by dS, then A, then U for the coalesced copy-out), Y = Gm.A is
// computed twice (MFMA is idle) instead of being held in 80 registers, cA is closed-form, and whole
// 16-word tiles beyond cap_len are skipped (their A1 columns are zero by construction).
// gsim == nullptr: gradients for dL/dsim = 1 (the caller scales the blocks afterwards).
// ---------------------------------------------------------------------------------------------
// -DPAIR_TIMING (tools/pair_timing.hip): shader clocks per phase of local_pair2_kernel, summed over workgroups
#ifdef PAIR_TIMING
__device__ unsigned long long g_pair_timing[16];
#define PAIR_T(...) __VA_ARGS__
#else
#define PAIR_T(...)
#endif
// Workgroup barrier for LDS hand-offs only: __syncthreads() also drains every global load and store in flight
// (s_waitcnt vmcnt(0)), which exposes the latency of the prefetched Gm fragments and of the three tile copy-outs at
// every one of the pair kernel's fourteen barriers.  No thread of this kernel reads global data another thread wrote.
__device__ __forceinline__ void lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int NHT, int NTT, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void local_pair2_kernel(bf16_t* __restrict__ a1_io, const float* __restrict__ lse_pre,
                                                             const bf16_t* __restrict__ gmp, const float* __restrict__ wnorm,
                                                             const int* __restrict__ cap_lens, const float* __restrict__ gsim,
                                                             float* __restrict__ sim, bf16_t* __restrict__ dS_out,
                                                             bf16_t* __restrict__ U_out, float* __restrict__ att_out, int B,
                                                             int Bc, int HW, int T, float temp1, float temp2, float eps,
                                                             const int* __restrict__ cap_list, int n_cap, long long col_base,
                                                             long long ldp) {
  // cap_list == nullptr: uniform layout, n_cap = Bc, caption j's tile at columns j*TP.  Otherwise one length class
  // of the ragged layout (see local_scores_kernel): caption cap_list[j] at columns col_base + j*TP.
  // NW waves per workgroup share the pair's tile: 8 waves x 2 workgroups per CU = four waves per SIMD (the kernel is a chain of
  // fourteen barriers and LDS round trips; with the 4-wave form the SIMDs sat idle between them)
  constexpr int MH = (NHT + NW - 1) / NW;
  constexpr int HWP = NHT * 16, TP = NTT * 16;
  constexpr int KS2 = (NHT + 1) / 2;
  constexpr int TS = KS2 * 64 + 16;
  constexpr int GW = KS2 * 32;
  // tile rows are TP*2 = 32*NTT bytes: the accumulator-shaped accesses below touch rows g*4 + r for the four lane groups g,
  // 4 rows = 128*NTT bytes apart = the same 8 banks four times.  32 bytes of padding after every 4 rows put the four groups
  // on four different bank octets (PMC: SQ_LDS_BANK_CONFLICT -41 %; the kernel's time did not move - the LDS is active 30 % of it).
  constexpr int TILEB = HWP * TP * 2 + (HWP / 4) * 32;
  auto trow = [](int row) { return row * (TP * 2) + (row >> 2) * 32; };
  constexpr int IMG = TP * TS;
  __shared__ __attribute__((aligned(16))) char smem[TILEB + IMG + (NW * 2 * TP + 8 * TP + 8) * 4];
  char* tile = smem;                                   // A1 tile, later dS / A / U staging
  char* img = smem + TILEB;                            // A image [t][lpos(hw)] for Y = Gm.A
  float* red = (float*)(smem + TILEB + IMG);           // [NW][2][TP]
  float* vnum = red + NW * 2 * TP;
  float* vn2 = vnum + TP;
  float* vcs = vn2 + TP;
  float* vdnum = vcs + TP;
  float* vd2 = vdnum + TP;
  float* vca = vd2 + TP;
  float* ve = vca + TP;
  float* vcos = ve + TP;
  float* scal = vcos + TP;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = pid / n_cap, j = pid - b * n_cap;
  const int i = cap_list ? cap_list[j] : j;
  const int cap = max(1, min(min(cap_lens[i], T), TP));
  const int nta = (cap + 15) >> 4;                     // active 16-word tiles (block-uniform)
  const long long tile_off = ((long long)b * HWP) * ldp + col_base + (long long)j * TP;
  bf16_t* gtile = a1_io + tile_off;
  const float c1 = temp1 * 1.44269504088896f;          // exp(temp1*x) = exp2(c1*x)

  // Gm fragments of this wave's first region tile: independent of everything else, issue now
  // Gm fragment prefetch into a second register set only in the 4-wave form (the 8-wave form has 128 registers and four waves
  // per SIMD to hide the load behind)
  constexpr bool GPF = (NW == 4);
  bf16x8_t gf[KS2], gfn[GPF ? KS2 : 1];
  auto load_g = [&](bf16x8_t (&dst)[KS2], int mh) {
    const int ht = min(wid + NW * mh, NHT - 1);
    const bf16_t* grow = gmp + ((long long)b * HWP + ht * 16 + fr) * GW + g * 8;
#pragma unroll
    for (int s = 0; s < KS2; ++s) dst[s] = __builtin_bit_cast(bf16x8_t, *(const uint4*)(grow + s * 32));
  };
  load_g(gf, 0);
  PAIR_T(long long pt[8]; int pti = 0; pt[pti++] = clock64();)

  // ---- phase 0: A1 tile -> LDS (rows >= HW are never written by local_scores: zero them), zero the image
  for (int z = tid; z < HWP * (TP / 8); z += NW * 64) {
    const int row = z / (TP / 8), ch = z - row * (TP / 8);
    const uint4 v = (row < HW) ? *(const uint4*)(gtile + (long long)row * ldp + ch * 8)
                               : make_uint4(LOGP_MIN_BITS2, LOGP_MIN_BITS2, LOGP_MIN_BITS2, LOGP_MIN_BITS2);
    *(uint4*)(tile + trow(row) + ch * 16) = v;
  }
  for (int z = tid; z < IMG / 16; z += NW * 64) *(uint4*)(img + z * 16) = make_uint4(0, 0, 0, 0);
  // only the LAST region tile (ht = NHT-1) can hold rows >= HW; tiles ht >= NHT do not exist (wave-uniform skip)
  float L[MH][4], mlast[4];
#pragma unroll
  for (int mh = 0; mh < MH; ++mh)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ht = wid + NW * mh, hw = ht * 16 + g * 4 + r;
      const bool ok = ht < NHT && hw < HW;
      L[mh][r] = ok ? lse_pre[((long long)b * Bc + i) * HWP + hw] : 0.f;      // [image][caption][region]: the pair's 208 values are contiguous
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) mlast[r] = ((NHT - 1) * 16 + g * 4 + r < HW) ? 1.f : 0.f;
  auto mrow_of = [&](int mh, int r) -> float { return (wid + NW * mh == NHT - 1) ? mlast[r] : 1.f; };
  float mcol[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) mcol[tt] = (tt * 16 + fr < cap) ? 1.f : 0.f;
  lds_sync();
  // the tile holds fp16 log-probabilities lp = S - lse of the word-softmax: a1 = exp(lp), S = lp + lse
  auto lp_at = [&](int mh, int tt, int r) -> float {
    const int hw = min((wid + NW * mh) * 16 + g * 4 + r, HWP - 1);
    return h2f(*(const uint16_t*)(tile + trow(hw) + (tt * 16 + fr) * 2));
  };
  auto a1_at = [&](int mh, int tt, int r) -> float { return __builtin_amdgcn_exp2f(1.44269504088896f * lp_at(mh, tt, r)); };
  // cross-wave column reduction of TWO per-lane partial vectors at once
  auto col_reduce2 = [&](float (&pa)[NTT], float (&pb)[NTT], float* da, float* db) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float va = pa[tt], vb = pb[tt];
      va += __shfl_xor(va, 16, 64); vb += __shfl_xor(vb, 16, 64);
      va += __shfl_xor(va, 32, 64); vb += __shfl_xor(vb, 32, 64);
      if (g == 0) { red[(wid * 2) * TP + tt * 16 + fr] = va; red[(wid * 2 + 1) * TP + tt * 16 + fr] = vb; }
    }
    lds_sync();
    if (tid < TP) {
      float sa_ = 0.f, sb_ = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sa_ += red[(2 * w) * TP + tid]; sb_ += red[(2 * w + 1) * TP + tid]; }
      da[tid] = sa_; db[tid] = sb_;
    }
    lds_sync();
  };

  PAIR_T(pt[pti++] = clock64();)
  // ---- phase 1: column sums of exp(temp1*A1) over the regions (losses.py:724-725)
  float pz[NTT], pcs[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) { pz[tt] = 0.f; pcs[tt] = 0.f; }
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    if (wid + NW * mh < NHT) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
        if (tt < nta)
#pragma unroll
          for (int r = 0; r < 4; ++r) pcs[tt] += __builtin_amdgcn_exp2f(c1 * a1_at(mh, tt, r)) * mrow_of(mh, r);
    }
    __builtin_amdgcn_sched_barrier(0);       // keep the LDS reads of later tiles from being hoisted (register blow-up)
  }
  col_reduce2(pcs, pz, vcs, vca);
  float cinv[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) cinv[tt] = mcol[tt] / fmaxf(vcs[tt * 16 + fr], 1e-30f);

  PAIR_T(pt[pti++] = clock64();)
  // ---- phase 2: A (bf16) -> registers + image ; num partials
  // A (bf16) per (region tile, word tile): registers in the 4-wave form; the 8-wave form (128 registers) re-reads it from the LDS
  // image it was written to
  constexpr bool AREG = (NW == 4) || NTT <= 3;      // classes 1-3 hold A in registers within 128 (measured 0.3-0.6 ms faster than re-reading)
  uint2 apk[AREG ? MH : 1][AREG ? NTT : 1];
  float pn[NTT], p2[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) { pn[tt] = 0.f; p2[tt] = 0.f; }
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    const int ht = wid + NW * mh;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      if constexpr (AREG) apk[mh][tt] = make_uint2(0u, 0u);
      if (tt < nta && ht < NHT) {
        float a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float lp = lp_at(mh, tt, r);
          const float a1 = __builtin_amdgcn_exp2f(1.44269504088896f * lp);
          a[r] = __builtin_amdgcn_exp2f(c1 * a1) * mrow_of(mh, r) * cinv[tt];
          pn[tt] += a[r] * (lp + L[mh][r]);            // S = lp + lse; masked words: a = 0 exactly, S finite
        }
        const uint2 av = make_uint2(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]));
        if constexpr (AREG) apk[mh][tt] = av;
        *(uint2*)(img + (tt * 16 + fr) * TS + lpos(ht * 16 + g * 4) * 2) = av;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  lds_sync();
  auto a_of = [&](int mh, int tt, int r) -> float {
    uint2 av;
    if constexpr (AREG) av = apk[mh][tt];
    else av = (tt < nta && wid + NW * mh < NHT) ? *(const uint2*)(img + (tt * 16 + fr) * TS + lpos((wid + NW * mh) * 16 + g * 4) * 2) : make_uint2(0u, 0u);
    const uint32_t w = (r < 2) ? av.x : av.y;
    return (r & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
  };
  // Y tile(s) of one region tile: Y[hw'][t] = sum_hw Gm[hw'][hw] A[hw][t]
  auto y_tiles = [&](f32x4_t (&y)[NTT], const bf16x8_t (&gfr)[KS2]) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      y[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
      if (tt < nta)
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
          const bf16x8_t af = *(const bf16x8_t*)(img + (tt * 16 + fr) * TS + s * 64 + g * 16);
          y[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gfr[s], af, y[tt], 0, 0, 0);
        }
    }
  };
  PAIR_T(pt[pti++] = clock64();)
  // ---- GEMM2 pass 1: n2 partials
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    if constexpr (GPF) { if (mh + 1 < MH) load_g(gfn, mh + 1); }
    if (wid + NW * mh < NHT) {
      f32x4_t y[NTT];
      y_tiles(y, gf);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) p2[tt] += a_of(mh, tt, r) * y[tt][r];
    }
    if (mh + 1 < MH) {
      if constexpr (GPF) {
#pragma unroll
        for (int s = 0; s < KS2; ++s) gf[s] = gfn[s];
      } else load_g(gf, mh + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  PAIR_T(pt[pti++] = clock64();)
  load_g(gf, 0);                                       // for pass 2, in flight during the scalar phase
  col_reduce2(pn, p2, vnum, vn2);
  if (tid < TP) {
    float e = 0.f, c = 0.f;
    if (tid < cap) {
      const float nw = wnorm[i * T + tid];
      c = vnum[tid] / fmaxf(nw * sqrtf(fmaxf(vn2[tid], 0.f)), eps);
      e = __expf(temp2 * c);
    }
    ve[tid] = e; vcos[tid] = c;
  }
  lds_sync();
  if (tid == 0) {
    float s = 0.f;
    for (int t = 0; t < cap; ++t) s += ve[t];
    scal[0] = s;
    if (sim) sim[(long long)b * Bc + i] = __logf(s);
  }
  if (att_out && b == i) {                             // attention map of the matching pair (losses.py:993-995)
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hw = (wid + NW * mh) * 16 + g * 4 + r, t = tt * 16 + fr;
          if (wid + NW * mh < NHT && hw < HW && t < T) att_out[((long long)i * T + t) * HW + hw] = a_of(mh, tt, r);
        }
  }
  if (!dS_out) return;
  lds_sync();
  if (tid < TP) {
    float dn = 0.f, d2 = 0.f;
    if (tid < cap) {
      const float gs = gsim ? gsim[(long long)b * Bc + i] : 1.f;
      const float dcos = gs * temp2 * ve[tid] / scal[0];
      const float nw = wnorm[i * T + tid];
      const float n2 = fmaxf(vn2[tid], 0.f);
      const float den = nw * sqrtf(n2);
      if (den >= eps) { dn = dcos / den; d2 = -dcos * vcos[tid] / fmaxf(n2, 1e-30f); }   // d2 = 2*dn2
      else dn = dcos / eps;
    }
    vdnum[tid] = dn; vd2[tid] = d2;
    vca[tid] = dn * vnum[tid] + d2 * vn2[tid];         // cA = sum_hw A*dA in closed form
  }
  lds_sync();
  float dnum[NTT], dd2[NTT], ca[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) { dnum[tt] = vdnum[tt * 16 + fr]; dd2[tt] = vd2[tt * 16 + fr]; ca[tt] = vca[tt * 16 + fr]; }

  PAIR_T(pt[pti++] = clock64();)
  // ---- phase 3: GEMM2 pass 2 + dS, written over the A1 tile in place (each wave owns its rows)
#pragma unroll
  for (int mh = 0; mh < MH; ++mh) {
    const int ht = wid + NW * mh;
    if constexpr (GPF) { if (mh + 1 < MH) load_g(gfn, mh + 1); }
    f32x4_t y[NTT];
    if (ht < NHT) y_tiles(y, gf);
    if (ht < NHT)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a1v[NTT], da1[NTT];
      float rd = 0.f;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        a1v[tt] = 0.f; da1[tt] = 0.f;
        if (tt < nta) {
          const float lp = lp_at(mh, tt, r);
          const float a1 = __builtin_amdgcn_exp2f(1.44269504088896f * lp);
          const float S = lp + L[mh][r];
          const float a = a_of(mh, tt, r);
          const float dA = dnum[tt] * S + dd2[tt] * y[tt][r];
          a1v[tt] = a1;
          da1[tt] = temp1 * a * (dA - ca[tt]);
          rd += a1 * da1[tt];
        }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) rd += __shfl_xor(rd, o, 64);
      {
        const int row = ht * 16 + g * 4 + r;
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
          if (tt < nta)
            *(bf16_t*)(tile + trow(row) + (tt * 16 + fr) * 2) = f2bf(dnum[tt] * a_of(mh, tt, r) + a1v[tt] * (da1[tt] - rd));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (mh + 1 < MH) {
      if constexpr (GPF) {
#pragma unroll
        for (int s = 0; s < KS2; ++s) gf[s] = gfn[s];
      } else load_g(gf, mh + 1);
    }
  }
  auto copy_out = [&](bf16_t* dst) {
    lds_sync();
    bf16_t* d = dst + tile_off;
    for (int z = tid; z < HWP * (TP / 8); z += NW * 64) {
      const int row = z / (TP / 8), ch = z - row * (TP / 8);
      // 16-word tiles beyond the caption were skipped above: the LDS tile still holds their log-probability fill there
      *(uint4*)(d + (long long)row * ldp + ch * 8) = (ch < 2 * nta) ? *(const uint4*)(tile + trow(row) + ch * 16) : make_uint4(0, 0, 0, 0);
    }
    lds_sync();
  };
  PAIR_T(pt[pti++] = clock64();)
  copy_out(dS_out);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int mh = 0; mh < MH; ++mh) {
      const int ht = wid + NW * mh;
      if (ht < NHT)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
          if (tt < nta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = a_of(mh, tt, r);
              *(bf16_t*)(tile + trow(ht * 16 + g * 4 + r) + (tt * 16 + fr) * 2) = f2bf(pass == 0 ? a : a * dd2[tt]);
            }
    }
    copy_out(pass == 0 ? a1_io : U_out);
  }
  PAIR_T(pt[pti++] = clock64(); if (tid == 0) { for (int q = 0; q + 1 < pti; ++q) atomicAdd(&g_pair_timing[q], (unsigned long long)(pt[q + 1] - pt[q])); atomicAdd(&g_pair_timing[15], 1ull); })
}

// waves per workgroup of local_pair2: eight (two workgroups per CU = four waves per SIMD) wherever the instantiation fits 128
// registers; measured at B = 1024, 208 regions, classes 1-5: 4.5 / 9.4 / 11.7 / 15.9 / 13.4 ms with four waves -> 4.3 / 8.0 / 9.4 /
// 13.9 / 11.4 ms (classes 4-5 re-read A from its LDS image instead of holding it: with A in registers they spill 21-26 and lose 3 ms)
#define PAIR_NW(H_, T_) (((H_) == 16 && (T_) == 5) ? 4 : 8)
extern "C" int medmoe_local_pair2(void* a1_io, const float* lse_pre, const void* gmp, const float* wnorm,
                                  const int* cap_lens, const float* gsim, float* sim, void* dS, void* U, float* att,
                                  int B, int Bc, int HW, int T, float temp1, float temp2, float eps,
                                  hipStream_t stream) {
  if (!a1_io || !lse_pre || !gmp || !wnorm || !cap_lens) return MM_ERR_ARG;
  if (!sim && !dS) return MM_ERR_ARG;
  if (dS && !U) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || T <= 0) return MM_ERR_SHAPE;
  const int nht = (HW + 15) / 16, ntt = (T + 15) / 16;
#define LP3(H_, T_) hipLaunchKernelGGL((local_pair2_kernel<H_, T_, PAIR_NW(H_, T_)>), dim3(B * Bc), dim3(PAIR_NW(H_, T_) * 64), 0, stream, (bf16_t*)a1_io, \
                                       lse_pre, (const bf16_t*)gmp, wnorm, cap_lens, gsim, sim, (bf16_t*)dS, (bf16_t*)U, \
                                       att, B, Bc, HW, T, temp1, temp2, eps, (const int*)nullptr, Bc, 0ll, \
                                       (long long)Bc * (T_ * 16))
  if (nht == 4 && ntt == 1) LP3(4, 1);
  else if (nht == 13 && ntt == 2) LP3(13, 2);
  else if (nht == 13 && ntt == 5) LP3(13, 5);
  else return MM_ERR_SHAPE;
#undef LP3
  return mm_check_launch();
}

// One caption length class of the ragged layout (see medmoe_local_scores_ragged); one workgroup per (image, class member).
extern "C" int medmoe_local_pair2_ragged(void* a1_io, const float* lse_pre, const void* gmp, const float* wnorm,
                                         const int* cap_lens, const float* gsim, float* sim, void* dS, void* U, int B, int Bc,
                                         int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap,
                                         int ntt, long long col_base, long long ldp, hipStream_t stream) {
  if (!a1_io || !lse_pre || !gmp || !wnorm || !cap_lens || !cap_list) return MM_ERR_ARG;
  if (!sim && !dS) return MM_ERR_ARG;
  if (dS && !U) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || T <= 0 || n_cap <= 0 || n_cap > Bc) return MM_ERR_SHAPE;
  if (ntt < 1 || ntt > 5 || col_base < 0 || (col_base % 16) || col_base + (long long)n_cap * ntt * 16 > ldp) return MM_ERR_SHAPE;
  const int nht = (HW + 15) / 16;
#define LP3(H_, T_) hipLaunchKernelGGL((local_pair2_kernel<H_, T_, PAIR_NW(H_, T_)>), dim3(B * n_cap), dim3(PAIR_NW(H_, T_) * 64), 0, stream, (bf16_t*)a1_io, \
                                       lse_pre, (const bf16_t*)gmp, wnorm, cap_lens, gsim, sim, (bf16_t*)dS, (bf16_t*)U, \
                                       (float*)nullptr, B, Bc, HW, T, temp1, temp2, eps, cap_list, n_cap, col_base, ldp)
  if (nht == 4 && ntt == 1) LP3(4, 1);
  else if (nht == 13 && ntt == 1) LP3(13, 1);
  else if (nht == 13 && ntt == 2) LP3(13, 2);
  else if (nht == 13 && ntt == 3) LP3(13, 3);
  else if (nht == 13 && ntt == 4) LP3(13, 4);
  else if (nht == 13 && ntt == 5) LP3(13, 5);
  else if (nht == 16 && ntt == 1) LP3(16, 1);        // 256 regions: ViT-L/14 at 224
  else if (nht == 16 && ntt == 2) LP3(16, 2);
  else if (nht == 16 && ntt == 3) LP3(16, 3);
  else if (nht == 16 && ntt == 4) LP3(16, 4);
  else if (nht == 16 && ntt == 5) LP3(16, 5);
  else return MM_ERR_SHAPE;
#undef LP3
  return mm_check_launch();
}

// X[(b,hw)][(i,t)] *= g[b][i] for two matrices at once (single-pass local loss: the pair kernel emits
// gradients for dL/dsim = 1, the CE over the sim matrix then supplies the per-pair factor).
// Ragged layout: cap_of_chunk[c] = caption of the 8-column chunk c (-1: padding columns, left alone), ld = row length.
__global__ __launch_bounds__(256) void scale_blocks_kernel(bf16_t* __restrict__ X0, bf16_t* __restrict__ X1,
                                                           const float* __restrict__ g, int B, int Bc, int HWp, int Tp,
                                                           const int* __restrict__ cap_of_chunk, long long ld) {
  const long long chunks_per_row = ld / 8;
  const long long total = (long long)B * HWp * chunks_per_row;
  for (long long z = blockIdx.x * 256LL + threadIdx.x; z < total; z += (long long)gridDim.x * 256) {
    const long long row = z / chunks_per_row, ch = z - row * chunks_per_row;
    const int b = row / HWp, i = cap_of_chunk ? cap_of_chunk[ch] : (int)((ch * 8) / Tp);
    if (i < 0) continue;
    const float f = g[(long long)b * Bc + i];
    bf16_t* ptrs[2] = {X0, X1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      uint4* p = (uint4*)(ptrs[k] + row * ld + ch * 8);
      uint4 v = *p;
      uint32_t* w = (uint32_t*)&v;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w[q] = pack2bf(__uint_as_float(w[q] << 16) * f, __uint_as_float(w[q] & 0xffff0000u) * f);
      *p = v;
    }
  }
}

extern "C" int medmoe_scale_blocks(void* X0, void* X1, const float* g, int B, int Bc, int HWp, int Tp, hipStream_t stream) {
  if (!X0 || !X1 || !g) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HWp <= 0 || Tp <= 0 || (Tp % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)B * HWp * ((long long)Bc * Tp / 8);
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(scale_blocks_kernel, dim3(grid), dim3(256), 0, stream, (bf16_t*)X0, (bf16_t*)X1, g, B, Bc, HWp, Tp,
                     (const int*)nullptr, (long long)Bc * Tp);
  return mm_check_launch();
}

extern "C" int medmoe_scale_blocks_ragged(void* X0, void* X1, const float* g, int B, int Bc, int HWp, const int* cap_of_chunk,
                                          long long ld, hipStream_t stream) {
  if (!X0 || !X1 || !g || !cap_of_chunk) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HWp <= 0 || ld <= 0 || (ld % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)B * HWp * (ld / 8);
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(scale_blocks_kernel, dim3(grid), dim3(256), 0, stream, (bf16_t*)X0, (bf16_t*)X1, g, B, Bc, HWp, 8,
                     cap_of_chunk, ld);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// GENERIC-GEOMETRY local loss (any number of regions, e.g. the 576 of ViT-L/14 at 336 px or the 3136 of the reference's
// Swin stage 0): the per-pair tile of local_pair2 (region rows x words, all in LDS) only exists for 64 / 208 / 256 regions.
// Here the pair work is the reference's own formulation (losses.py:698-736, 985-1012) as grouped GEMMs plus four
// elementwise kernels over the uniform pair matrices [B*HWp, Bc*Tp] (column (i, t) at i*Tp + t):
//   lp   = word-softmax log-probabilities (local_scores), a1 = exp(lp)
//   A    = softmax over regions of temp1*a1                                  local_gen_fwd_a
//   wctx = A_b^T ctx_b                 [Bc*Tp, D] per image                  grouped gemm_tn (fp32)
//   cos, e, sim = log sum_t e                                                local_gen_cos
//   d wctx = dcos (w / (|w||c|) - cos c / |c|^2)                             local_gen_dwctx  (after the CE over sim)
//   dA = ctx_b d wctx_b^T ; d ctx_b += A_b d wctx_b                          two grouped gemm_nt
//   dS = a1 (da1 - sum_t a1 da1), da1 = temp1 A (dA - sum_hw A dA)           local_gen_bwd_s
//   d ctx_b += dS_b W                                                        gemm_nt
// One workgroup per (image, caption); column reductions meet in LDS float atomics.  Not tuned: the geometries that use it
// run 64 pairs per rank (BASELINE configs[3]) where the whole local loss is < 1 % of the step.
// ---------------------------------------------------------------------------------------------
// Eight waves per (image, caption) block.  A wave pass covers 64 >> LP region rows: lane = (row in the pass, 16-byte piece of 8 word columns),
// 1 << LP pieces per row (Tp / 8 rounded up to a power of two: 32 words -> 4 pieces, 16 rows per pass).  The column sums accumulate in
// registers, meet across the rows of a wave by xor shuffles (a fixed order) and across waves in LDS.  (The first form - one row per wave pass,
// 4 bytes per lane, 40 of 64 lanes - ran the reference's 3136-region geometry at a ninth of the HBM rate: 784 dependent iterations per wave.)
#define LG_NW 8
template <int LP>
__global__ __launch_bounds__(64 * LG_NW) void local_gen_fwd_a_kernel(const uint16_t* __restrict__ lp, const int* __restrict__ cap_lens,
                                                                     bf16_t* __restrict__ A, int Bc, int HW, int HWp, int T, int Tp,
                                                                     float temp1, long long ldp) {
  constexpr int PP = 1 << LP, RPW = 64 >> LP;
  __shared__ float cs[LG_NW][80];
  const int b = blockIdx.x / Bc, i = blockIdx.x - b * Bc;
  const int cap = max(1, min(min(cap_lens[i], T), Tp));
  const long long off = ((long long)b * HWp) * ldp + (long long)i * Tp;
  const float c1 = temp1 * 1.44269504088896f;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t0 = (lane & (PP - 1)) * 8, rsub = lane >> LP;
  const bool on = t0 < Tp;
  float msk[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) msk[e] = t0 + e < cap ? 1.f : 0.f;
  auto ex8 = [&](const uint4& w, float (&ev)[8]) {
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ev[2 * q] = msk[2 * q] * __builtin_amdgcn_exp2f(c1 * __builtin_amdgcn_exp2f(1.44269504088896f * h2f((uint16_t)(ww[q] & 0xffff))));
      ev[2 * q + 1] = msk[2 * q + 1] * __builtin_amdgcn_exp2f(c1 * __builtin_amdgcn_exp2f(1.44269504088896f * h2f((uint16_t)(ww[q] >> 16))));
    }
  };
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on)
    for (int hw = wid * RPW + rsub; hw < HW; hw += LG_NW * RPW) {
      float ev[8];
      ex8(*(const uint4*)(lp + off + (long long)hw * ldp + t0), ev);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += ev[e];
    }
#pragma unroll
  for (int m = PP; m < 64; m <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += __shfl_xor(s[e], m);
  if (on && rsub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[wid][t0 + e] = s[e];
  }
  __syncthreads();
  if (!on) return;
  float inv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < LG_NW; ++w) tot += cs[w][t0 + e];
    inv[e] = 1.f / fmaxf(tot, 1e-30f);
  }
  for (int hw = wid * RPW + rsub; hw < HWp; hw += LG_NW * RPW) {
    float ev[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (hw < HW) ex8(*(const uint4*)(lp + off + (long long)hw * ldp + t0), ev);
    uint4 o;
    o.x = pack2bf(ev[0] * inv[0], ev[1] * inv[1]); o.y = pack2bf(ev[2] * inv[2], ev[3] * inv[3]);
    o.z = pack2bf(ev[4] * inv[4], ev[5] * inv[5]); o.w = pack2bf(ev[6] * inv[6], ev[7] * inv[7]);
    *(uint4*)(A + off + (long long)hw * ldp + t0) = o;
  }
}

static inline int lg_pieces_log2(int Tp) { const int pc = Tp / 8; return pc <= 2 ? 1 : pc <= 4 ? 2 : pc <= 8 ? 3 : 4; }

extern "C" int medmoe_local_gen_fwd_a(const void* lp, const int* cap_lens, void* A, int B, int Bc, int HW, int HWp, int T, int Tp,
                                      float temp1, long long ldp, hipStream_t stream) {
  if (!lp || !cap_lens || !A) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || HWp < HW || T <= 0 || Tp < T || Tp > 80 || (Tp % 16) || (ldp % 8) || ldp < (long long)Bc * Tp) return MM_ERR_SHAPE;
#define LG_FWD(L) hipLaunchKernelGGL(local_gen_fwd_a_kernel<L>, dim3(B * Bc), dim3(64 * LG_NW), 0, stream, (const uint16_t*)lp, cap_lens, (bf16_t*)A, Bc, HW, HWp, T, Tp, temp1, ldp)
  switch (lg_pieces_log2(Tp)) { case 1: LG_FWD(1); break; case 2: LG_FWD(2); break; case 3: LG_FWD(3); break; default: LG_FWD(4); break; }
#undef LG_FWD
  return mm_check_launch();
}

// wave per word: n2 = |wctx|^2, num = <w, wctx>; stats[b][col] = {cos, n2, e, 0}; sim[b][i] = log sum_t e; sume[b][i] = sum_t e
__global__ __launch_bounds__(256) void local_gen_cos_kernel(const float* __restrict__ wc, const bf16_t* __restrict__ words,
                                                            const float* __restrict__ wnorm, const int* __restrict__ cap_lens,
                                                            float* __restrict__ sim, float4* __restrict__ stats, float* __restrict__ sume,
                                                            int Bc, int T, int Tp, int D, float temp2, float eps, long long Kp) {
  __shared__ float ve[80];
  const int b = blockIdx.x / Bc, i = blockIdx.x - b * Bc;
  const int cap = max(1, min(min(cap_lens[i], T), Tp));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int t = wid; t < Tp; t += 4) {
    float e = 0.f, c = 0.f, n2 = 0.f;
    if (t < cap) {
      const float* cr = wc + ((long long)b * Kp + (long long)i * Tp + t) * D;
      const bf16_t* wr = words + ((long long)i * T + t) * D;
      float num = 0.f;
      for (int d = lane; d < D; d += 64) { const float v = cr[d]; n2 += v * v; num += v * bf2f(wr[d]); }
      n2 = wave_sum(n2); num = wave_sum(num);
      c = num / fmaxf(wnorm[i * T + t] * sqrtf(n2), eps);
      e = __expf(temp2 * c);
    }
    if (lane == 0) { ve[t] = e; stats[(long long)b * Kp + (long long)i * Tp + t] = make_float4(c, n2, e, 0.f); }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float sm = 0.f;
    for (int t = 0; t < cap; ++t) sm += ve[t];
    sume[(long long)b * Bc + i] = sm;
    sim[(long long)b * Bc + i] = __logf(sm);
  }
}

extern "C" int medmoe_local_gen_cos(const float* wc, const void* words, const float* wnorm, const int* cap_lens, float* sim, float* stats,
                                    float* sume, int B, int Bc, int T, int Tp, int D, float temp2, float eps, long long Kp,
                                    hipStream_t stream) {
  if (!wc || !words || !wnorm || !cap_lens || !sim || !stats || !sume) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || T <= 0 || Tp < T || Tp > 80 || D <= 0 || Kp < (long long)Bc * Tp) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(local_gen_cos_kernel, dim3(B * Bc), dim3(256), 0, stream, wc, (const bf16_t*)words, wnorm, cap_lens, sim,
                     (float4*)stats, sume, Bc, T, Tp, D, temp2, eps, Kp);
  return mm_check_launch();
}

// d wctx[b][col][:] (bf16) from the cosine / log-sum-exp backward (losses.py:690-695, 1005-1010 differentiated)
__global__ __launch_bounds__(256) void local_gen_dwctx_kernel(const float* __restrict__ wc, const bf16_t* __restrict__ words,
                                                              const float* __restrict__ wnorm, const int* __restrict__ cap_lens,
                                                              const float* __restrict__ gsim, const float4* __restrict__ stats,
                                                              const float* __restrict__ sume, bf16_t* __restrict__ dwc, int Bc, int T,
                                                              int Tp, int D, float temp2, float eps, long long Kp) {
  const int b = blockIdx.x / Bc, i = blockIdx.x - b * Bc;
  const int cap = max(1, min(min(cap_lens[i], T), Tp));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float gs = gsim[(long long)b * Bc + i], se = sume[(long long)b * Bc + i];
  for (int t = wid; t < Tp; t += 4) {
    const long long row = (long long)b * Kp + (long long)i * Tp + t;
    bf16_t* dr = dwc + row * D;
    if (t >= cap) { for (int d = lane; d < D; d += 64) dr[d] = 0; continue; }
    const float4 st = stats[row];                       // cos, n2, e
    const float dcos = gs * temp2 * st.z / se;
    const float nw = wnorm[i * T + t];
    const float den = nw * sqrtf(fmaxf(st.y, 0.f));
    float kw, kc;                                       // d wctx = kw * w + kc * wctx
    if (den >= eps) { kw = dcos / den; kc = -dcos * st.x / fmaxf(st.y, 1e-30f); }
    else { kw = dcos / eps; kc = 0.f; }
    const float* cr = wc + row * D;
    const bf16_t* wr = words + ((long long)i * T + t) * D;
    for (int d = lane; d < D; d += 64) dr[d] = f2bf(kw * bf2f(wr[d]) + kc * cr[d]);
  }
}

extern "C" int medmoe_local_gen_dwctx(const float* wc, const void* words, const float* wnorm, const int* cap_lens, const float* gsim,
                                      const float* stats, const float* sume, void* dwc, int B, int Bc, int T, int Tp, int D, float temp2,
                                      float eps, long long Kp, hipStream_t stream) {
  if (!wc || !words || !wnorm || !cap_lens || !gsim || !stats || !sume || !dwc) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || T <= 0 || Tp < T || Tp > 80 || D <= 0 || Kp < (long long)Bc * Tp) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(local_gen_dwctx_kernel, dim3(B * Bc), dim3(256), 0, stream, wc, (const bf16_t*)words, wnorm, cap_lens, gsim,
                     (const float4*)stats, sume, (bf16_t*)dwc, Bc, T, Tp, D, temp2, eps, Kp);
  return mm_check_launch();
}

// dS (written over dA in place): region-softmax backward (column sums over hw), then word-softmax backward (row sums over t).
// The lane layout of local_gen_fwd_a_kernel: the column sums meet by xor shuffles over the rows of a wave and in LDS over the waves, the
// row sum is a shuffle reduction over the 1 << LP pieces of the row.
template <int LP>
__global__ __launch_bounds__(64 * LG_NW) void local_gen_bwd_s_kernel(const uint16_t* __restrict__ lp, const bf16_t* __restrict__ A,
                                                                     bf16_t* __restrict__ dA_io, const int* __restrict__ cap_lens, int Bc, int HW,
                                                                     int HWp, int T, int Tp, float temp1, long long ldp) {
  constexpr int PP = 1 << LP, RPW = 64 >> LP;
  __shared__ float cas[LG_NW][80];
  const int b = blockIdx.x / Bc, i = blockIdx.x - b * Bc;
  const int cap = max(1, min(min(cap_lens[i], T), Tp));
  const long long off = ((long long)b * HWp) * ldp + (long long)i * Tp;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t0 = (lane & (PP - 1)) * 8, rsub = lane >> LP;
  const bool on = t0 < Tp;
  float msk[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) msk[e] = t0 + e < cap ? 1.f : 0.f;
  float c[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on)
    for (int hw = wid * RPW + rsub; hw < HW; hw += LG_NW * RPW) {
      const long long o = off + (long long)hw * ldp + t0;
      const uint4 a = *(const uint4*)(A + o), d = *(const uint4*)(dA_io + o);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[2 * q] += __uint_as_float(aw[q] << 16) * __uint_as_float(dw[q] << 16);
        c[2 * q + 1] += __uint_as_float(aw[q] & 0xffff0000u) * __uint_as_float(dw[q] & 0xffff0000u);
      }
    }
#pragma unroll
  for (int m = PP; m < 64; m <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] += __shfl_xor(c[e], m);
  if (on && rsub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) cas[wid][t0 + e] = c[e];
  }
  __syncthreads();
  float ca[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (on) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int w = 0; w < LG_NW; ++w) ca[e] += cas[w][t0 + e];
  }
  for (int base = wid * RPW; base < HWp; base += LG_NW * RPW) {        // wave-uniform trip count: the row reduction needs every lane of a row
    const int hw = base + rsub;
    const long long o = off + (long long)hw * ldp + t0;
    float a1[8], dd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; dd[e] = 0.f; }
    if (on && hw < HW) {
      const uint4 l = *(const uint4*)(lp + o), a = *(const uint4*)(A + o), d = *(const uint4*)(dA_io + o);
      const uint32_t lw[4] = {l.x, l.y, l.z, l.w}, aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a1[2 * q] = msk[2 * q] * __builtin_amdgcn_exp2f(1.44269504088896f * h2f((uint16_t)(lw[q] & 0xffff)));
        a1[2 * q + 1] = msk[2 * q + 1] * __builtin_amdgcn_exp2f(1.44269504088896f * h2f((uint16_t)(lw[q] >> 16)));
        dd[2 * q] = temp1 * __uint_as_float(aw[q] << 16) * (__uint_as_float(dw[q] << 16) - ca[2 * q]);            // d a1 = temp1 A (dA - sum_hw A dA)
        dd[2 * q + 1] = temp1 * __uint_as_float(aw[q] & 0xffff0000u) * (__uint_as_float(dw[q] & 0xffff0000u) - ca[2 * q + 1]);
      }
    }
    float rd = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) rd += a1[e] * dd[e];
#pragma unroll
    for (int m = 1; m < PP; m <<= 1) rd += __shfl_xor(rd, m);
    if (on && hw < HWp) {                                              // rows >= HW and words >= cap: zeros
      uint4 w;
      w.x = pack2bf(a1[0] * (dd[0] - rd), a1[1] * (dd[1] - rd)); w.y = pack2bf(a1[2] * (dd[2] - rd), a1[3] * (dd[3] - rd));
      w.z = pack2bf(a1[4] * (dd[4] - rd), a1[5] * (dd[5] - rd)); w.w = pack2bf(a1[6] * (dd[6] - rd), a1[7] * (dd[7] - rd));
      *(uint4*)(dA_io + o) = w;
    }
  }
}

extern "C" int medmoe_local_gen_bwd_s(const void* lp, const void* A, void* dA_io, const int* cap_lens, int B, int Bc, int HW, int HWp,
                                      int T, int Tp, float temp1, long long ldp, hipStream_t stream) {
  if (!lp || !A || !dA_io || !cap_lens) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW <= 0 || HWp < HW || T <= 0 || Tp < T || Tp > 80 || (Tp % 16) || (ldp % 8) || ldp < (long long)Bc * Tp) return MM_ERR_SHAPE;
#define LG_BWD(L) hipLaunchKernelGGL(local_gen_bwd_s_kernel<L>, dim3(B * Bc), dim3(64 * LG_NW), 0, stream, (const uint16_t*)lp, (const bf16_t*)A, (bf16_t*)dA_io, cap_lens, Bc, HW, HWp, T, Tp, temp1, ldp)
  switch (lg_pieces_log2(Tp)) { case 1: LG_BWD(1); break; case 2: LG_BWD(2); break; case 3: LG_BWD(3); break; default: LG_BWD(4); break; }
#undef LG_BWD
  return mm_check_launch();
}

// dst bf16 [B][HW][D] = (src + src2) f32 [B][HWp][D] rows < HW
__global__ __launch_bounds__(256) void unpad_cast2_kernel(const float* __restrict__ src, const float* __restrict__ src2,
                                                          bf16_t* __restrict__ dst, int B, int HW, int HWp, int D) {
  const long long total = (long long)B * HW * D / 4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const int col = e % D;
    const long long bh = e / D;
    const int hw = bh % HW, b = bh / HW;
    const long long so = ((long long)b * HWp + hw) * D + col;
    const float4 v = *(const float4*)(src + so), w = *(const float4*)(src2 + so);
    uint2 o; o.x = pack2bf(v.x + w.x, v.y + w.y); o.y = pack2bf(v.z + w.z, v.w + w.w);
    *(uint2*)(dst + e) = o;
  }
}

extern "C" int medmoe_unpad_cast2(const float* src, const float* src2, void* dst, int B, int HW, int HWp, int D, hipStream_t stream) {
  if (!src || !src2 || !dst || B <= 0 || HW <= 0 || HWp < HW || (D % 4)) return MM_ERR_ARG;
  const long long total = (long long)B * HW * D / 4;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(unpad_cast2_kernel, dim3(grid), dim3(256), 0, stream, src, src2, (bf16_t*)dst, B, HW, HWp, D);
  return mm_check_launch();
}

// 1: the tiled pair kernels (local_pair2 / local_pair) exist for this geometry; 0: the generic path above
extern "C" int medmoe_local_fast_path(int HW, int T) {
  const int nht = (HW + 15) / 16, ntt = (T + 15) / 16;
  return ((nht == 4 && ntt == 1) || ((nht == 13 || nht == 16) && ntt >= 1 && ntt <= 5)) ? 1 : 0;
}

extern "C" int medmoe_local_geometry(int HW, int T, int* HWp, int* Tp, int* GW) {
  const int nht = (HW + 15) / 16, ntt = (T + 15) / 16;
  // (4,1), (13,2), (13,5) exist in the uniform-layout kernels too; the ragged path has every class 1..ntt for 13 / 16 region tiles;
  // every other geometry with <= 80 words runs the generic path (medmoe_local_fast_path() == 0; GW is unused there)
  if (HW <= 0 || T <= 0 || ntt > 5) return MM_ERR_SHAPE;
  *HWp = nht * 16; *Tp = ntt * 16; *GW = ((nht + 1) / 2) * 32;
  return MM_OK;
}
