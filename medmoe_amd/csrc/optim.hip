// Optimizer-side kernels (HBM-bound): global grad-norm, clip + Adam on the flat fp32 master
// buffer (torch.optim.Adam semantics: L2 weight decay added to the gradient; reference settings
// configs/model/med-moe_pretraining.yaml:7-11, clip 0.25 configs/experiment/pretraining_medmoe.yaml:23),
// fused bf16 down-cast of the updated weights, batched bf16 transposes (dgrad reads W^T), casts.
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = *(const float4*)(g + i * 4);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

extern "C" int medmoe_sumsq(const float* g, long long n, float* out, hipStream_t stream) {
  if (!g || !out || n <= 0) return MM_ERR_ARG;
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long long)2048);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, stream, g, n, out);
  return mm_check_launch();
}

// Deterministic sum of squares: out[0] = sum g^2 with a FIXED summation order (per-block partials to scratch, the last
// block to arrive adds them in index order).  The atomicAdd form above depends on block arrival order in its last
// bits; with data parallelism every rank would clip with a slightly different coefficient and the replicas' weights
// would drift apart step by step (found with tools/two_rank_gpu.py: reduced gradients bit-identical, weights not).
// scratch: >= 2049 floats, scratch[2048] (an arrival counter) must be 0 on entry and is 0 again on exit.
__global__ __launch_bounds__(256) void sumsq_det_kernel(const float* __restrict__ g, long long n, float* __restrict__ out,
                                                       float* __restrict__ scratch) {
  __shared__ float red[4];
  __shared__ int last;
  float s = 0.f;
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = *(const float4*)(g + i * 4);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __threadfence();
    const unsigned t = atomicAdd((unsigned*)(scratch + 2048), 1u);
    last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float a = 0.f;
  if (threadIdx.x < 64) {
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) a += __builtin_nontemporal_load(scratch + i);
    a = wave_sum(a);
    if (threadIdx.x == 0) { out[0] = a; *(unsigned*)(scratch + 2048) = 0u; }
  }
}

extern "C" int medmoe_sumsq_det(const float* g, long long n, float* out, float* scratch, hipStream_t stream) {
  if (!g || !out || !scratch || n <= 0) return MM_ERR_ARG;
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long long)2048);
  hipLaunchKernelGGL(sumsq_det_kernel, dim3(grid), dim3(256), 0, stream, g, n, out, scratch);
  return mm_check_launch();
}

// clip coefficient = min(1, max_norm / (sqrt(normsq) + 1e-6))  (torch.nn.utils.clip_grad_norm_).
// Update in torch.optim.Adam's operation order (exp_avg.lerp_, exp_avg_sq.mul_().addcmul_(), sqrt / sqrt(bc2) + eps,
// addcdiv_ with step size lr / bc1); the scalars 1-b1, 1-b2, sqrt(bc2), lr/bc1 are formed in double on the host, as torch
// forms them in Python floats (1 - 0.999f in fp32 is off by 1.3e-5 relative: the second moment would inherit that).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ p16, long long n,
                                                   float b2, float omb1, float omb2, float eps, float wd, float step_size,
                                                   float bc2_sqrt, const float* __restrict__ normsq, float max_norm,
                                                   float grad_scale) {
  float coef = grad_scale;
  if (normsq && max_norm > 0.f) {
    const float nrm = sqrtf(*normsq) * grad_scale;
    coef *= fminf(1.f, max_norm / (nrm + 1e-6f));
  }
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pp = *(float4*)(p + i * 4);
    const float4 gg = *(const float4*)(g + i * 4);
    float4 mm = *(float4*)(m + i * 4), vv = *(float4*)(v + i * 4);
    float* P = (float*)&pp; const float* G = (const float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = G[e] * coef + wd * P[e];
      M[e] = M[e] + omb1 * (gr - M[e]);
      V[e] = b2 * V[e] + omb2 * gr * gr;
      const float denom = sqrtf(V[e]) / bc2_sqrt + eps;
      P[e] -= step_size * (M[e] / denom);
    }
    *(float4*)(p + i * 4) = pp; *(float4*)(m + i * 4) = mm; *(float4*)(v + i * 4) = vv;
    if (p16) { uint2 o; o.x = pack2bf(P[0], P[1]); o.y = pack2bf(P[2], P[3]); *(uint2*)(p16 + i * 4) = o; }
  }
}

extern "C" int medmoe_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, long long n, double lr,
                                double beta1, double beta2, double eps, double weight_decay, int step,
                                const float* grad_normsq, float max_norm, float grad_scale, hipStream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return MM_ERR_ARG;
  if (n % 4) return MM_ERR_SHAPE;   // flat buffers are padded to a multiple of 4 by the host
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const int grid = (int)min((n / 4 + 255) / 256, (long long)256 * 8);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n, (float)beta2,
                     (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)(lr / bc1),
                     (float)sqrt(bc2), grad_normsq, max_norm, grad_scale);
  return mm_check_launch();
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long long n) {
  const long long n4 = n >> 2;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = *(const float4*)(s + i * 4);
    uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w);
    *(uint2*)(d + i * 4) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) d[i] = f2bf(s[i]);
}

extern "C" int medmoe_cast_bf16(const float* src, void* dst, long long n, hipStream_t stream) {
  if (!src || !dst || n <= 0) return MM_ERR_ARG;
  const int grid = (int)min((n / 4 + 255) / 256 + 1, (long long)256 * 8);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, stream, src, (bf16_t*)dst, n);
  return mm_check_launch();
}

// batched 2-D transposes: table[i] = {src_off, dst_off, rows, cols} (element offsets into the flat
// bf16 buffers); dst[c][r] = src[r][c].  grid = (n_entries, max 64x64 tiles per entry).
__global__ __launch_bounds__(256) void transpose_many_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const long long* __restrict__ table) {
  // 64 x 64 tile through LDS.  Both global sides move 16 bytes per lane (8 consecutive bf16 of a source row in, 8 consecutive bf16 of a
  // destination row out) whenever the tile is whole and the pitches are multiples of 8; the 2-byte path covers the ragged edges.
  __shared__ bf16_t tile[64][72];
  const long long so = table[blockIdx.x * 4 + 0], doff = table[blockIdx.x * 4 + 1];
  const int rows = (int)table[blockIdx.x * 4 + 2], cols = (int)table[blockIdx.x * 4 + 3];
  const int tc = (cols + 63) / 64, tr = (rows + 63) / 64;
  if ((int)blockIdx.y >= tc * tr) return;
  const int r0 = (blockIdx.y / tc) * 64, c0 = (blockIdx.y % tc) * 64;
  const bool whole = r0 + 64 <= rows && c0 + 64 <= cols && !(rows & 7) && !(cols & 7) && !(so & 7) && !(doff & 7);
  if (whole) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = t + i * 256, r = ch >> 3, c8 = (ch & 7) * 8;
      *(uint4*)&tile[r][c8] = *(const uint4*)(src + so + (long long)(r0 + r) * cols + c0 + c8);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = t + i * 256, c = ch >> 3, r8 = (ch & 7) * 8;       // destination row c0 + c, its columns r0 + r8 .. + 7
      bf16_t v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tile[r8 + q][c];
      *(uint4*)(dst + doff + (long long)(c0 + c) * rows + r0 + r8) = *(const uint4*)v;
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = src[so + (long long)(r0 + r) * cols + c0 + tx];
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < cols && r0 + tx < rows) dst[doff + (long long)(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

extern "C" int medmoe_transpose_many(const void* src, void* dst, const long long* table, int n_entries,
                                     int max_tiles, hipStream_t stream) {
  if (!src || !dst || !table || n_entries <= 0 || max_tiles <= 0) return MM_ERR_ARG;
  hipLaunchKernelGGL(transpose_many_kernel, dim3(n_entries, max_tiles), dim3(256), 0, stream, (const bf16_t*)src,
                     (bf16_t*)dst, table);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Stream ordering helper for hosts that schedule two streams by hand (the weight-gradient GEMMs of a backward run on a second stream
// underneath the dgrad chain): `to` waits for everything enqueued on `from` so far.  One hipEventRecord + hipStreamWaitEvent on an event
// from a small ring (a wait captures the event's state when it is enqueued, so re-recording a ring slot later does not disturb it) - the
// same two calls torch.cuda.Event.record / Stream.wait_event make, without ~20 us of Python per fork at a hundred forks per step.
// ---------------------------------------------------------------------------------------------
extern "C" int medmoe_stream_fork(hipStream_t from, hipStream_t to) {
  static hipEvent_t ring[32];
  static int dev_of[32];
  static bool made[32];
  static unsigned next = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MM_ERR_LAUNCH;
  const unsigned slot = next++ & 31u;
  if (made[slot] && dev_of[slot] != dev) { (void)hipEventDestroy(ring[slot]); made[slot] = false; }
  if (!made[slot]) {
    if (hipEventCreateWithFlags(&ring[slot], hipEventDisableTiming) != hipSuccess) return MM_ERR_LAUNCH;
    made[slot] = true; dev_of[slot] = dev;
  }
  if (hipEventRecord(ring[slot], from) != hipSuccess) return MM_ERR_LAUNCH;
  if (hipStreamWaitEvent(to, ring[slot], 0) != hipSuccess) return MM_ERR_LAUNCH;
  return MM_OK;
}
