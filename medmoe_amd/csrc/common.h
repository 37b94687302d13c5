// Shared device helpers for the MedMoE gfx950 kernels (wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(2))) short short2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define MM_OK 0
#define MM_ERR_ARG (-1)
#define MM_ERR_SHAPE (-2)
#define MM_ERR_LAUNCH (-3)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; NaN stays NaN via the hardware cvt the compiler emits for __bf16
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// two floats -> one dword of two bf16 (lo in bits 0-15), ONE v_cvt_pk_bf16_f32; same rounding as f2bf
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

// fp16 log-probabilities (the local-loss pair tiles, see loss.hip): two floats -> one dword of two halves, round to nearest
// even (v_cvt_f16_f32 x2 + v_pack_b32_f16); LOGP_MIN marks masked words (finite: it is multiplied by exact zeros later).
typedef _Float16 f16x2_hw_t __attribute__((ext_vector_type(2)));
#define LOGP_MIN (-60000.f)
#define LOGP_MIN_BITS2 0xFB53FB53u
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  const f16x2_hw_t v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float h2f(uint16_t bits) { return (float)__builtin_bit_cast(_Float16, bits); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf via Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, below bf16/fp32-activation noise): 1 exp + 1 rcp + 6 FMA
// instead of the ~40-instruction libm erff in every GELU epilogue element.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
  const float y = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// The same two functions on PAIRS (GEMM epilogues): the polynomial runs as packed fp32 math (v_pk_fma_f32 /
// v_pk_mul_f32, two lanes of work per instruction), GELU' takes exp(-x^2/2) once for both the erf tail and the density,
// and GELU is formed as x Phi(x).  Same approximation as gelu_f / dgelu_f (|error| < 1e-6 against them).
// half_tail2(|x|, e) = (1 - erf(|x| / sqrt 2)) / 2 = 1 - Phi(|x|), e = exp(-x^2 / 2): the A&S 7.1.26 polynomial with the
// 1/sqrt2 of the argument and the final 1/2 folded into its constants (13 packed ops + 4 transcendentals per PAIR for
// GELU and GELU' together; the epilogue of a 256x256 tile runs this 32768 times per wave with nothing to overlap it).
__device__ __forceinline__ f32x2_t half_tail2(f32x2_t axx, f32x2_t e) {
  const f32x2_t d = 1.0f + (0.3275911f * 0.70710678118654752f) * axx;
  const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2_t p = (0.5f * 1.061405429f) * t - (0.5f * 1.453152027f);
  p = p * t + (0.5f * 1.421413741f);
  p = p * t - (0.5f * 0.284496736f);
  p = p * t + (0.5f * 0.254829592f);
  return p * t * e;
}
__device__ __forceinline__ f32x2_t gauss2(f32x2_t axx) {                     // exp(-x^2 / 2)
  const f32x2_t a2 = axx * axx * (-0.5f * 1.4426950408889634f);
  return (f32x2_t){__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
}
__device__ __forceinline__ f32x2_t cdf2(f32x2_t x, f32x2_t h) {              // Phi(x) from h = 1 - Phi(|x|)
  const f32x2_t u = 1.0f - h;
  return (f32x2_t){x[0] >= 0.f ? u[0] : h[0], x[1] >= 0.f ? u[1] : h[1]};
}
__device__ __forceinline__ f32x2_t gelu2(f32x2_t x) {                        // x Phi(x)
  const f32x2_t axx = __builtin_elementwise_abs(x);
  return x * cdf2(x, half_tail2(axx, gauss2(axx)));
}
// GELU and its derivative together (the forward FC1 epilogue stores GELU' so the backward epilogue is one multiply)
__device__ __forceinline__ void gelu_and_grad2(f32x2_t x, f32x2_t& gl, f32x2_t& dg) {
  const f32x2_t axx = __builtin_elementwise_abs(x);
  const f32x2_t e = gauss2(axx);
  const f32x2_t cdf = cdf2(x, half_tail2(axx, e));
  gl = x * cdf;
  dg = cdf + x * (0.39894228040143268f * e);                                 // Phi(x) + x phi(x)
}
__device__ __forceinline__ f32x2_t dgelu2(f32x2_t x) {
  const f32x2_t axx = __builtin_elementwise_abs(x);
  const f32x2_t e = gauss2(axx);
  return cdf2(x, half_tail2(axx, e)) + x * (0.39894228040143268f * e);
}

// bijective XCD-aware remap of a 1-D block id (cdna guide T1): blocks that share an XCD
// (id % 8) get a contiguous chunk of tile ids, so neighbouring tiles hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, s = bid >> 3;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + s;
}

// gemm.hip: the ping-pong score kernel of one caption length class (false: shape not taken, use the loss.hip kernel)
bool mm_launch_scores512(const void* ctx, const void* words, const int* cap_lens, void* a1, float* lse, int B, int Bc, int HW, int T,
                         int D, const int* cap_list, int n_cap, int ntt, long long col_base, long long ldp, hipStream_t stream);

static inline int mm_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MM_OK : MM_ERR_LAUNCH;
}
