// GLoRIA local loss, pair stage, TRANSPOSED pair matrices (losses.py:979-1012 after the word-softmax; the same mathematics
// as local_pair2_kernel in loss.hip, which stays for geometries with more than 224 regions).
//
// Input: fp16 LOG2-probabilities lp2 = (S - lse) / ln 2 of the word softmax (a1 = exp2(lp2), S = lp2 ln 2 + lse) from
// medmoe_local_scores_t (gemm.hip) + the fp32 row log-sum-exps.
// Layout: element (row, image b, region hw) of a pair matrix at row*ld + b*bstride + hw, row = row_base + j*TP + t for word t of the
// class's j-th caption, hw < pw (ld = B*pw, bstride = pw: [word rows][image region columns]; ld = pw, bstride = rows*pw: image-major;
// pw = 224 for 196 regions makes every 64-byte wave segment 64-byte aligned: with 208 every second row started mid-burst and the tile
// stores cost 1.34x their bytes in HBM writes).  One WAVE owns one (image b, caption i, 16-word tile tt) unit: lane (fr, g) holds word t = tt*16 + fr
// and the regions hw = 32 s + 8 g + e (s < NS, e < 8) - 16-byte loads and stores of 8 consecutive regions, and exactly the B-operand
// fragment of v_mfma_f32_16x16x32_bf16 for k-step s, so Y = Gm . A needs no LDS image of A.  The Gm rows are permuted when the
// image's Gram matrix is staged into LDS (row tile rt = 2 s' + h, MFMA row m -> region 32 s' + 8 (m >> 2) + 4 h + (m & 3)): the
// accumulator registers of row tile rt then belong to the regions this lane holds (e = 4 h + r of k-step s'), and every
// reduction over the regions is in-lane plus two cross-group shuffles.  Reductions over a caption's WORDS (the word-softmax
// backward and sum_t exp(temp2 cos_t)) cross the 16 lanes of a row (DPP) and, for captions of more than 16 words, the caption's
// waves: those exchange through LDS mailboxes guarded by epoch flags (no workgroup barrier: the sixteen waves of a workgroup
// drift apart and hide each other's load latency; local_pair2 is a lock-step chain of fourteen barriers).
// A workgroup = 16 waves = 16 / NTT captions at a time against ONE image, whose permuted Gram matrix (98 KB for 196 regions)
// it stages once and keeps for a whole list of captions.
//
// Two launches per class: FWD writes sim, A (bf16) and the per-word sums (num, n2); after the cross-entropy over the sim matrix has
// produced gsim = dL/dsim, BWD reads the log-probabilities, A and those sums back and writes dS (over the log-probabilities, in
// place) and U = 2 dn2 A, both already scaled by gsim.
#include "common.h"

struct Pair3Args {
  const uint16_t* lp; bf16_t* dS; bf16_t* A; bf16_t* U;
  const float* lse; const bf16_t* gm; const float* wnorm; float* stats; float* d2; float* dwn;
  const int* cap_lens; const float* gsim; float* sim; float* att; const int* cap_list;
  long long row_base, ld, bstride;
  int n_cap, B, Bc, HW, HWP, T, caps_per_wg, n_chunk, pw;      // pw: region columns stored per (row, image), HW <= pw <= 32 ceil(HW / 32), % 8
  long long stat_rows;
  float temp1, temp2, eps;
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row, result in every lane (bit-identical in all of them: each step adds the same two values)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);     // row_half_mirror
  v += dpp_mov<0x140>(v);     // row_mirror
  return v;
}
// Four independent row sums interleaved (as sc_row16_sum4 of the score GEMM): `v += dpp_mov(v)` on four values is vectorised into packed adds,
// which cannot take a DPP operand, so every step costs a v_mov_b32_dpp, half a packed add and wait states (208 movs + ~100 s_nop per
// unit of the backward launch); v_add_f32_dpp does the step in one instruction, and three other instructions between two uses of a
// register cover the two wait states.  Same additions in the same order: bit-identical.
#define P3_DPP4(CTRL)                                                                                          \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void row16_sum4(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t" P3_DPP4("quad_perm:[1,0,3,2]") P3_DPP4("quad_perm:[2,3,0,1]") P3_DPP4("row_half_mirror") P3_DPP4("row_mirror")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ float grp4_sum(float v) {      // over the four 16-lane groups
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ uint32_t dword_of(const uint4& v, int d) { return d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w; }
// a new SSA name for the same registers: fp32 copies of the packed tiles (56 registers each) must not be carried from one pass of the
// kernel to the next by common-subexpression elimination
__device__ __forceinline__ void opaque(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// Epoch flags and mailboxes in LDS, by LDS byte address with ds_ instructions: a volatile access through a generic pointer compiles
// to flat_load / flat_store + s_waitcnt vmcnt(0), i.e. every poll would wait for the wave's outstanding tile stores.
__device__ __forceinline__ void lds_store_b32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned lds_load_b32(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void flag_publish(unsigned addr, int epoch) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's mailbox writes are in the LDS before the flag
  lds_store_b32(addr, (unsigned)epoch);
}
__device__ __forceinline__ void flag_wait(unsigned addr, int epoch) {
  while ((int)__builtin_amdgcn_readfirstlane(lds_load_b32(addr)) < epoch) __builtin_amdgcn_s_sleep(1);
}

template <int HW, int NTT, bool BWD, bool WN = false>       // WN: the backward launch also writes the word-norm coefficients p.dwn
__global__ __launch_bounds__(1024) void local_pair3_kernel(Pair3Args p) {
  constexpr int NS = (HW + 31) / 32, HWP = ((HW + 15) / 16) * 16;       // HWP: row length of lse
  const int pw = p.pw;
  constexpr int NRT = 2 * NS, GR = NS * 32, CPI = 16 / NTT, TP = NTT * 16;
  constexpr int NRTA = (HW - 32 * (NS - 1) > 4) ? NRT : NRT - 1;       // row tiles with a region < HW (tile rt starts at region 32 (rt >> 1) + 4 (rt & 1))
  constexpr int OFF_L = NRT * NS * 1024, OFF_R = OFF_L + 16 * GR * 4, OFF_E = OFF_R + 2 * 16 * GR * 4, OFF_F = OFF_E + 128;
  __shared__ __attribute__((aligned(16))) char smem[OFF_F + 128];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, g = lane >> 4;
  const int pid = blockIdx.x;
  const int b = pid / p.n_chunk, chunk = pid - b * p.n_chunk;
  float* Lw = (float*)(smem + OFF_L) + wid * GR;
  float* Rbase = (float*)(smem + OFF_R);               // [2 (epoch parity)][16 waves][GR]
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned eboxA = lds0 + OFF_E, flagE = lds0 + OFF_F, flagR = flagE + 64;      // LDS byte addresses
  if (tid < 32) ((int*)(smem + OFF_F))[tid] = 0;
  // ---- the image's Gram matrix, rows permuted, as MFMA A-operand fragments [rt][s][lane] ----
  for (int f = wid; f < NRT * NS; f += 16) {
    const int rt = f / NS, s = f - rt * NS;
    const int row = 32 * (rt >> 1) + 8 * (fr >> 2) + 4 * (rt & 1) + (fr & 3);
    *(uint4*)(smem + f * 1024 + lane * 16) = *(const uint4*)(p.gm + ((long long)b * GR + row) * GR + 32 * s + 8 * g);
  }
  __syncthreads();
  const int grp = wid / NTT, tt = wid - grp * NTT, w0 = grp * NTT;
  if (grp >= CPI) return;                                   // 16 = CPI * NTT + idle waves (NTT 3 and 5)
  const float c1 = p.temp1 * 1.44269504088896f;
  constexpr float LN2 = 0.6931471805599453f;
  const int j_end = min(p.n_cap, (chunk + 1) * p.caps_per_wg);
  constexpr uint32_t hmin = 0xFB53u;                        // fp16 bits of LOGP_MIN
  const char* gfrag = smem + lane * 16;
  // the fragments of ONE row tile (7 x 4 registers); the next tile's are requested right behind the tile's MFMAs, so the LDS latency
  // runs under the tile's elementwise work
  auto load_frags = [&](bf16x8_t (&f)[NS], int rt) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NS; ++s) f[s] = *(const bf16x8_t*)(gfrag + (rt * NS + s) * 1024);
  };
  int epoch = 0;
  for (int j = chunk * p.caps_per_wg + grp; j < j_end; j += CPI) {
    ++epoch;
    const int i = p.cap_list ? p.cap_list[j] : j;
    const int cap = max(1, min(min(p.cap_lens[i], p.T), TP));
    const int t = tt * 16 + fr;
    const float mcol = t < cap ? 1.f : 0.f;
    // everything this unit reads from global memory is requested here: a load issued behind the tile stores below would wait for them
    const float nw = p.wnorm[i * p.T + min(t, p.T - 1)];
    const float gs = (BWD && p.gsim) ? p.gsim[(long long)b * p.Bc + i] : 1.f;
    const float simv = BWD ? p.sim[(long long)b * p.Bc + i] : 0.f;
    float2 st = make_float2(0.f, 0.f);
    if (BWD) st = *(const float2*)(p.stats + ((long long)b * p.stat_rows + p.row_base + (long long)j * TP + tt * 16 + fr) * 2);
    const long long off0 = (p.row_base + (long long)j * TP + t) * p.ld + (long long)b * p.bstride + 8 * g;
    // ---- loads: the unit's log-probabilities (8 regions per k-step) and the row log-sum-exps of the pair ----
    uint4 lpv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int h0 = 32 * s + 8 * g;
      if (32 * s + 31 < HW) lpv[s] = *(const uint4*)(p.lp + off0 + 32 * s);
      else lpv[s] = (h0 < pw) ? *(const uint4*)(p.lp + off0 + 32 * s) : make_uint4(LOGP_MIN_BITS2, LOGP_MIN_BITS2, LOGP_MIN_BITS2, LOGP_MIN_BITS2);
      if (32 * s + 31 >= HW) {                              // regions >= HW are never written by the score kernel
        auto fix = [&](uint32_t w, int d) -> uint32_t {
          if (h0 + 2 * d >= HW) w = (w & 0xffff0000u) | hmin;
          if (h0 + 2 * d + 1 >= HW) w = (w & 0x0000ffffu) | (hmin << 16);
          return w;
        };
        lpv[s] = make_uint4(fix(lpv[s].x, 0), fix(lpv[s].y, 1), fix(lpv[s].z, 2), fix(lpv[s].w, 3));
      }
    }
    if (lane * 4 < GR) {
      const int h0 = lane * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h0 < HWP) v = *(const float4*)(p.lse + ((long long)b * p.Bc + i) * HWP + h0);
      if (h0 >= HW) v.x = 0.f;
      if (h0 + 1 >= HW) v.y = 0.f;
      if (h0 + 2 >= HW) v.z = 0.f;
      if (h0 + 3 >= HW) v.w = 0.f;
      *(float4*)(Lw + h0) = v;
    }
    auto lp_of = [&](int s, int e) -> float {
      const uint32_t w = dword_of(lpv[s], e >> 1);
      return h2f((uint16_t)((e & 1) ? (w >> 16) : (w & 0xffffu)));
    };
    uint4 af[NS];
    float num, n2;
    if constexpr (BWD) {                                    // the forward launch's A (bf16): the MFMA operand and the elementwise a
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = (32 * s + 31 < HW || 32 * s + 8 * g < pw) ? *(const uint4*)(p.A + off0 + 32 * s) : make_uint4(0u, 0u, 0u, 0u);
    }
    auto a_of = [&](int s, int e) -> float {
      const uint32_t w = dword_of(af[s], e >> 1);
      return (e & 1) ? bf_hi(w) : bf_lo(w);
    };
    bf16x8_t fg[NS];
    // Y tile rt from the fragments in fg, then fg <- the fragments of tile `next` (-1: none)
    auto y_tile = [&](int next) __attribute__((always_inline)) -> f32x4_t {
      f32x4_t y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[s], __builtin_bit_cast(bf16x8_t, af[s]), y, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (next >= 0) load_frags(fg, next);
      __builtin_amdgcn_sched_barrier(0);
      return y;
    };
    // the same without look-ahead (backward launch: two packed tiles + a fragment set + the tile's temporaries do not fit 128 registers, and a
    // spilled register's reload waits for every tile store in flight): the seven reads of a row tile are issued together and waited once
    auto y_tile_now = [&](int rt) __attribute__((always_inline)) -> f32x4_t {
      bf16x8_t f[NS];
      load_frags(f, rt);
      f32x4_t y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s], __builtin_bit_cast(bf16x8_t, af[s]), y, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      return y;
    };
    if constexpr (!BWD) {
      // ---- phase 1: e1 = exp(temp1 a1), its sum over the regions (losses.py:724-725) and the unnormalised sum e1 S ----
      float e1[NS][8];
      float cs = 0.f, un = 0.f;
  #pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float4 La = *(const float4*)(Lw + 32 * s + 8 * g), Lb = *(const float4*)(Lw + 32 * s + 8 * g + 4);
        const float Ls[8] = {La.x, La.y, La.z, La.w, Lb.x, Lb.y, Lb.z, Lb.w};
  #pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float lp = lp_of(s, e);
          const float a1 = __builtin_amdgcn_exp2f(lp);
          float x = __builtin_amdgcn_exp2f(c1 * a1);
          if (32 * s + 24 + e >= HW) x = (32 * s + 8 * g + e < HW) ? x : 0.f;
          e1[s][e] = x;
          cs += x;
          un += x * fmaf(lp, LN2, Ls[e]);                       // S = lp ln 2 + lse; masked regions: x = 0 exactly, S finite
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      cs = grp4_sum(cs);
      un = grp4_sum(un);
      const float cinv = mcol / fmaxf(cs, 1e-30f);
      num = un * cinv;                                        // sum_hw A S
      // ---- phase 2: A = e1 / sum (bf16, the MFMA operand) ----
  #pragma unroll
      for (int s = 0; s < NS; ++s) {
        float a[8];
  #pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = e1[s][e] * cinv;
        const uint4 av = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(a[4], a[5]), pack2bf(a[6], a[7]));
        af[s] = av;
        if (32 * s + 31 < HW || 32 * s + 8 * g < pw) *(uint4*)(p.A + off0 + 32 * s) = av;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- n2 = a^T Gm a ----
  #pragma unroll
      for (int s = 0; s < NS; ++s) { opaque(lpv[s]); opaque(af[s]); }
      load_frags(fg, 0);
      n2 = 0.f;
  #pragma unroll
      for (int rt = 0; rt < NRTA; ++rt) {
        const f32x4_t y = y_tile(rt + 1 < NRTA ? rt + 1 : -1);
  #pragma unroll
        for (int r = 0; r < 4; ++r) n2 += a_of(rt >> 1, 4 * (rt & 1) + r) * y[r];
        __builtin_amdgcn_sched_barrier(0);
      }
      n2 = grp4_sum(n2);
      // per-word statistics for the backward launch ([image][row] pairs: 16 lanes = 128 contiguous bytes)
      if (g == 0) *(float2*)(p.stats + ((long long)b * p.stat_rows + p.row_base + (long long)j * TP + t) * 2) = make_float2(num, n2);
    } else {
      num = st.x; n2 = st.y;
    }
    // ---- per-word cosine, sum over the caption's words ----
    const float n2c = fmaxf(n2, 0.f);
    const float den = nw * sqrtf(n2c);
    const float cosv = num / fmaxf(den, p.eps);
    const float ev = (t < cap) ? __expf(p.temp2 * cosv) : 0.f;
    float se;
    if constexpr (!BWD) {
      se = row16_sum(ev);
      if (NTT > 1) {
        // two mailboxes by epoch parity: there is no second exchange between two of these, so a wave may publish epoch k+1 while
        // a partner still reads epoch k (never k+2: that needs the partner's k+1)
        const unsigned eb = eboxA + (epoch & 1) * 64;
        lds_store_b32(eb + wid * 4, __float_as_uint(se));       // every lane holds the same value
        flag_publish(flagE + wid * 4, epoch);
        se = 0.f;
#pragma unroll
        for (int q = 0; q < NTT; ++q) {
          flag_wait(flagE + (w0 + q) * 4, epoch);
          se += __uint_as_float(lds_load_b32(eb + (w0 + q) * 4));
        }
      }
    } else se = __expf(simv);                               // sim = log sum_t e_t from the forward launch
    if (!BWD) {
      if (tt == 0 && lane == 0) p.sim[(long long)b * p.Bc + i] = __logf(se);
      if (p.att && b == i && t < p.T) {                     // attention map of the matching pair (losses.py:993-995)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int hw = 32 * s + 8 * g + e;
            if (hw < HW) p.att[((long long)i * p.T + t) * HW + hw] = a_of(s, e);
          }
      }
      continue;
    }
    // ---- gradient coefficients of this lane's word ----
    float dn = 0.f, d2 = 0.f;
    {
      const float dcos = gs * p.temp2 * ev / se;
      if (den >= p.eps) { dn = dcos / den; d2 = -dcos * cosv / fmaxf(n2c, 1e-30f); }      // d2 = 2 dL/dn2
      else dn = dcos / p.eps;
      // the word embedding enters the cosine through S (-> dS) and through its own norm: d cos / d ||w|| = -cos / ||w|| (losses.py:690-695;
      // nothing when the denominator sits on its eps clamp), i.e. d loss / d w_t gets -dcos cos / ||w||^2 times w_t from this image
      if (WN && g == 0)
        p.dwn[(long long)b * p.stat_rows + p.row_base + (long long)j * TP + t] = (den >= p.eps) ? -dcos * cosv / (nw * nw) : 0.f;
    }
    const float ca = dn * num + d2 * n2;                    // sum_hw A dA in closed form
    // da1 = temp1 a (dA - ca), dA = dn S + d2 y:  da1 = a (k1 S + k2 y - k3)
    const float k1 = p.temp1 * dn, k2 = p.temp1 * d2, k3 = p.temp1 * ca;
#pragma unroll
    for (int s = 0; s < NS; ++s) { opaque(lpv[s]); opaque(af[s]); }
    // dGm_b = sum over words of d2 a a^T: the row weight d2 on its own (p.d2: medmoe_gemm_tn_gram multiplies the A rows by it) and / or
    // the product d2 * A as a matrix (p.U: the two-operand form, medmoe_gemm_tn_cols)
    if (p.d2 && g == 0) p.d2[(long long)b * p.stat_rows + p.row_base + (long long)j * TP + t] = d2;

    if (p.U) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (32 * s + 31 < HW || 32 * s + 8 * g < pw) {
          float u[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) u[e] = a_of(s, e) * d2;
          *(uint4*)(p.U + off0 + 32 * s) = make_uint4(pack2bf(u[0], u[1]), pack2bf(u[2], u[3]), pack2bf(u[4], u[5]), pack2bf(u[6], u[7]));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- pass a: rd[hw] = sum over the caption's words of a1 da1 (word-softmax backward) ----
    // Row-sum mailboxes by epoch parity: the backward launch has no other exchange between two of these, so a wave can write epoch
    // k+1 while a partner still reads epoch k (never k+2: its own pass b of k+1 needs that partner's k+1 flag first)
    float* Rall = Rbase + (epoch & 1) * 16 * GR;
    float* Rw = Rall + wid * GR;
#pragma unroll
    for (int s = 0; s < NS; ++s) { opaque(lpv[s]); opaque(af[s]); }
#pragma unroll
    for (int rt = 0; rt < NRTA; ++rt) {
      const int sp = rt >> 1, h = rt & 1;
      const float4 L4 = *(const float4*)(Lw + 32 * sp + 8 * g + 4 * h);
      const f32x4_t y = y_tile_now(rt);
      const float Lr[4] = {L4.x, L4.y, L4.z, L4.w};
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float lp = lp_of(sp, 4 * h + r);
        const float a1 = __builtin_amdgcn_exp2f(lp);
        const float a = a_of(sp, 4 * h + r);
        pr[r] = a1 * (a * (k1 * fmaf(lp, LN2, Lr[r]) + (k2 * y[r] - k3)));
      }
      row16_sum4(pr[0], pr[1], pr[2], pr[3]);
      if (fr == 0) *(float4*)(Rw + 32 * sp + 8 * g + 4 * h) = make_float4(pr[0], pr[1], pr[2], pr[3]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NTT > 1) {
      flag_publish(flagR + wid * 4, epoch);
#pragma unroll
      for (int q = 0; q < NTT; ++q) flag_wait(flagR + (w0 + q) * 4, epoch);
    }
    // ---- pass b: dS = dnum A + a1 (da1 - rd), over the log-probabilities in place ----
#pragma unroll
    for (int s = 0; s < NS; ++s) { opaque(lpv[s]); opaque(af[s]); }
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      float ds[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rt = 2 * sp + h;
        if (rt < NRTA) {
          const float4 L4 = *(const float4*)(Lw + 32 * sp + 8 * g + 4 * h);
          float rd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int q = 0; q < NTT; ++q) {
            const float4 v = *(const float4*)(Rall + (w0 + q) * GR + 32 * sp + 8 * g + 4 * h);
            rd[0] += v.x; rd[1] += v.y; rd[2] += v.z; rd[3] += v.w;
          }
          const f32x4_t y = y_tile_now(rt);
          const float Lr[4] = {L4.x, L4.y, L4.z, L4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float lp = lp_of(sp, 4 * h + r);
            const float a1 = __builtin_amdgcn_exp2f(lp);
            const float a = a_of(sp, 4 * h + r);
            const float da1 = a * (k1 * fmaf(lp, LN2, Lr[r]) + (k2 * y[r] - k3));
            ds[4 * h + r] = dn * a + a1 * (da1 - rd[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[4 * h + r] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (32 * sp + 31 < HW || 32 * sp + 8 * g < pw)
        *(uint4*)(p.dS + off0 + 32 * sp) = make_uint4(pack2bf(ds[0], ds[1]), pack2bf(ds[2], ds[3]), pack2bf(ds[4], ds[5]), pack2bf(ds[6], ds[7]));
    }
  }
}

// Host entry.  lp / dS / A / U: [rows][ld] matrices (dS may be lp itself); gm: [B][GR][GR] bf16, GR = 32 ceil(HW / 32), zero outside
// [HW][HW]; lse: [B][Bc][HWP] fp32; stats: [B][stat_rows][2] fp32 (num, n2 of every (image, caption word)).
// dS == nullptr: the FORWARD launch - writes sim, A, stats (and att of the matching pairs).  Otherwise the BACKWARD launch - reads lp,
// A, stats, sim of a forward launch over the same class and writes dS, and U = d2 * A (if U) and / or the row weights d2 [B][stat_rows] (if d2).  Returns MM_ERR_SHAPE for geometries without an
// instantiation (medmoe_local_pair3_supported).
// Tests: force the number of caption chunks per image (0 = automatic: enough workgroups to fill the chip).  One chunk makes every
// workgroup walk the whole caption list, i.e. many loop iterations with mailbox exchanges, at test-sized batches too.
static int g_pair3_chunks = 0;
extern "C" int medmoe_local_pair3_chunks(int n) {
  if (n < 0) return MM_ERR_ARG;
  g_pair3_chunks = n;
  return MM_OK;
}

extern "C" int medmoe_local_pair3_supported(int HW, int T) {
  const int ntt = (T + 15) / 16;
  return ((HW == 64 && ntt == 1) || (HW == 196 && ntt >= 1 && ntt <= 5)) ? 1 : 0;
}

static int pair3_launch(const void* lp, void* dS, void* A, void* U, const float* lse, const void* gm, const float* wnorm,
                        const int* cap_lens, const float* gsim, float* sim, float* att, float* stats, long long stat_rows,
                        int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap,
                        int ntt, long long row_base, long long ld, long long bstride, int pw, float* d2, float* dwn, hipStream_t stream);

extern "C" int medmoe_local_pair3(const void* lp, void* dS, void* A, void* U, const float* lse, const void* gm, const float* wnorm,
                                  const int* cap_lens, const float* gsim, float* sim, float* att, float* stats, long long stat_rows,
                                  int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap,
                                  int ntt, long long row_base, long long ld, long long bstride, int pw, float* d2, hipStream_t stream) {
  return pair3_launch(lp, dS, A, U, lse, gm, wnorm, cap_lens, gsim, sim, att, stats, stat_rows, B, Bc, HW, T, temp1, temp2, eps, cap_list, n_cap, ntt,
                      row_base, ld, bstride, pw, d2, nullptr, stream);
}

// The backward launch that also writes dwn [B][stat_rows] fp32: the coefficient of w_t in d loss / d w_t that comes from the word's own norm
// in the cosine (-dL/dcos * cos / ||w_t||^2, per image; summed over the images by the caller).  The part through the scores is dS^T ctx.
extern "C" int medmoe_local_pair3_wgrad(const void* lp, void* dS, void* A, void* U, const float* lse, const void* gm, const float* wnorm,
                                        const int* cap_lens, const float* gsim, float* sim, float* att, float* stats, long long stat_rows,
                                        int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap,
                                        int ntt, long long row_base, long long ld, long long bstride, int pw, float* d2, float* dwn,
                                        hipStream_t stream) {
  if (!dS || !dwn) return MM_ERR_ARG;
  return pair3_launch(lp, dS, A, U, lse, gm, wnorm, cap_lens, gsim, sim, att, stats, stat_rows, B, Bc, HW, T, temp1, temp2, eps, cap_list, n_cap, ntt,
                      row_base, ld, bstride, pw, d2, dwn, stream);
}

static int pair3_launch(const void* lp, void* dS, void* A, void* U, const float* lse, const void* gm, const float* wnorm,
                        const int* cap_lens, const float* gsim, float* sim, float* att, float* stats, long long stat_rows,
                        int B, int Bc, int HW, int T, float temp1, float temp2, float eps, const int* cap_list, int n_cap,
                        int ntt, long long row_base, long long ld, long long bstride, int pw, float* d2, float* dwn, hipStream_t stream) {
  if (dS && !U && !d2) return MM_ERR_ARG;
  if (!lp || !lse || !gm || !wnorm || !cap_lens || !A || !sim || !stats) return MM_ERR_ARG;
  if (stat_rows < row_base + (long long)n_cap * ntt * 16) return MM_ERR_SHAPE;
  if (B <= 0 || Bc <= 0 || n_cap <= 0 || (ld % 8) || (bstride % 8) || (pw % 8) || pw < HW || pw > ((HW + 31) / 32) * 32 || ntt < 1 || ntt > 5 || ntt * 16 > ((T + 15) / 16) * 16) return MM_ERR_SHAPE;
  if (!medmoe_local_pair3_supported(HW, ntt * 16)) return MM_ERR_SHAPE;
  Pair3Args p;
  p.lp = (const uint16_t*)lp; p.dS = (bf16_t*)dS; p.A = (bf16_t*)A; p.U = (bf16_t*)U;
  p.lse = lse; p.gm = (const bf16_t*)gm; p.wnorm = wnorm; p.stats = stats; p.d2 = d2; p.dwn = dwn; p.stat_rows = stat_rows; p.cap_lens = cap_lens; p.gsim = gsim; p.sim = sim; p.att = att;
  p.cap_list = cap_list; p.row_base = row_base; p.ld = ld; p.bstride = bstride; p.n_cap = n_cap; p.B = B; p.Bc = Bc; p.HW = HW;
  p.HWP = ((HW + 15) / 16) * 16; p.pw = pw; p.T = T; p.temp1 = temp1; p.temp2 = temp2; p.eps = eps;
  // one workgroup per (image, caption chunk): >= ~4 workgroups per CU in total, chunks a multiple of the captions per iteration
  const int cpi = 16 / ntt;
  int n_chunk = g_pair3_chunks > 0 ? g_pair3_chunks : max(1, (1024 + B - 1) / B);
  int cpw = ((n_cap + n_chunk - 1) / n_chunk + cpi - 1) / cpi * cpi;
  n_chunk = (n_cap + cpw - 1) / cpw;
  p.caps_per_wg = cpw; p.n_chunk = n_chunk;
  const dim3 grid(B * n_chunk), block(1024);
#define P3(HW_, T_)                                                                                       \
  { if (dS && dwn) hipLaunchKernelGGL((local_pair3_kernel<HW_, T_, true, true>), grid, block, 0, stream, p);   \
    else if (dS) hipLaunchKernelGGL((local_pair3_kernel<HW_, T_, true>), grid, block, 0, stream, p);           \
    else hipLaunchKernelGGL((local_pair3_kernel<HW_, T_, false>), grid, block, 0, stream, p); }
  if (HW == 64) P3(64, 1)
  else switch (ntt) { case 1: P3(196, 1) break; case 2: P3(196, 2) break; case 3: P3(196, 3) break; case 4: P3(196, 4) break; default: P3(196, 5) break; }
#undef P3
  return mm_check_launch();
}
