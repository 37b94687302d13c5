// bf16 MFMA GEMMs for gfx950 (MI355X).
//
//   gemm_nt : C[M,N]  = epi(alpha * A[M,K] . B[N,K]^T)        (forward + dgrad; weights are
//             stored [out,in] = the reference's nn.Linear layout, multi_head_attention.py:35-36,
//             mlp.py:55-64, swin.py:18-30,88-92)
//   gemm_tn : dW[Nn,Kk] += G[M,Nn]^T . X[M,Kk]                 (wgrad, split over M, fp32 atomics)
//
// Both: 128x128 block tile, 4 waves (2x2, 64x64 each), K-step 64, operands staged HBM->LDS with
// global_load_lds_dwordx4 (16 B/lane, lane-linear LDS image, XOR swizzle applied on the SOURCE
// address and again on the LDS read), double-buffered, fp32 accumulation in MFMA.
// gemm_nt: v_mfma_f32_16x16x32_bf16 with the operands swapped so a lane owns 4 consecutive
// output columns (8-byte bf16 stores).  gemm_tn: v_mfma_f32_32x32x16_bf16 fed by
// ds_read_b64_tr_b16 transposed LDS reads (the contraction index is the ROW of both operands);
// a 32x32 accumulator register is two 128-B row segments = the full-rate float-atomic shape.
#include "common.h"
#include <utility>

#define BM 128
#define BN 128
#define BK 64

__device__ __attribute__((aligned(256))) bf16_t g_zero_page[2048];  // 4 KiB of zeros (OOB rows of gemm_tn)

struct GemmNTArgs {
  const bf16_t* A; const bf16_t* B; void* C;
  const float* bias; const bf16_t* residual; bf16_t* aux;
  const int* a_rowmap; const int* c_rowmap; const int4* tiles; const int* tile_count;
  long long strideB; long long strideBias;
  int M, N, K, lda, ldb, ldc, ldr, ldaux;
  int n_tiles_n, max_tiles_m;
  float alpha; int epi; int out_f32; int col_perm;
  const int* m_dev;      // plain (ungrouped) launches: if set, the row count M is read from the device (<= the host's M, which sizes the grid)
};

static int g_use_nt256 = 1, g_use_nt512 = 1, g_use_tn512 = 1, g_tn_rows = 1024, g_nt_max_grid = 256, g_use_scores512 = 1, g_use_nt4w = 1, g_use_tn4w = 1, g_tn_min_rows = 2048, g_tn_rows4w = 0, g_scores_skip_epi = 0;
// -DNT_TIMING (tools/nt_timing.hip): per-wave, per-phase shader-clock totals of gemm_nt256_kernel
#ifdef NT_EXPERIMENT
__device__ int g_nt_dbg_skip = 0;        // experiment (wrong results): bit 0 skip the LDS fragment reads, 1 the MFMAs, 2 the epilogue, 3 the DMA
#define NT_X(...) __VA_ARGS__
#else
#define NT_X(...)
#endif
#ifdef NT_TIMING
__device__ unsigned long long g_nt_timing[8 * 8];     // [wave][phase]
#define NT_T(...) __VA_ARGS__
__device__ __forceinline__ long long nt_clk() {      // a clock read the scheduler cannot move MFMAs across
  __builtin_amdgcn_sched_barrier(0); const long long c = clock64(); __builtin_amdgcn_sched_barrier(0); return c;
}
#else
#define NT_T(...)
#endif
extern int g_attn_resident;      // attention.hip
extern int g_sa_rows;            // moe.hip
extern int g_attn_rt13;          // attention.hip
// Which kernel the most recent medmoe_gemm_nt / _rows / _tiles256 call of this process launched: 0 gemm_nt_kernel (128x128), 1 gemm_nt256_kernel,
// 2 gemm_nt512_kernel, 3 gemm_nt512_kernel GROUPED, 4 gemm_nt4w_kernel, 5 gemm_nt4w_kernel GROUPED.  Measurement aid only (bench.py labels its
// per-launch HIP-event timings with it); nothing in the product path reads it.
static int g_last_nt_kernel = -1;
extern "C" int medmoe_last_gemm_nt_kernel() { return g_last_nt_kernel; }
extern int g_ln_one_row;
int g_use_nt_direct = 1;          // K % 64 == 32 products on gemm_nt_direct_kernel (below)
extern "C" int medmoe_set_option(int key, int value) {
  if (key == 16) { g_ln_one_row = value; return MM_OK; }
  if (key == 17) { g_use_nt_direct = value; return MM_OK; }                       // 0: K % 64 == 32 products on the 128x128 DMA kernel (A/B runs)                          // LayerNorm: 1 = one row per wave whatever the width
  if (key == 1) { g_use_nt256 = value; return MM_OK; }
  if (key == 2) { g_use_nt512 = value; return MM_OK; }
  if (key == 3) { g_use_tn512 = value; return MM_OK; }
  if (key == 4 && value >= 256) { g_tn_rows = value; return MM_OK; }
  if (key == 5 && value >= 1 && value <= 256) { g_nt_max_grid = value; return MM_OK; }      // experiments: fewer CUs
  if (key == 6) { g_use_scores512 = value; return MM_OK; }
  if (key == 7) { g_use_nt4w = value; return MM_OK; }
  if (key == 8) { g_use_tn4w = value; return MM_OK; }
  if (key == 9 && value >= 64) { g_tn_min_rows = value; return MM_OK; }
  if (key == 12) { g_scores_skip_epi = value; return MM_OK; }                      // MEASUREMENT ONLY: 1 = score GEMM without its epilogue, 2 = epilogue arithmetic without its stores
  if (key == 15 && (value == 0 || value == 448 || value == 512)) { g_attn_rt13 = value; return MM_OK; }   // attention, 13 key tiles: threads per workgroup
  if (key == 13 && value >= 0) { g_sa_rows = value; return MM_OK; }                   // scale_attn_bwd: rows per wave (0 = auto)
  if (key == 11) { g_attn_resident = value; return MM_OK; }                        // attention: 1 = resident kernels for N <= 272 (round-1 path)
  if (key == 10 && value >= 0) { g_tn_rows4w = value; return MM_OK; }               // grouped wgrad on gemm_tn4w: rows per range (0 = auto)             // plain wgrad: fewest rows per M range
  return MM_ERR_ARG;
}

// EPI_GELU stores the PRE-activation in aux and EPI_MUL_DGELU evaluates GELU' of it (two transcendentals per element in
// the backward epilogue: measured 21-31k clk per 256x256 tile against 26.5k for its MFMAs).  EPI_GELU_DAUX stores GELU'(z)
// instead and EPI_MUL_AUX multiplies by it: the engine's pair.
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_RELU = 2, EPI_MUL_DGELU = 3, EPI_MUL_DRELU = 4, EPI_GELU_DAUX = 5, EPI_MUL_AUX = 6 };

// Shared epilogue.  The B-matrix fragment rows are read through the permutation sigma(i) = 4*perm(i>>2) + (i&3),
// perm = {0,2,1,3} (see compute()), so lane (g = lane>>4) owns C[m][n..n+3] with n = 16*tn + 4*perm(g):
// lanes l and l+32 hold ADJACENT 4-column groups and one v_permlane32_swap per dword turns two 8-byte
// stores into one 16-byte store (cdna guide T21: the bf16 store tail is issue-bound per instruction).
// All operand loads (bias, residual, aux, row map) are issued up front with clamped addresses; only
// the stores are predicated.
// A finished bf16 C tile PARKED in registers as eight ready-to-store 16-byte pieces per lane: the 256x128
// kernel issues them one per LOAD segment of the next tile.  Stored at the seam, the 64 KB per CU leave as
// one burst from all 256 CUs at once (every tile takes the same time) and the chip waits for HBM to absorb
// 16 MB: measured 9200 clk per tile and wave in the epilogue against 13000 for the tile's MFMAs.
struct ParkedTile {
  uint4 c[8];
  int m_base, n_base, n_items, next;
};

// Returns the number of global STORE instructions this wave issued (wave-uniform): vmcnt counts stores
// too, so the counted waits of the DMA ring must leave exactly those youngest ops in flight.
// acc is the wave's whole accumulator array; the 64x64 block handled here starts at its 16-row tile TM0
// (no pointer into the array: it has to stay in registers through every inlined copy of this function).
// LDSX (gemm_nt4w, plain rows, bf16 output): the 16-byte pieces of a 32-row half of the block go through a wave-private 8-KB LDS
// region (C in its first 4 KB, the aux output in the second; 128-byte rows, 16-byte slots XOR-swizzled by (row >> 1) & 7: conflict-free
// both ways) and leave as FULL 128-byte lines, 8 rows per instruction.  The accumulator layout stores 16 rows x 64 bytes per instruction,
// and that shape is what a CU drains at 12.5 B/clk (tools/store_bw.hip: 30 GB/s per CU against 94 with whole lines) - it, not HBM, set
// the 9.2k clocks of a plain tile's epilogue.
typedef unsigned u32x4e_t __attribute__((ext_vector_type(4)));
template <bool PARK, int TM0 = 0, int TMS = 4, int TN0 = 0, int TNS = 4, bool LDSX = false>
__device__ __forceinline__ int nt_epilogue(const GemmNTArgs& p, f32x4_t (&acc)[TMS][TNS], int m_base, int m_end,
                                           int n_base, int group, int frag_row, int frag_q, ParkedTile* park, unsigned lds_x = 0u) {
  int n_stores = 0;
  const int pg = ((frag_q & 1) << 1) | (frag_q >> 1);
  long long mc[4];
  bool mok[4];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const int m = m_base + tm * 16 + frag_row;
    mok[tm] = m < m_end;
    const int mr = min(m, m_end - 1);
    mc[tm] = p.c_rowmap ? (long long)p.c_rowmap[mr] : (long long)mr;
  }
  int nn[4];
  bool nok[4];
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int n = n_base + tn * 16 + pg * 4;
    nok[tn] = n < p.N;
    nn[tn] = min(n, p.N - 4);
  }
  float4 b4[4];
  if (p.bias) {
    const float* bias = p.bias + (long long)group * p.strideBias;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) b4[tn] = *(const float4*)(bias + nn[tn]);
  }
  uint2 res[4][4], axv[4][4];
  if (p.residual) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) res[tm][tn] = *(const uint2*)(p.residual + mc[tm] * p.ldr + nn[tn]);
  }
  const bool mul_epi = p.epi == EPI_MUL_DGELU || p.epi == EPI_MUL_DRELU || p.epi == EPI_MUL_AUX;
  if (mul_epi) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) axv[tm][tn] = *(const uint2*)(p.aux + mc[tm] * p.ldaux + nn[tn]);
  }
  const bool wide = !p.out_f32 && !p.col_perm && (p.N & 7) == 0;
  const bool upper = frag_q >= 2;
  // 16-byte store of two adjacent 4-column groups after the half-wave exchange
  auto store_pair = [&](bf16_t* base, long long ld, int tm, int j, uint2 lo, uint2 hi, bool is_aux) {
    // lo = this lane's packed tile 2j, hi = tile 2j+1
    auto r0 = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
    if constexpr (PARK) {
      if (!is_aux) { park->c[tm * 2 + j] = make_uint4(r0[0], r1[0], r0[1], r1[1]); return; }
    }
    if constexpr (LDSX) {
      const int row_l = (tm & 1) * 16 + frag_row, slot = (2 * j + (upper ? 1 : 0)) * 2 + (frag_q & 1);
      const unsigned a = lds_x + (is_aux ? 4096u : 0u) + row_l * 128 + ((slot ^ ((row_l >> 1) & 7)) << 4);
      const u32x4e_t d = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
      asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(d) : "memory");
      return;
    }
    const int col = n_base + (2 * j + (upper ? 1 : 0)) * 16 + (frag_q & 1) * 8;
    const bool pred = mok[tm] && col < p.N;
    n_stores += (__ballot(pred) != 0ull) ? 1 : 0;     // an all-inactive store is branched around by the compiler
    if (pred) *(uint4*)(base + mc[tm] * ld + col) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
  };
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    uint2 o[4], zz[4];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[TM0 + tm][TN0 + tn][r] * p.alpha;
      if (p.bias) { v[0] += b4[tn].x; v[1] += b4[tn].y; v[2] += b4[tn].z; v[3] += b4[tn].w; }
      const bool ok = mok[tm] && nok[tn];
      zz[tn] = make_uint2(0u, 0u);
      if (p.epi == EPI_GELU_DAUX) {
        f32x2_t g0, g1, d0, d1;
        gelu_and_grad2((f32x2_t){v[0], v[1]}, g0, d0); gelu_and_grad2((f32x2_t){v[2], v[3]}, g1, d1);
        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
        if (p.aux) {
          zz[tn].x = pack2bf(d0[0], d0[1]); zz[tn].y = pack2bf(d1[0], d1[1]);
          if (!wide) { n_stores += (__ballot(ok) != 0ull) ? 1 : 0; if (ok) *(uint2*)(p.aux + mc[tm] * p.ldaux + nn[tn]) = zz[tn]; }
        }
      } else       if (p.epi == EPI_GELU) {
        if (p.aux) {
          zz[tn].x = pack2bf(v[0], v[1]); zz[tn].y = pack2bf(v[2], v[3]);
          if (!wide) { n_stores += (__ballot(ok) != 0ull) ? 1 : 0; if (ok) *(uint2*)(p.aux + mc[tm] * p.ldaux + nn[tn]) = zz[tn]; }
        }
        {
          const f32x2_t g0 = gelu2((f32x2_t){v[0], v[1]}), g1 = gelu2((f32x2_t){v[2], v[3]});
          v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
        }
      } else if (p.epi == EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      }
      if (p.residual) {
        const uint2 z = res[tm][tn];
        v[0] += bf2f((bf16_t)(z.x & 0xffff)); v[1] += bf2f((bf16_t)(z.x >> 16));
        v[2] += bf2f((bf16_t)(z.y & 0xffff)); v[3] += bf2f((bf16_t)(z.y >> 16));
      }
      if (mul_epi) {   // (acc + residual) * act'(aux)
        const uint2 z = axv[tm][tn];
        const float zf[4] = {bf2f((bf16_t)(z.x & 0xffff)), bf2f((bf16_t)(z.x >> 16)),
                             bf2f((bf16_t)(z.y & 0xffff)), bf2f((bf16_t)(z.y >> 16))};
        if (p.epi == EPI_MUL_DGELU) {
          const f32x2_t d0 = dgelu2((f32x2_t){zf[0], zf[1]}), d1 = dgelu2((f32x2_t){zf[2], zf[3]});
          v[0] *= d0[0]; v[1] *= d0[1]; v[2] *= d1[0]; v[3] *= d1[1];
        } else if (p.epi == EPI_MUL_AUX) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= zf[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= zf[r] > 0.f ? 1.f : 0.f;
        }
      }
      o[tn].x = pack2bf(v[0], v[1]); o[tn].y = pack2bf(v[2], v[3]);
      if (!wide) n_stores += (__ballot(ok) != 0ull) ? 1 : 0;
      if (!wide && ok) {
        // col_perm: store column n at position lpos(n) (the k-order the local-loss Gm.A product reads)
        const int n = nn[tn];
        const int ns = p.col_perm ? ((n & ~31) + ((((n & 31) & 15) >> 2) << 3) + (((n & 31) >> 4) << 2)) : n;
        if (p.out_f32) *(float4*)((float*)p.C + mc[tm] * p.ldc + ns) = make_float4(v[0], v[1], v[2], v[3]);
        else *(uint2*)((bf16_t*)p.C + mc[tm] * p.ldc + ns) = o[tn];
      }
    }
    if (wide) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        store_pair((bf16_t*)p.C, p.ldc, tm, j, o[2 * j], o[2 * j + 1], false);
        if ((p.epi == EPI_GELU || p.epi == EPI_GELU_DAUX) && p.aux) store_pair(p.aux, p.ldaux, tm, j, zz[2 * j], zz[2 * j + 1], true);
      }
      if constexpr (LDSX) {
        if (tm & 1) {                                   // rows (tm - 1) * 16 .. tm * 16 + 15 of the block are in the LDS: out as whole lines
          const bool has_aux = (p.epi == EPI_GELU || p.epi == EPI_GELU_DAUX) && p.aux;
          const int lane = frag_q * 16 + frag_row, c8 = lane & 7;
          const int col = n_base + c8 * 8;
          u32x4e_t vc[4], va[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row_l = i * 8 + (lane >> 3);
            const unsigned a = lds_x + row_l * 128 + ((c8 ^ ((row_l >> 1) & 7)) << 4);
            asm volatile("ds_read_b128 %0, %1" : "=v"(vc[i]) : "v"(a));
            if (has_aux) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(va[i]) : "v"(a));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = m_base + (tm - 1) * 16 + i * 8 + (lane >> 3);
            const bool pred = m < m_end && col < p.N;
            const int ns = (__ballot(pred) != 0ull) ? 1 : 0;
            n_stores += has_aux ? 2 * ns : ns;
            if (pred) {
              // nontemporal: C is written once and read by a later kernel - streaming it past the L2 keeps the A / B panels of the tiles
              // in flight resident (QKV forward 1040 -> 1163, plain 3072 x 768 1092 -> 1182 TFLOP/s; no change where N = 768)
              __builtin_nontemporal_store(vc[i], (u32x4e_t*)((bf16_t*)p.C + (long long)m * p.ldc + col));
              if (has_aux) __builtin_nontemporal_store(va[i], (u32x4e_t*)(p.aux + (long long)m * p.ldaux + col));
            }
          }
        }
      }
    }
  }
  if constexpr (PARK) { park->m_base = m_base; park->n_base = n_base; park->next = 0; park->n_items = 8; }
  return n_stores;
}

// store the next parked piece (number `item`, 0..7) of the C tile and rotate the queue - the pieces live in
// registers, which cannot be indexed at run time, and an 8-way switch costs more scalar branching per k-step
// than 28 v_mov; returns 1 if a store instruction issued (wave-uniform)
__device__ __forceinline__ int park_store_next(const GemmNTArgs& p, ParkedTile& pk, int m_end, int frag_row, int frag_q) {
  const int item = pk.next++;
  const uint4 v = pk.c[0];
#pragma unroll
  for (int i = 0; i < 7; ++i) pk.c[i] = pk.c[i + 1];
  const int tm = item >> 1, j = item & 1;
  const int row = pk.m_base + tm * 16 + frag_row;
  const int col = pk.n_base + (2 * j + (frag_q >= 2 ? 1 : 0)) * 16 + (frag_q & 1) * 8;
  const bool pred = row < m_end && col < p.N;
  if (__builtin_amdgcn_readfirstlane(__ballot(pred) != 0ull ? 1 : 0) == 0) return 0;
  if (pred) *(uint4*)((bf16_t*)p.C + (long long)row * p.ldc + col) = v;
  return 1;
}

// s_waitcnt vmcnt(<= n): the immediate must be a literal, and a 64-way switch on it costs hundreds of cycles
// of scalar branching per k-step.  Waiting for a smaller count than allowed is always correct (it only waits
// for more), so only the counts the DMA rings actually produce get their own immediate: 6 = one stage in
// flight, +1/+2 parked stores, +8 pre-activation stores of a GELU epilogue; anything else rounds down.
__device__ __forceinline__ void wait_vmcnt(int n) {
  if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// s_waitcnt vmcnt(6 + n): the immediate must be a literal
__device__ __forceinline__ void wait_vmcnt_plus6(int n) {
  switch (n) {
#define WV(k, k6) case k: asm volatile("s_waitcnt vmcnt(" #k6 ")" ::: "memory"); break;
    WV(1, 7) WV(2, 8) WV(3, 9) WV(4, 10) WV(5, 11) WV(6, 12) WV(7, 13) WV(8, 14) WV(9, 15) WV(10, 16) WV(11, 17) WV(12, 18)
    WV(13, 19) WV(14, 20) WV(15, 21) WV(16, 22)
#undef WV
    default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
  }
}

// Persistent over output tiles: a block walks tiles  t = round*grid + xcd_remap(block)  and runs ONE
// continuous double-buffered pipeline over (tile, k-step); the first k-tile of the next output tile is
// already in flight while the current tile's epilogue stores run, so the per-tile prologue/epilogue
// (~1/3 of a K=768 tile's lifetime) hides behind LDS-DMA traffic.
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNTArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 32768];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int frag_row = lane & 15, frag_q = lane >> 4, swz = lane & 7;
  const int G = gridDim.x;
  const int my = xcd_remap(blockIdx.x, G);
  if (!p.tiles && p.m_dev) { p.M = min(*p.m_dev, p.M); p.max_tiles_m = (p.M + BM - 1) / BM; }
  const int n_tiles_m = p.tiles ? min(*p.tile_count, p.max_tiles_m) : p.max_tiles_m;
  const int total = n_tiles_m * p.n_tiles_n;
  // K % 64 == 32 (Swin-T stage 1: 96 and 288 channels): the last k-step is half a stage - lanes that would fetch its upper four
  // 16-byte chunks re-fetch the lower four (valid memory, never multiplied) and the step runs one MFMA k-half instead of two
  const int nt = (p.K + BK - 1) / BK;
  const bool khalf = (p.K & 32) != 0;
  const int tail_adj = (khalf && (((lane & 7) ^ ((lane >> 3) & 7)) >= 4)) ? -32 : 0;

  struct Tile { int group, m0, m_end, n0; };
  auto decode = [&](int id) -> Tile {
    Tile t;
    const int tile_m = id / p.n_tiles_n, tile_n = id - tile_m * p.n_tiles_n;
    if (p.tiles) { const int4 q = p.tiles[tile_m]; t.group = q.x; t.m0 = q.y; t.m_end = q.z; }
    else { t.group = 0; t.m0 = tile_m * BM; t.m_end = p.M; }
    t.n0 = tile_n * BN;
    return t;
  };
  const bf16_t* asrc[4];
  const bf16_t* bsrc[4];
  auto setup = [&](const Tile& t) {
    const bf16_t* Bg = p.B + (long long)t.group * p.strideB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 32 + wid * 8 + (lane >> 3);
      const int c = (lane & 7) ^ (row & 7);
      int ra = min(t.m0 + row, t.m_end - 1);
      if (p.a_rowmap) ra = p.a_rowmap[ra];
      asrc[i] = p.A + (long long)ra * p.lda + c * 8;
      const int rb = min(t.n0 + row, p.N - 1);
      bsrc[i] = Bg + (long long)rb * p.ldb + c * 8;
    }
  };
  auto stage = [&](int buf, int ks) {
    char* sA = smem + buf * 32768 + wid * 1024;
    char* sB = sA + 16384;
    const int k0 = ks * BK + ((ks == nt - 1) ? tail_adj : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i] + k0), LDS_PTR(sA + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(bsrc[i] + k0), LDS_PTR(sB + i * 4096), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  // B-matrix fragment rows go through sigma(i) = 4*perm(i>>2) + (i&3), perm = {0,2,1,3}: see nt_epilogue
  const int sig = ((((frag_row >> 2) & 1) << 1 | (frag_row >> 3)) << 2) | (frag_row & 3);
  auto compute = [&](int buf, int nks) {
    const char* sA = smem + buf * 32768 + (wm * 64 + frag_row) * 128;
    const char* sB = smem + buf * 32768 + 16384 + (wn * 64 + sig) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks >= nks) break;
      const int coff = ((ks * 4 + frag_q) ^ swz) * 16;
      const int coffb = ((ks * 4 + frag_q) ^ (sig & 7)) * 16;
      bf16x8_t af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *(const bf16x8_t*)(sA + t * 2048 + coff);
        bf[t] = *(const bf16x8_t*)(sB + t * 2048 + coffb);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tm][tn], 0, 0, 0);
    }
  };
  auto epilogue = [&](const Tile& t) -> int { return nt_epilogue<false>(p, acc, t.m0 + wm * 64, t.m_end, t.n0 + wn * 64, t.group, frag_row, frag_q, nullptr); };

  int id = my;
  if (id >= total) return;
  Tile cur_t = decode(id);
  setup(cur_t);
  zero_acc();
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
  while (true) {
    const int nid = id + G;
    const bool has_next = nid < total;
    Tile next_t = cur_t;
    for (int ks = 0; ks < nt; ++ks) {
      if (ks + 1 < nt) {
        stage(cur ^ 1, ks + 1);
      } else if (has_next) {            // nothing is in flight here: row-map loads cost no DMA drain
        next_t = decode(nid);
        setup(next_t);
        stage(cur ^ 1, 0);
      }
      compute(cur, (khalf && ks == nt - 1) ? 1 : 2);
      if (ks == nt - 1) { epilogue(cur_t); zero_acc(); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
    if (!has_next) break;
    id = nid; cur_t = next_t;
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_nt_direct: short contractions that are not a multiple of the 64-wide k-step of the DMA kernels (K % 64 == 32, K <= 288: the 96 and
// 288 channels of Swin-T's first stage over 100 k token rows).  Such a product is a stream of A and C rows around a weight matrix of a few
// tens of KB: nothing to stage.  A wave owns a 64 x 64 block of C and takes BOTH operands' fragments straight from global memory (the
// weights stay in L2 / the vector L1; the six column blocks of one 64-row A panel are consecutive waves), runs K / 32 steps of 16 MFMAs with
// the next step's fragments in flight, and leaves through nt_epilogue - no LDS, no barrier, no tile pipeline to fill and drain every two
// k-steps (the 128 x 128 DMA kernel spent its time there: 1.75 TB/s at [100352, 384] x K = 96).
// ---------------------------------------------------------------------------------------------
template <int SPEC>
__global__ __launch_bounds__(256) void gemm_nt_direct_kernel(GemmNTArgs p_in) {
  GemmNTArgs p = p_in;
  if constexpr (SPEC >= 0) {          // as gemm_nt256_kernel: the epilogue flags are compile-time constants, only that path is emitted
    p.epi = SPEC & 7; p.out_f32 = 0; p.col_perm = 0; p.c_rowmap = nullptr; p.a_rowmap = nullptr; p.alpha = 1.f;
    if (!(SPEC & 8)) p.bias = nullptr;
    if (!(SPEC & 16)) p.residual = nullptr;
    if (!(SPEC & 32)) p.aux = nullptr;
    __builtin_assume((p.N & 7) == 0);
    if (SPEC & 8) __builtin_assume(p.bias != nullptr);
    if (SPEC & 16) __builtin_assume(p.residual != nullptr);
    if (SPEC & 32) __builtin_assume(p.aux != nullptr);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int frag_row = lane & 15, frag_q = lane >> 4;
  const int nblk_n = (p.N + 63) >> 6, nblk_m = (p.M + 63) >> 6;
  const long long w = (long long)blockIdx.x * 4 + wid;
  if (w >= (long long)nblk_m * nblk_n) return;                      // whole waves leave: no barrier anywhere in this kernel
  const int mb = (int)(w / nblk_n), nb = (int)(w - (long long)mb * nblk_n);
  const int m0 = mb * 64, n0 = nb * 64;
  const int sig = ((((frag_row >> 2) & 1) << 1 | (frag_row >> 3)) << 2) | (frag_row & 3);   // sigma(frag_row), see nt_epilogue
  const bf16_t* ar[4];
  const bf16_t* br[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    ar[t] = p.A + (long long)min(m0 + t * 16 + frag_row, p.M - 1) * p.lda + frag_q * 8;
    br[t] = p.B + (long long)min(n0 + t * 16 + sig, p.N - 1) * p.ldb + frag_q * 8;
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int nks = p.K >> 5;
  bf16x8_t af[4], bf[4], an[4], bn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { af[t] = *(const bf16x8_t*)ar[t]; bf[t] = *(const bf16x8_t*)br[t]; }
  for (int ks = 0; ks < nks; ++ks) {
    if (ks + 1 < nks) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { an[t] = *(const bf16x8_t*)(ar[t] + (ks + 1) * 32); bn[t] = *(const bf16x8_t*)(br[t] + (ks + 1) * 32); }
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) { af[t] = an[t]; bf[t] = bn[t]; }
  }
  nt_epilogue<false>(p, acc, m0, p.M, n0, 0, frag_row, frag_q, nullptr);
}

// ---------------------------------------------------------------------------------------------
// gemm_nt256: 256x128 block tile, 8 waves (4x2 of 64x64), K-step 64, THREE-stage LDS ring filled by
// LDS-DMA with a COUNTED s_waitcnt vmcnt(6) + raw s_barrier per k-step (one stage = 6 DMA ops per
// thread stays in flight across the barrier; cdna guide "Pipelining across barriers"), persistent
// over tiles with separate load / compute cursors so the ring never drains at tile seams.
// 85 flop per LDS-filled byte (vs 64 for the 128x128 kernel): the 128^2 kernel sits on the
// L2->LDS fill rate at K=768.  Used for the plain (ungrouped, unmapped-A) Linear GEMMs.
// ---------------------------------------------------------------------------------------------
#define BM2 256
#define STAGE2 (384 * 128)
// SPEC < 0: every epilogue option decided at run time (70 KB of code - more than the instruction cache, and
// the tile seam then runs at instruction-fetch speed).  SPEC >= 0 = epi | bias << 3 | residual << 4 | aux << 5
// for a bf16 C with N % 8 == 0 and alpha == 1: the flags are compile-time constants and only that path is emitted.
#define NT_SPEC(epi, bias, res, aux) ((epi) | ((bias) << 3) | ((res) << 4) | ((aux) << 5))
template <int SPEC>
__global__ __launch_bounds__(512, 2) void gemm_nt256_kernel(GemmNTArgs p_in) {
  GemmNTArgs p = p_in;
  if (p.m_dev) { p.M = min(*p.m_dev, p.M); p.max_tiles_m = (p.M + BM2 - 1) / BM2; }
  if constexpr (SPEC >= 0) {
    p.epi = SPEC & 7; p.out_f32 = (SPEC >> 6) & 1; p.col_perm = 0; p.c_rowmap = nullptr; p.a_rowmap = nullptr; p.alpha = 1.f;
    if (!(SPEC & 8)) p.bias = nullptr;
    if (!(SPEC & 16)) p.residual = nullptr;
    if (!(SPEC & 32)) p.aux = nullptr;
    __builtin_assume((p.N & 7) == 0);
    if (SPEC & 8) __builtin_assume(p.bias != nullptr);
    if (SPEC & 16) __builtin_assume(p.residual != nullptr);
    if (SPEC & 32) __builtin_assume(p.aux != nullptr);
  }
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int frag_row = lane & 15, frag_q = lane >> 4, swz = lane & 7;
  const int G = gridDim.x;
  const int my = xcd_remap(blockIdx.x, G);
  const int total = p.max_tiles_m * p.n_tiles_n;
  const int nt = p.K / BK;

  // Tile order: ids walk "super rows" of SM m-tiles; inside one, blocks of (SM m x SN n) tiles are
  // consecutive, so the 32 tiles an XCD holds at a time share SM A panels and SN B panels (each A chunk is
  // fetched once per SN tiles, each B chunk once per SM tiles) instead of one A panel and every B panel.
  constexpr int SM = 4, SN = 8;   // 2x16 / 8x4 / non-temporal A loads measured within noise of this (profiles/r01_notes.md)
  struct Tile { int m0, n0; };
  auto decode = [&](int id) -> Tile {
    Tile t;
    const int per_super = SM * p.n_tiles_n;
    const int sg = id / per_super, r = id - sg * per_super;
    const int rows = min(SM, p.max_tiles_m - sg * SM);          // m-tiles in this (possibly last) super row
    const int blk = SN * rows;                                  // tiles per (rows x SN) block
    const int nfull = p.n_tiles_n / SN;
    int tile_m, tile_n;
    if (r < nfull * blk) { const int nb = r / blk, w = r - nb * blk; tile_n = nb * SN + (w % SN); tile_m = sg * SM + (w / SN); }
    else {                                                      // ragged tail block: n_tiles_n % SN columns
      const int r2 = r - nfull * blk, nc = p.n_tiles_n - nfull * SN;
      tile_n = nfull * SN + r2 % nc; tile_m = sg * SM + r2 / nc;
    }
    t.m0 = tile_m * BM2; t.n0 = tile_n * BN;
    return t;
  };
  // per-lane sources as 32-bit BYTE offsets from the (uniform) A / B base: the DMA then uses the SGPR-base +
  // VGPR-offset address form and the six cursors cost 6 VGPRs instead of 12 (the host routes operands
  // of 4 GB or more to the 128x128 kernel)
  NT_X(const int dbg_skip = g_nt_dbg_skip;)
  unsigned src[6];
  auto setup = [&](const Tile& t) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int q = i * 512 + tid;
      const int row = q >> 3;
      const int c = (q & 7) ^ (row & 7);
      if (i < 4) src[i] = (unsigned)min(t.m0 + row, p.M - 1) * (unsigned)(p.lda * 2) + c * 16;
      else src[i] = (unsigned)min(t.n0 + row - BM2, p.N - 1) * (unsigned)(p.ldb * 2) + c * 16;
    }
  };
  auto stage = [&](int buf, int k0) {
    char* sb = smem + buf * STAGE2 + wid * 1024;
#pragma unroll
    for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)(i < 4 ? p.A : p.B) + k0 * 2 + src[i]), LDS_PTR(sb + i * 8192), 16, 0, 0);
  };
  f32x4_t acc[4][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  const int sig = ((((frag_row >> 2) & 1) << 1 | (frag_row >> 3)) << 2) | (frag_row & 3);   // sigma(frag_row), see nt_epilogue
  bf16x8_t af[2][4], bf[2][4];          // one k-step of fragments: [k sub-step][16-row tile]
  auto read_frags = [&](int buf) {
    const char* sA = smem + buf * STAGE2 + (wm * 64 + frag_row) * 128;
    const char* sB = smem + buf * STAGE2 + BM2 * 128 + (wn * 64 + sig) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + frag_q) ^ swz) * 16;
      const int coffb = ((ks * 4 + frag_q) ^ (sig & 7)) * 16;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[ks][t] = *(const bf16x8_t*)(sA + t * 2048 + coff);
        bf[ks][t] = *(const bf16x8_t*)(sB + t * 2048 + coffb);
      }
    }
  };
  // the 32 MFMAs of one k-step with the wave's six DMA pieces of a later stage issued IN BETWEEN (after every
  // fifth MFMA): a DMA issue stalls the in-order wave for tens of cycles while the address unit is busy, and
  // here that stall only delays MFMA issue behind MFMAs still executing; in the LOAD segment it was the
  // critical path (measured 1100 clk per LOAD segment against 550 for the partner's MFMAs)
  auto compute = [&](bool dma, int wbuf, int k0) __attribute__((always_inline)) {
    char* sw = smem + wbuf * STAGE2 + wid * 1024;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][tn], af[ks][tm], acc[tm][tn], 0, 0, 0);
        if (tm < 3) {
          const int i = ks * 3 + tm;
          if (dma) __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)(i < 4 ? p.A : p.B) + k0 * 2 + src[i]), LDS_PTR(sw + i * 8192), 16, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
  };
  const bool can_park = !p.out_f32 && !p.col_perm && (p.N & 7) == 0;
  ParkedTile pk;
  pk.n_items = 0; pk.next = 0; pk.m_base = 0; pk.n_base = 0;
  auto epilogue = [&](const Tile& t) -> int {
    if (can_park) return nt_epilogue<true>(p, acc, t.m0 + wm * 64, p.M, t.n0 + wn * 64, 0, frag_row, frag_q, &pk);
    return nt_epilogue<false>(p, acc, t.m0 + wm * 64, p.M, t.n0 + wn * 64, 0, frag_row, frag_q, nullptr);
  };
  auto flush_parked = [&]() -> int {
    int n = 0;
    while (pk.next < pk.n_items) n += park_store_next(p, pk, p.M, frag_row, frag_q);
    return n;
  };

  if (my >= total) return;
  const int my_tiles = (total - my + G - 1) / G;
  Tile ct = decode(my);
  setup(ct);
  // vmcnt bookkeeping: the wait is always for the wave's SECOND-newest stage; younger than it are the stores
  // issued between the two stages (st_old), the newest stage's pieces (dn) and the stores since (st_new)
  int st_old = 0, st_new = 0, dn = 0, wb = 0, rb = 0;
  // DMA cursor: tile lid, k-step lk.  Every wave issues its own six pieces of every stage, in stage order.
  int lid = my, lk = 0;
  bool lmore = true;
  auto advance_load = [&]() {
    wb = (wb == 2) ? 0 : wb + 1;
    if (++lk == nt) {
      lk = 0; lid += G;
      if (lid < total) { const Tile lt = decode(lid); setup(lt); } else lmore = false;
    }
  };
  auto wait_second_newest = [&]() { wait_vmcnt(__builtin_amdgcn_readfirstlane(st_old + dn + st_new)); };
  const int grp = __builtin_amdgcn_readfirstlane(wid >> 2);
  zero_acc();
  // prologue: group 0 starts two stages ahead of its compute, group 1 three (see below)
  for (int i = 0; i < 2 + grp; ++i) { st_old = st_new; st_new = 0; stage(wb, lk * BK); dn = 6; advance_load(); }
  NT_T(long long t_wait = 0, t_bar = 0, t_load = 0, t_comp = 0, t_epi = 0; const long long t_begin = nt_clk();)
  if (grp == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // stage 0 landed for every wave
  asm volatile("" ::: "memory");
  // PING-PONG: the two waves of a SIMD (wave w and w+4) alternate roles every segment, separated by one
  // s_barrier: while group 0 runs the 32 MFMAs of its k-step (the matrix pipe is per SIMD and fully paced),
  // group 1 reads its next fragments from LDS, then they swap.  In lock-step (both waves reading, then both
  // computing) the matrix pipe idled through every LDS-read latency: measured 1800 clk per k-step against
  // 1024 of MFMA issue (tools/nt_timing.hip, profiles/r01_notes.md).
  // Every wave runs the same program  { LOAD(t); barrier; COMPUTE(t) [+ epilogue]; barrier }  and group 1 is
  // shifted by one barrier, so its LOAD coincides with group 0's COMPUTE (absolute segment 2t+g for LOAD(t),
  // 2t+1+g for COMPUTE(t)).  Stage s lives in buffer s % 3 and is read in segments 2s (group 0) and 2s+1
  // (group 1).  COMPUTE(t) issues the wave's pieces of stage t+2 (group 0, segment 2t+1) or t+3 (group 1,
  // segment 2t+2) into the buffer last read in segment 2t-1 resp. 2t+1.  A stage must be complete before
  // segment 2s: group 0 waits for its SECOND-newest stage at the end of COMPUTE (segment 2s-1), group 1 at the
  // end of LOAD (segment 2s-1) - three and four segments after issue.
  auto seg_barrier = [&]() {
    NT_T(const long long b0 = nt_clk();)
    __builtin_amdgcn_sched_barrier(0);
    if (true NT_X(&& !(dbg_skip & 16))) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    NT_T(t_bar += nt_clk() - b0;)
  };
  auto step = [&](bool last) __attribute__((always_inline)) {
    NT_T(const long long c0 = nt_clk();)
    if (true NT_X(&& !(dbg_skip & 1))) read_frags(rb);
    rb = (rb == 2) ? 0 : rb + 1;
    if (pk.next < pk.n_items) st_new += park_store_next(p, pk, p.M, frag_row, frag_q);   // one parked piece of the previous tile
    NT_T(const long long c1 = nt_clk(); t_load += c1 - c0;)
    if (grp == 1 NT_X(&& !(dbg_skip & 32))) { wait_second_newest(); NT_T(t_wait += nt_clk() - c1;) }
    seg_barrier();
    NT_T(const long long c2 = nt_clk();)
    const bool dma = lmore NT_X(&& !(dbg_skip & 8));
    st_old = st_new; st_new = 0; dn = dma ? 6 : 0;
    if (true NT_X(&& !(dbg_skip & 2))) compute(dma, wb, lk * BK);
    if (lmore) advance_load();
    NT_T(const long long c3 = nt_clk(); t_comp += c3 - c2;)
    if (last) {
      st_new += flush_parked();                   // only when K is shorter than the piece count
      if (true NT_X(&& !(dbg_skip & 4))) st_new += __builtin_amdgcn_readfirstlane(epilogue(ct));
      zero_acc();
      NT_T(t_epi += nt_clk() - c3;)
    }
    NT_T(const long long c4 = nt_clk();)
    if (grp == 0 NT_X(&& !(dbg_skip & 32))) { wait_second_newest(); NT_T(t_wait += nt_clk() - c4;) }
    seg_barrier();
  };
  if (grp == 1) seg_barrier();
  for (int ti = 0; ti < my_tiles; ++ti) {
    for (int k = 1; k < nt; ++k) step(false);
    step(true);
    if (ti + 1 < my_tiles) ct = decode(my + (ti + 1) * G);
  }
  if (grp == 0) seg_barrier();
  flush_parked();                                 // the last tile
  NT_T(if ((tid & 63) == 0) { unsigned long long* g = g_nt_timing + wid * 8; atomicAdd(&g[0], (unsigned long long)(nt_clk() - t_begin));
         atomicAdd(&g[1], (unsigned long long)t_wait); atomicAdd(&g[2], (unsigned long long)t_bar); atomicAdd(&g[3], (unsigned long long)t_load);
         atomicAdd(&g[4], (unsigned long long)t_comp); atomicAdd(&g[5], (unsigned long long)t_epi); atomicAdd(&g[6], 1ull); })
}

// ---------------------------------------------------------------------------------------------
// gemm_nt512: 256x256 block tile, 8 waves (2x4 of 128x64), for wide-N Linear GEMMs.
// The 256x128 kernel moves 48 KB through the CU's address/L1 path per 1100 clk of MFMA work (43 B/clk of
// the 64 B/clk the path can carry, measured ~46 B/clk with four waves issuing): every DMA issue stalls its
// wave ~90 clk, and that stall, not the matrix pipe, sets the k-step.  256x256 needs 32 KB per 1100 clk.
//   * k sub-step 32 (ONE 16x16x32 MFMA deep): 32 MFMAs per wave and sub-step, 12 ds_read_b128, 4 DMA pieces.
//   * LDS: ring of FOUR 32 KB sub-stages (512 rows x 64 B).  Two 64-B rows share one 128-B LDS line:
//     row r, 16-B k-chunk c lives in line r >> 1 at chunk (((r & 1) << 2 | c) ^ ((r >> 1) & 7)) - conflict
//     free for the b128 fragment reads (same lane-group argument as the 128-B-row image).
//   * ping-pong as in gemm_nt256: { LOAD(u); barrier; COMPUTE(u); barrier }, group 1 shifted by one barrier.
//     LOAD(u) issues sub-stage u+3 into the buffer of u-1 (read one barrier earlier at the latest) and the
//     wave waits for its THIRD-newest sub-stage before the barrier that precedes the first read of it:
//     four to five segments after issue.
// ---------------------------------------------------------------------------------------------
#define SUB3 (512 * 64)
#define NRING 4        // sub-stages in the LDS ring (5 x 32 KB = the whole 160 KB measured no faster, 3 measured 5 % slower)
__device__ __forceinline__ void wait_vmcnt_ring4(int n) {
  if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// GROUPED: tiles come from the device-side table p.tiles (group, m0, m_end, -) of 256-ROW tiles, B / bias are
// per group, A rows are gathered through p.a_rowmap and C rows scattered through p.c_rowmap (expert GEMMs, dGm).
template <int SPEC, bool GROUPED = false>
__global__ __launch_bounds__(512, 2) void gemm_nt512_kernel(GemmNTArgs p_in) {
  GemmNTArgs p = p_in;
  if (!GROUPED && p.m_dev) { p.M = min(*p.m_dev, p.M); p.max_tiles_m = (p.M + 255) / 256; }
  if constexpr (SPEC >= 0) {
    p.epi = SPEC & 7; p.out_f32 = (SPEC >> 6) & 1; p.col_perm = 0; p.alpha = 1.f;
    if (!GROUPED) { p.c_rowmap = nullptr; p.a_rowmap = nullptr; }
    if (!(SPEC & 8)) p.bias = nullptr;
    if (!(SPEC & 16)) p.residual = nullptr;
    if (!(SPEC & 32)) p.aux = nullptr;
    __builtin_assume((p.N & 7) == 0);
    if (SPEC & 8) __builtin_assume(p.bias != nullptr);
    if (SPEC & 16) __builtin_assume(p.residual != nullptr);
    if (SPEC & 32) __builtin_assume(p.aux != nullptr);
  }
  __shared__ __attribute__((aligned(16))) char smem[NRING * SUB3];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;          // the two waves of a SIMD (w, w+4) take different column strips
  const int frag_row = lane & 15, frag_q = lane >> 4;
  const int G = gridDim.x;
  const int my = xcd_remap(blockIdx.x, G);
  const int total = GROUPED ? *p.tile_count * p.n_tiles_n : p.max_tiles_m * p.n_tiles_n;
  const int nu = p.K / 32;                        // sub-steps per tile

  constexpr int SM = 4, SN = 8;                   // tile order as in gemm_nt256
  struct Tile { int m0, n0, m_end, group; };
  auto decode = [&](int id) -> Tile {
    Tile t;
    if constexpr (GROUPED) {                      // n-tiles of one m-tile are consecutive (they share the gathered A rows)
      const int tm = id / p.n_tiles_n;
      const int4 e = p.tiles[tm];
      t.group = e.x; t.m0 = e.y; t.m_end = e.z; t.n0 = (id - tm * p.n_tiles_n) * 256;
      return t;
    }
    t.group = 0; t.m_end = p.M;
    const int per_super = SM * p.n_tiles_n;
    const int sg = id / per_super, r = id - sg * per_super;
    const int rows = min(SM, p.max_tiles_m - sg * SM);
    const int blk = SN * rows;
    const int nfull = p.n_tiles_n / SN;
    int tile_m, tile_n;
    if (r < nfull * blk) { const int nb = r / blk, w = r - nb * blk; tile_n = nb * SN + (w % SN); tile_m = sg * SM + (w / SN); }
    else {
      const int r2 = r - nfull * blk, nc = p.n_tiles_n - nfull * SN;
      tile_n = nfull * SN + r2 % nc; tile_m = sg * SM + r2 / nc;
    }
    // row tiles are walked from the LAST to the first: the kernel that produced A wrote it front to back, so its tail is
    // what still sits in the Infinity Cache / L2 when this kernel starts
    t.m0 = (p.max_tiles_m - 1 - tile_m) * 256; t.n0 = tile_n * 256;
    return t;
  };
  // DMA piece i of wave w fills LDS bytes [(i * 8 + w) * 1024, +1024) of the sub-stage: 8 lines = 16 rows.
  // Lane l writes chunk l & 7 of line (i * 8 + w) * 8 + (l >> 3): it must FETCH what that slot holds.
  // Per-lane sources are 32-bit byte offsets from a per-TILE base (uniform, 64-bit): a tile's rows span at most
  // 256 * ld elements, so operands beyond 4 GB (the 22 GB pair matrices) still work; gathered A rows (a_rowmap)
  // are offsets from p.A itself and need the whole A below 4 GB (checked on the host).
  unsigned src[4];
  const char* baseA = (const char*)p.A;
  const char* baseB = (const char*)p.B;
  auto setup = [&](const Tile& t_in) {
    Tile t = t_in;
    NT_X(if (g_nt_dbg_skip & 64) { t.m0 = 0; t.n0 = 0; })      // experiment: every workgroup fetches tile 0's operands (all L2 hits)
    const bool gather = GROUPED && p.a_rowmap;
    baseA = (const char*)(p.A + (gather ? 0ll : (long long)t.m0 * p.lda));
    baseB = (const char*)(p.B + (GROUPED ? (long long)t.group * p.strideB : 0ll) + (long long)t.n0 * p.ldb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int line = (i * 8 + wid) * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ (line & 7);
      const int row = 2 * line + (lc >> 2);       // 0..255 A rows, 256..511 B rows
      if (i < 2) {
        const int m = min(t.m0 + row, t.m_end - 1);
        const int r = gather ? p.a_rowmap[m] : m - t.m0;
        src[i] = (unsigned)r * (unsigned)(p.lda * 2) + (lc & 3) * 16;
      } else src[i] = (unsigned)min(row - 256, p.N - 1 - t.n0) * (unsigned)(p.ldb * 2) + (lc & 3) * 16;
    }
  };
  auto stage = [&](int buf, int k0) {
    char* sb = smem + buf * SUB3 + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR((i < 2 ? baseA : baseB) + k0 * 2 + src[i]), LDS_PTR(sb + i * 8192), 16, 0, 0);
  };
  f32x4_t acc[8][4];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  const int sig = ((((frag_row >> 2) & 1) << 1 | (frag_row >> 3)) << 2) | (frag_row & 3);   // sigma(frag_row), see nt_epilogue
  const int offA = (wm * 128) * 64 + (frag_row >> 1) * 128 + (((((frag_row & 1) << 2) | frag_q) ^ ((frag_row >> 1) & 7)) << 4);
  const int offB = (256 + wn * 64) * 64 + (sig >> 1) * 128 + (((((sig & 1) << 2) | frag_q) ^ ((sig >> 1) & 7)) << 4);
  bf16x8_t af[8], bf[4];
  auto read_frags = [&](int buf) __attribute__((always_inline)) {
    const char* sA = smem + buf * SUB3 + offA;
    const char* sB = smem + buf * SUB3 + offB;
#pragma unroll
    for (int t = 0; t < 8; ++t) af[t] = *(const bf16x8_t*)(sA + t * 1024);
#pragma unroll
    for (int t = 0; t < 4; ++t) bf[t] = *(const bf16x8_t*)(sB + t * 1024);
  };
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < 8; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tm][tn], 0, 0, 0);
  };
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) -> int {
    int n = 0;
    n += nt_epilogue<false, 0, 8>(p, acc, t.m0 + wm * 128, t.m_end, t.n0 + wn * 64, t.group, frag_row, frag_q, nullptr);
    n += nt_epilogue<false, 4, 8>(p, acc, t.m0 + wm * 128 + 64, t.m_end, t.n0 + wn * 64, t.group, frag_row, frag_q, nullptr);
    return n;
  };

  if (my >= total) return;
  const int my_tiles = (total - my + G - 1) / G;
  Tile ct = decode(my);
  setup(ct);
  // vmcnt bookkeeping: the wait is for the wave's THIRD-newest sub-stage; younger than it are s2 stores,
  // d1 pieces, s1 stores, d0 pieces, s0 stores (issue order)
  int s3 = 0, s2 = 0, s1 = 0, s0 = 0, d2 = 0, d1 = 0, d0 = 0, wb = 0, rb = 0;
  int lid = my, lk = 0;
  bool lmore = true;
  auto issue = [&]() __attribute__((always_inline)) {
    s3 = s2; s2 = s1; s1 = s0; s0 = 0; d2 = d1; d1 = d0; d0 = 0;
    if (lmore) {
      stage(wb, lk * 32);
      d0 = 4;
      wb = (wb == NRING - 1) ? 0 : wb + 1;
      if (++lk == nu) {
        lk = 0; lid += G;
        if (lid < total) { const Tile lt = decode(lid); setup(lt); } else lmore = false;
      }
    }
  };
  auto wait_third_newest = [&]() { wait_vmcnt_ring4(__builtin_amdgcn_readfirstlane(NRING == 5 ? s3 + d2 + s2 + d1 + s1 + d0 + s0 : NRING == 3 ? s1 + d0 + s0 : s2 + d1 + s1 + d0 + s0)); };
  const int grp = __builtin_amdgcn_readfirstlane(wid >> 2);
  zero_acc();
  for (int i = 0; i < NRING - 1; ++i) issue();    // sub-stages 0 .. NRING-2 (the host guarantees K >= 128)
  wait_third_newest();
  __builtin_amdgcn_s_barrier();                   // sub-stage 0 landed for every wave
  asm volatile("" ::: "memory");
  NT_T(long long t_wait = 0, t_bar = 0, t_load = 0, t_comp = 0, t_epi = 0; const long long t_begin = nt_clk();)
  auto seg_barrier = [&]() {
    NT_T(const long long b0 = nt_clk();)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    NT_T(t_bar += nt_clk() - b0;)
  };
  // Tile seam: group 1 runs its epilogue right after its last COMPUTE, before the barrier; group 0 runs its
  // own AFTER that barrier - i.e. during group 1's last COMPUTE + epilogue.  Both epilogues (store issue,
  // operand-load latency) then overlap each other instead of each idling the other group.  One copy of the
  // epilogue code serves both: the loop body is rotated to start at COMPUTE, and the barrier sits before the
  // epilogue for group 0 and after it for group 1 (the GELU epilogue alone is 20 KB of a 64 KB I-cache).
  auto load_seg = [&]() __attribute__((always_inline)) {
    NT_T(const long long c0 = nt_clk();)
    read_frags(rb);
    rb = (rb == NRING - 1) ? 0 : rb + 1;
    issue();
    NT_T(const long long c1 = nt_clk(); t_load += c1 - c0;)
    if (grp == 1) { wait_third_newest(); NT_T(t_wait += nt_clk() - c1;) }
    seg_barrier();
  };
  auto body = [&](bool last) __attribute__((always_inline)) {
    NT_T(const long long c2 = nt_clk();)
    compute();
    NT_T(const long long c3 = nt_clk(); t_comp += c3 - c2;)
    if (grp == 0) { wait_third_newest(); NT_T(t_wait += nt_clk() - c3;) seg_barrier(); }
    if (last) {
      NT_T(const long long c4 = nt_clk();)
      s0 += __builtin_amdgcn_readfirstlane(epilogue(ct));
      zero_acc();
      NT_T(t_epi += nt_clk() - c4;)
    }
    if (grp == 1) seg_barrier();
    load_seg();                                   // the next sub-step's (past the end: reads nothing that is used)
  };
  if (grp == 1) seg_barrier();
  load_seg();
  for (int ti = 0; ti < my_tiles; ++ti) {
    for (int k = 1; k < nu; ++k) body(false);
    body(true);
    if (ti + 1 < my_tiles) ct = decode(my + (ti + 1) * G);
  }
  if (grp == 0) seg_barrier();
  NT_T(if ((tid & 63) == 0) { unsigned long long* g = g_nt_timing + wid * 8; atomicAdd(&g[0], (unsigned long long)(nt_clk() - t_begin));
         atomicAdd(&g[1], (unsigned long long)t_wait); atomicAdd(&g[2], (unsigned long long)t_bar); atomicAdd(&g[3], (unsigned long long)t_load);
         atomicAdd(&g[4], (unsigned long long)t_comp); atomicAdd(&g[5], (unsigned long long)t_epi); atomicAdd(&g[6], 1ull); })
}


// ---------------------------------------------------------------------------------------------
// gemm_nt4w: a 256x256 output tile run by FOUR waves (one per SIMD, 512 registers each) that own 128x128 of it: 256
// accumulator registers, two fragment sets of 16 x 16 bytes.  A 128x128 wave tile reads 16 KB of LDS per 64 MFMAs where
// the 128x64 tile of the eight-wave gemm_nt512 reads 12 KB per 32: LDS traffic per k = 32 falls from 96 KB + 32 KB of
// DMA writes (the whole 128 B/clk the LDS has during the 1024 clk the MFMAs take) to 64 KB + 32 KB.
// Stages are k = 64: 512 rows (256 A + 256 B) of 128 bytes, XOR-swizzled as in gemm_nt256, two buffers of 64 KB.  Every
// DMA instruction moves whole 128-byte lines (8 lanes per row); the k = 32 sub-stages of gemm_nt512 fetch half lines,
// twice the L1 tag work per byte, and that was what each DMA cost the issuing wave (measured: without DMA 1575, with
// 1219 TFLOP/s on random data).
// With one wave per SIMD nothing else fills the matrix pipe, and a wave issues one instruction per four clocks - three
// besides the MFMA per 16-clk MFMA slot - so every wave interleaves by hand.  Iteration s (stage s in buffer s & 1,
// set X = its k-half 0 fragments on entry), 128 MFMAs:
//   MFMA   0.. 15 : 16 reads of k-half 1 -> set Y (one per MFMA)
//   MFMA  21      : lgkmcnt(0), barrier 1 - every wave has everything it needs from buffer s & 1
//   MFMA  23.. 98 : the 16 DMA pieces of stage s+2 -> buffer s & 1 (one per 5 MFMAs); then the scalar bookkeeping
//   MFMA 101      : counted vmcnt for stage s+1 (issued one iteration ago), barrier 2
//   MFMA 102..117 : 16 reads of stage s+1's k-half 0 -> set X;  lgkmcnt(0) after MFMA 125
// The epilogue of a tile's last stage follows MFMA 127 (its stores are counted into the next vmcnt wait).
// ---------------------------------------------------------------------------------------------
// gemm_nt4w epilogue for the builds that READ a tile - C = acc (+ bias) + residual, or C = acc x aux (the stored GELU') - on tiles
// inside the matrix.  nt_epilogue loads that operand in the accumulator layout (16 rows x 32 bytes per instruction) block by block,
// each block's loads BEHIND the previous block's stores: vmcnt returns in issue order and hipcc waits with vmcnt(0) once loads and stores
// are mixed, so every block paid a store acknowledgement plus a load latency (30k clocks per tile against 8k for the plain epilogue).
// Here the fp32 accumulators of 32 rows x 64 columns go through the wave's 8-KB LDS region (16-byte slots XOR-swizzled by row & 15,
// conflict-free both ways) and come back row-major; the operand is loaded in that shape too - whole 128-byte lines, 16 bytes per lane -
// by inline-asm loads one half block AHEAD (16 registers per set) and consumed behind a COUNTED wait that leaves the stores and the next
// set's loads in flight.  Same arithmetic in the same order as nt_epilogue (one rounding): bit-identical results.
template <bool MUL>
__device__ __forceinline__ void nt4w_rows_load(const GemmNTArgs& p, u32x4e_t (&r)[4], int h, int mb, int nb, int rl, int c8) {
  const bf16_t* src = MUL ? (const bf16_t*)p.aux : p.residual;
  const long long ld = MUL ? p.ldaux : p.ldr;
  const int tmh = (h & 3) * 2, tnq = (h >> 2) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16_t* a = src + (long long)(mb + tmh * 16 + i * 8 + rl) * ld + nb + tnq * 16 + c8 * 8;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(a));
  }
}
// half block H (rows (H & 3) * 32 .., columns (H >> 2) * 64 .. of the wave tile): accumulators (+ bias) -> LDS -> row-major, operand set r
// behind a wait that leaves WAIT younger vector-memory operations in flight, combine, store.  No closure over acc: the accumulator array
// has to stay in registers through every inlined copy.
template <int SPEC, int H, int WAIT>
__device__ __forceinline__ void nt4w_rows_half(const GemmNTArgs& p, f32x4_t (&acc)[8][8], const float4 (&b4)[8], u32x4e_t (&r)[4], int mb,
                                               int nb, int frag_row, int pg, int rl, int c8, unsigned lx) {
  constexpr bool MUL = (SPEC & 7) == EPI_MUL_AUX;
  constexpr int tmh = (H & 3) * 2, tnq = (H >> 2) * 4;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int row_l = t * 16 + frag_row, slot = tn * 4 + pg;
      const unsigned a = lx + row_l * 256 + ((slot ^ (row_l & 15)) << 4);
      // acc + b in VGPRs ("v"): an accumulator register as a direct asm operand - "v" or "a" - makes the allocator copy and spill
      // accumulator tuples; without a bias b is an opaque zero
      const float4 b = b4[tnq + tn];
      const f32x4_t v = {acc[tmh + t][tnq + tn][0] + b.x, acc[tmh + t][tnq + tn][1] + b.y, acc[tmh + t][tnq + tn][2] + b.z, acc[tmh + t][tnq + tn][3] + b.w};
      asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory");
    }
  f32x4_t lo[4], hi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row_l = i * 8 + rl;
    const unsigned a0 = lx + row_l * 256 + (((2 * c8) ^ (row_l & 15)) << 4), a1 = lx + row_l * 256 + (((2 * c8 + 1) ^ (row_l & 15)) << 4);
    asm volatile("ds_read_b128 %0, %1" : "=v"(lo[i]) : "v"(a0));
    asm volatile("ds_read_b128 %0, %1" : "=v"(hi[i]) : "v"(a1));
  }
  // the waits carry every register the loads and LDS reads write as in/out operands: no use (not even a copy) can be scheduled above them
#define NT4R_PINS "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
  constexpr bool OPERAND = MUL || ((SPEC & 16) != 0);       // else: nothing is read (bias + GELU + GELU')
  constexpr bool GELU = (SPEC & 7) == EPI_GELU_DAUX;
  // (with an operand every use of lo / hi is an operation with an r value, i.e. behind the wait already; pinning lo / hi there as well costs
  // the bias + residual build seven spilled registers)
  if constexpr (!OPERAND) asm volatile("s_waitcnt lgkmcnt(0)" : NT4R_PINS :: "memory");
  else if constexpr (WAIT == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
  else asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) :: "memory");
#undef NT4R_PINS
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float f[8] = {lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2], hi[i][3]};
    float o[8], dg[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (GELU) {
        f32x2_t g_, d_;
        gelu_and_grad2((f32x2_t){f[2 * k], f[2 * k + 1]}, g_, d_);
        o[2 * k] = g_[0]; o[2 * k + 1] = g_[1]; dg[2 * k] = d_[0]; dg[2 * k + 1] = d_[1];
      } else {
        const float z0 = __uint_as_float(r[i][k] << 16), z1 = __uint_as_float(r[i][k] & 0xffff0000u);
        o[2 * k] = MUL ? f[2 * k] * z0 : f[2 * k] + z0;
        o[2 * k + 1] = MUL ? f[2 * k + 1] * z1 : f[2 * k + 1] + z1;
      }
    }
    const u32x4e_t pk = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    __builtin_nontemporal_store(pk, (u32x4e_t*)((bf16_t*)p.C + (long long)(mb + tmh * 16 + i * 8 + rl) * p.ldc + nb + tnq * 16 + c8 * 8));
    if constexpr (GELU) {
      const u32x4e_t pd = {pack2bf(dg[0], dg[1]), pack2bf(dg[2], dg[3]), pack2bf(dg[4], dg[5]), pack2bf(dg[6], dg[7])};
      __builtin_nontemporal_store(pd, (u32x4e_t*)(p.aux + (long long)(mb + tmh * 16 + i * 8 + rl) * p.ldaux + nb + tnq * 16 + c8 * 8));
    }
  }
}
template <int SPEC>
__device__ __forceinline__ int nt4w_epilogue_rows(const GemmNTArgs& p, f32x4_t (&acc)[8][8], int mb, int nb, int frag_row, int frag_q,
                                                  unsigned lx) {
  constexpr bool MUL = (SPEC & 7) == EPI_MUL_AUX;
  // the LDS addresses below depend on the lane only: left to itself the compiler computes all of them once, outside the tile loop, and
  // carries ~40 registers through the k-loop (first build: 281 spilled registers).  An opaque copy of the lane id keeps them here.
  asm volatile("" : "+v"(frag_row), "+v"(frag_q));
  const int lane = frag_q * 16 + frag_row, rl = lane >> 3, c8 = lane & 7;
  const int pg = ((frag_q & 1) << 1) | (frag_q >> 1);
  float4 b4[8];
  float zero = 0.f;
  asm volatile("" : "+v"(zero));
#pragma unroll
  for (int tn = 0; tn < 8; ++tn) b4[tn] = (SPEC & 8) ? *(const float4*)(p.bias + nb + tn * 16 + pg * 4) : make_float4(zero, zero, zero, zero);
  u32x4e_t ra[4], rb[4];
  constexpr bool OPERAND = MUL || ((SPEC & 16) != 0);
#define NT4R_LOAD(R, H) do { if constexpr (OPERAND) nt4w_rows_load<MUL>(p, R, H, mb, nb, rl, c8); } while (0)
#define NT4R_HALF(R, H, W) nt4w_rows_half<SPEC, H, W>(p, acc, b4, R, mb, nb, frag_row, pg, rl, c8, lx)
  NT4R_LOAD(ra, 0);
  NT4R_LOAD(rb, 1); NT4R_HALF(ra, 0, 4);
  NT4R_LOAD(ra, 2); NT4R_HALF(rb, 1, 8);
  NT4R_LOAD(rb, 3); NT4R_HALF(ra, 2, 8);
  NT4R_LOAD(ra, 4); NT4R_HALF(rb, 3, 8);
  NT4R_LOAD(rb, 5); NT4R_HALF(ra, 4, 8);
  NT4R_LOAD(ra, 6); NT4R_HALF(rb, 5, 8);
  NT4R_LOAD(rb, 7); NT4R_HALF(ra, 6, 8);
  NT4R_HALF(rb, 7, 4);
#undef NT4R_LOAD
#undef NT4R_HALF
  return ((SPEC & 7) == EPI_GELU_DAUX) ? 64 : 32;
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_seq(F& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F& f) { static_for_seq(f, std::make_integer_sequence<int, N>{}); }
#define DS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#define STAGE4 (512 * 128)
#ifndef NT4_DSTEP
#define NT4_DSTEP 5      // MFMAs per DMA piece (2: -4 %, 3: -2 % against 5; the TA path wants them spread out)
#endif
#ifndef NT4_RS
#define NT4_RS 1         // MFMAs per fragment read in the first phase
#endif
#define NT4_B1 (16 * NT4_RS + 5)    // MFMA after which barrier 1 sits
#ifndef NT4_B2
#define NT4_B2 101                  // MFMA after which the vmcnt wait + barrier 2 sit
#endif
#ifndef NT4_RS3
#define NT4_RS3 1                   // MFMAs per fragment read in the last phase
#endif
static_assert(NT4_B1 + 2 + 15 * NT4_DSTEP + 2 < NT4_B2, "the DMA pieces must be out before the vmcnt wait");
static_assert(NT4_B2 + 16 * NT4_RS3 < 125, "the last fragment reads need time to land");
__device__ __forceinline__ void wait_vmcnt_4w(int n) {
  if (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (n >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
  else if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
  else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// GROUPED: tiles come from the device-side table p.tiles (group, m0, m_end, -) of 256-row tiles, B / bias are per group,
// A rows are gathered through p.a_rowmap and C rows scattered through p.c_rowmap (expert GEMMs, dGm) - as in gemm_nt512.
template <int SPEC, bool GROUPED = false>
__global__ __launch_bounds__(256) void gemm_nt4w_kernel(GemmNTArgs p_in) {
  GemmNTArgs p = p_in;
  if (!GROUPED && p.m_dev) { p.M = min(*p.m_dev, p.M); p.max_tiles_m = (p.M + 255) / 256; }
#ifdef NT4_SKIP
  constexpr int dbg4 = NT4_SKIP;      // compile-time ablation (tools/nt_timing.hip -DNT_EXPERIMENT -DNT4_SKIP=bits): 1 reads, 8 DMA, 16 barriers
#else
  constexpr int dbg4 = 0;
#endif
  if constexpr (SPEC >= 0) {
    p.epi = SPEC & 7; p.out_f32 = (SPEC >> 6) & 1; p.col_perm = 0; p.alpha = 1.f;
    if (!GROUPED) { p.c_rowmap = nullptr; p.a_rowmap = nullptr; }
    if (!(SPEC & 8)) p.bias = nullptr;
    if (!(SPEC & 16)) p.residual = nullptr;
    if (!(SPEC & 32)) p.aux = nullptr;
    __builtin_assume((p.N & 7) == 0);
    if (SPEC & 8) __builtin_assume(p.bias != nullptr);
    if (SPEC & 16) __builtin_assume(p.residual != nullptr);
    if (SPEC & 32) __builtin_assume(p.aux != nullptr);
  }
  // the plain builds with bf16 output store through a wave-private 8-KB LDS region behind the ring (nt_epilogue LDSX)
#ifdef NT4_NO_XLDS
  constexpr bool XLDS = false;            // A/B builds (tools): the accumulator-layout stores
#else
  constexpr bool XLDS = !GROUPED && SPEC >= 0 && !(SPEC & 64);
#endif
  // builds whose epilogue READS a tile (residual add, x stored GELU') or writes two (GELU + GELU'): nt4w_epilogue_rows on tiles inside the
  // matrix; the next tile's first fragments are then read AFTER the epilogue, which needs their 64 registers
  constexpr bool OPER = XLDS && (((SPEC & 16) != 0 && (SPEC & 7) == EPI_NONE) || (SPEC & 7) == EPI_MUL_AUX || ((SPEC & 7) == EPI_GELU_DAUX && (SPEC & 32) != 0));
  __shared__ __attribute__((aligned(128))) char smem[2 * STAGE4 + (XLDS ? 32768 : 0)];      // 128: the k-half switch is an XOR of the byte address
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA bases stay scalar
  const int wm = wid & 1, wn = wid >> 1;
  const int frag_row = lane & 15, frag_q = lane >> 4;
  const int G = gridDim.x;
  const int my = xcd_remap(blockIdx.x, G);
  const int total = GROUPED ? *p.tile_count * p.n_tiles_n : p.max_tiles_m * p.n_tiles_n;
  const int nk = p.K / 64;                        // stages per tile
  constexpr int SM = 4, SN = 8;
  struct Tile { int m0, n0, m_end, group; };
  auto decode = [&](int id) -> Tile {             // tile order of gemm_nt512 (row tiles last to first)
    Tile t;
    if constexpr (GROUPED) {                      // n-tiles of one m-tile are consecutive (they share the gathered A rows)
      const int tm = id / p.n_tiles_n;
      const int4 e = p.tiles[tm];
      t.group = e.x; t.m0 = e.y; t.m_end = e.z; t.n0 = (id - tm * p.n_tiles_n) * 256;
      return t;
    }
    t.group = 0; t.m_end = p.M;
    const int per_super = SM * p.n_tiles_n;
    const int sg = id / per_super, r = id - sg * per_super;
    const int rows = min(SM, p.max_tiles_m - sg * SM);
    const int blk = SN * rows;
    const int nfull = p.n_tiles_n / SN;
    int tile_m, tile_n;
    if (r < nfull * blk) { const int nb = r / blk, w = r - nb * blk; tile_n = nb * SN + (w % SN); tile_m = sg * SM + (w / SN); }
    else {
      const int r2 = r - nfull * blk, nc = p.n_tiles_n - nfull * SN;
      tile_n = nfull * SN + r2 % nc; tile_m = sg * SM + r2 / nc;
    }
    t.m0 = (p.max_tiles_m - 1 - tile_m) * 256; t.n0 = tile_n * 256;
    return t;
  };
  // DMA piece i (0..15) moves 32 rows: thread t fills LDS slot q = i * 256 + t (16 bytes) = row q >> 3, position q & 7,
  // which holds the row's 16-byte chunk (q & 7) ^ (row & 7).  Per-lane sources are 32-bit byte offsets from the tile's
  // A / B base (one tile's rows span 256 * ld * 2 bytes < 4 GB, checked on the host).
  unsigned src[16];
  const char* baseA = (const char*)p.A;
  const char* baseB = (const char*)p.B;
  const unsigned r0 = tid >> 3, c16 = ((tid & 7) ^ (r0 & 7)) * 16;   // i * 32 leaves row & 7 alone: one chunk index for all pieces
  auto setup = [&](const Tile& t) {
    const bool gather = GROUPED && p.a_rowmap;                                 // gathered rows: offsets from A itself (whole A < 4 GB, host check)
    baseA = (const char*)(p.A + (gather ? 0ll : (long long)t.m0 * p.lda));
    baseB = (const char*)(p.B + (GROUPED ? (long long)t.group * p.strideB : 0ll) + (long long)t.n0 * p.ldb);
    const unsigned limA = t.m_end - 1 - t.m0, limB = p.N - 1 - t.n0;          // last valid row of the tile (rows past it re-read it)
    const unsigned ldaB = p.lda * 2, ldbB = p.ldb * 2;                         // < 2^24 (host check): 24-bit multiplies
    unsigned r0v = r0;
    if constexpr (OPER) asm volatile("" : "+v"(r0v));      // i * 32 + r0 recomputed here: hoisted out of the tile loop they were spilled
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned ra = min(i * 32 + r0v, limA);
      if (gather) src[i] = (unsigned)p.a_rowmap[t.m0 + ra] * ldaB + c16;
      else src[i] = __umul24(ra, ldaB) + c16;
      src[8 + i] = __umul24(min(i * 32 + r0v, limB), ldbB) + c16;
    }
  };
  f32x4_t acc[8][8];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  const int sig = ((((frag_row >> 2) & 1) << 1 | (frag_row >> 3)) << 2) | (frag_row & 3);   // sigma(frag_row), see nt_epilogue
  const unsigned lds0 = (unsigned)(size_t)smem;   // LDS byte address of buffer 0
  // fragment addresses in buffer 0 for k-half 0; k-half 1 is the same address ^ 64, buffer 1 is + STAGE4
  const unsigned offA = lds0 + (wm * 128 + frag_row) * 128 + ((frag_q ^ (frag_row & 7)) << 4);
  const unsigned offB = lds0 + (256 + wn * 128 + sig) * 128 + ((frag_q ^ (sig & 7)) << 4);

  if (my >= total) return;
  const int my_tiles = (total - my + G - 1) / G;
  Tile ct = decode(my);
  setup(ct);
  // the tile / k position the DMA is at (two stages ahead of the MFMAs); past the last tile the addresses stay where
  // they are (valid memory, the data is never used).  gA / gB / ldsW: scalar bases of the stage being issued, advanced
  // in an idle stretch of the iteration, never next to a DMA.
  int lid = my, lk = 0, wb = 0;
  const char* gA = baseA;
  const char* gB = baseB;
  unsigned ldsW = lds0 + wid * 1024;
  auto advance = [&]() __attribute__((always_inline)) {
    wb ^= 1;
    if (++lk == nk) {
      lk = 0; lid += G;
      if (lid < total) { const Tile lt = decode(lid); setup(lt); }
    }
    gA = baseA + lk * 128; gB = baseB + lk * 128;
    ldsW = lds0 + wb * STAGE4 + wid * 1024;
  };
  auto piece = [&](int i) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds(GLB_PTR((i < 8 ? gA : gB) + src[i]), (__attribute__((address_space(3))) void*)(size_t)(ldsW + i * 4096), 16, 0, 0);
  };
  int s_prev = 0;                                 // C / aux stores issued by the previous iteration's epilogue
  NT_T(long long t_wait = 0, t_bar = 0, t_comp = 0, t_epi = 0; const long long t_begin = nt_clk();)
  int rbuf = 0;                                   // buffer of the stage being computed
  bf16x8_t fa0[8], fb0[8], fa1[8], fb1[8];
  auto iteration = [&](bool last) __attribute__((always_inline)) {
    const unsigned aA1 = (offA + rbuf * STAGE4) ^ 64u, aB1 = (offB + rbuf * STAGE4) ^ 64u;          // this stage, k-half 1
    const unsigned aA0 = offA + (rbuf ^ 1) * STAGE4, aB0 = offB + (rbuf ^ 1) * STAGE4;              // next stage, k-half 0
    auto one = [&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value, h = q >> 6, tm = (q >> 3) & 7, tn = q & 7;
      if constexpr (h == 0) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[tn], fa0[tm], acc[tm][tn], 0, 0, 0);
      else acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[tn], fa1[tm], acc[tm][tn], 0, 0, 0);
      if constexpr (q < 16 * NT4_RS && (q % NT4_RS) == NT4_RS - 1) {   // k-half 1 of this stage -> set Y
        constexpr int g = q / NT4_RS;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(dbg4 & 1)) { if constexpr (g < 8) DS_READ128(fa1[g], aA1, g * 2048); else DS_READ128(fb1[g - 8], aB1, (g - 8) * 2048); }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (q == NT4_B1) {
        NT_T(const long long c0 = nt_clk();)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (!(dbg4 & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        NT_T(t_bar += nt_clk() - c0;)
      }
      if constexpr (q >= NT4_B1 + 2 && q < NT4_B1 + 2 + 16 * NT4_DSTEP && (q - NT4_B1 - 2) % NT4_DSTEP == 0) {      // stage s+2 -> the buffer just released
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(dbg4 & 8)) piece((q - NT4_B1 - 2) / NT4_DSTEP);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (q == NT4_B1 + 2 + 15 * NT4_DSTEP + 2) advance();
      if constexpr (q == NT4_B2) {
        NT_T(const long long c0 = nt_clk();)
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt_4w(__builtin_amdgcn_readfirstlane(s_prev + 16));
        NT_T(const long long c1 = nt_clk(); t_wait += c1 - c0;)
        if constexpr (!(dbg4 & 16)) __builtin_amdgcn_s_barrier();
        NT_T(t_bar += nt_clk() - c1;)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
      }
      if constexpr (q > NT4_B2 && q <= NT4_B2 + 16 * NT4_RS3 && ((q - NT4_B2 - 1) % NT4_RS3) == 0) {        // next stage's k-half 0 -> set X (its MFMAs are done)
        constexpr int g = (q - NT4_B2 - 1) / NT4_RS3;
        __builtin_amdgcn_sched_barrier(0);
        if (!(OPER && last)) {
          if constexpr (!(dbg4 & 1)) { if constexpr (g < 8) DS_READ128(fa0[g], aA0, g * 2048); else DS_READ128(fb0[g - 8], aB0, (g - 8) * 2048); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (q == 125) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    NT_T(const long long i0 = nt_clk();)
    static_for<128>(one);
    NT_T(t_comp += nt_clk() - i0;)
    rbuf ^= 1;
    s_prev = 0;
    if (last) {
      int n = 0;
      NT_T(const long long e0 = nt_clk();)
      if constexpr (!(dbg4 & 4)) {
      const unsigned lx = lds0 + 2 * STAGE4 + wid * 8192;
      if (OPER && ct.m0 + 256 <= ct.m_end && ct.n0 + 256 <= p.N) {
        if constexpr (OPER) n += nt4w_epilogue_rows<SPEC>(p, acc, ct.m0 + wm * 128, ct.n0 + wn * 128, frag_row, frag_q, lx);
      } else {
      n += nt_epilogue<false, 0, 8, 0, 8, XLDS>(p, acc, ct.m0 + wm * 128, ct.m_end, ct.n0 + wn * 128, ct.group, frag_row, frag_q, nullptr, lx);
      n += nt_epilogue<false, 0, 8, 4, 8, XLDS>(p, acc, ct.m0 + wm * 128, ct.m_end, ct.n0 + wn * 128 + 64, ct.group, frag_row, frag_q, nullptr, lx);
      n += nt_epilogue<false, 4, 8, 0, 8, XLDS>(p, acc, ct.m0 + wm * 128 + 64, ct.m_end, ct.n0 + wn * 128, ct.group, frag_row, frag_q, nullptr, lx);
      n += nt_epilogue<false, 4, 8, 4, 8, XLDS>(p, acc, ct.m0 + wm * 128 + 64, ct.m_end, ct.n0 + wn * 128 + 64, ct.group, frag_row, frag_q, nullptr, lx);
      }
      }
      s_prev = __builtin_amdgcn_readfirstlane(n);
      zero_acc();
      if constexpr (OPER) {                        // the fragment reads this iteration skipped (rbuf already points at the next stage)
        const unsigned nA0 = offA + rbuf * STAGE4, nB0 = offB + rbuf * STAGE4;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) DS_READ128(fa0[t], nA0, t * 2048);
#pragma unroll
        for (int t = 0; t < 8; ++t) DS_READ128(fb0[t], nB0, t * 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      NT_T(t_epi += nt_clk() - e0;)
    }
  };

  zero_acc();
  for (int v = 0; v < 2; ++v) {                   // stages 0 and 1 (the host guarantees K >= 128)
#pragma unroll
    for (int i = 0; i < 16; ++i) piece(i);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // stage 0 landed for every wave
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t = 0; t < 8; ++t) DS_READ128(fa0[t], offA, t * 2048);
#pragma unroll
  for (int t = 0; t < 8; ++t) DS_READ128(fb0[t], offB, t * 2048);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  for (int ti = 0; ti < my_tiles; ++ti) {
    for (int k = 1; k < nk; ++k) iteration(false);
    iteration(true);
    if (ti + 1 < my_tiles) ct = decode(my + (ti + 1) * G);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA ran two stages past the end
  NT_T(if (lane == 0) { unsigned long long* g = g_nt_timing + wid * 8; atomicAdd(&g[0], (unsigned long long)(nt_clk() - t_begin));
         atomicAdd(&g[1], (unsigned long long)t_wait); atomicAdd(&g[2], (unsigned long long)t_bar); atomicAdd(&g[4], (unsigned long long)t_comp);
         atomicAdd(&g[5], (unsigned long long)t_epi); atomicAdd(&g[6], 1ull); })
}

static int gemm_nt_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        int M, int N, int K, const float* bias, const void* residual, int ldr,
                        void* aux, int ldaux, const int* a_rowmap, const int* c_rowmap,
                        const int* tiles, const int* tile_count, int max_tiles,
                        long long strideB, long long strideBias, float alpha, int epi,
                        int out_f32, int col_perm, const int* m_dev, hipStream_t stream);

extern "C" int medmoe_gemm_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                              int M, int N, int K, const float* bias, const void* residual, int ldr,
                              void* aux, int ldaux, const int* a_rowmap, const int* c_rowmap,
                              const int* tiles, const int* tile_count, int max_tiles,
                              long long strideB, long long strideBias, float alpha, int epi,
                              int out_f32, int col_perm, hipStream_t stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, aux, ldaux, a_rowmap, c_rowmap, tiles, tile_count, max_tiles,
                      strideB, strideBias, alpha, epi, out_f32, col_perm, nullptr, stream);
}

// The same product on the first *m_dev rows only (m_dev: device int, 1 <= *m_dev <= M; M sizes the launch): the packed rows of a
// variable-length batch (the text tower's non-padding tokens) without a device-to-host copy of the count.  Plain operands only.
extern "C" int medmoe_gemm_nt_rows(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                                   const void* residual, int ldr, void* aux, int ldaux, float alpha, int epi, int out_f32, const int* m_dev,
                                   hipStream_t stream) {
  if (!m_dev) return MM_ERR_ARG;
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, aux, ldaux, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, alpha, epi,
                      out_f32, 0, m_dev, stream);
}

static int gemm_nt_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                        int M, int N, int K, const float* bias, const void* residual, int ldr,
                        void* aux, int ldaux, const int* a_rowmap, const int* c_rowmap,
                        const int* tiles, const int* tile_count, int max_tiles,
                        long long strideB, long long strideBias, float alpha, int epi,
                        int out_f32, int col_perm, const int* m_dev, hipStream_t stream) {
  if (!A || !B || !C) return MM_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (K % 32) != 0 || (N % 4) != 0) return MM_ERR_SHAPE;      // K % 64 == 32: the 128x128 kernel only
  if ((lda % 8) || (ldb % 8) || (ldc % 4) || (residual && (ldr % 4)) || (aux && (ldaux % 4))) return MM_ERR_SHAPE;
  if (epi < 0 || epi > EPI_MUL_AUX) return MM_ERR_ARG;
  if ((epi == EPI_MUL_DGELU || epi == EPI_MUL_DRELU || epi == EPI_MUL_AUX) && !aux) return MM_ERR_ARG;
  if (tiles && (!tile_count || max_tiles <= 0)) return MM_ERR_ARG;
  GemmNTArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.bias = bias; p.residual = (const bf16_t*)residual; p.aux = (bf16_t*)aux;
  p.a_rowmap = a_rowmap; p.c_rowmap = c_rowmap; p.tiles = (const int4*)tiles; p.tile_count = tile_count;
  p.strideB = strideB; p.strideBias = strideBias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux;
  p.n_tiles_n = (N + BN - 1) / BN;
  p.alpha = alpha; p.epi = epi; p.out_f32 = out_f32; p.col_perm = col_perm; p.m_dev = m_dev;
  const bool fits32 = (long long)M * lda * 2 < (1ll << 32) && (long long)N * ldb * 2 < (1ll << 32);   // 32-bit DMA offsets
  const bool plain = !tiles && !a_rowmap && !c_rowmap && !col_perm && M >= 4 * BM2 && g_use_nt256 && (K % BK) == 0;
  const bool big = plain && K >= 3 * BK && fits32;
  // 256x256 tiles: wide N always; N of two or three tiles when K is long enough to amortise the seam or the tile
  // count fills the chip evenly (measured: N=768 K=3072 858 -> ~1150 TF/s; N=768 K=768 slower at 591 tiles).  Offsets are
  // tile-relative there, so only one tile's rows (256 * ld * 2 bytes) have to fit 32 bits.
  const bool tile32 = 256ll * lda * 2 < (1ll << 32) && 256ll * ldb * 2 < (1ll << 32);
  const bool pitch24 = 2ll * lda < (1ll << 24) && 2ll * ldb < (1ll << 24);        // gemm_nt4w multiplies row x pitch in 24 bits
  // 256x256 or 256x128 tiles: the big tile (gemm_nt4w) runs ~1.35x faster per flop, the small one wastes less of
  // the last round of 256 workgroups.  Compare fill x speed.
  const long long t512 = (long long)((M + 255) / 256) * ((N + 255) / 256), t256 = (long long)((M + 255) / 256) * ((N + 127) / 128);
  const double fill512 = (double)t512 / (double)(((t512 + 255) / 256) * 256), fill256 = (double)t256 / (double)(((t256 + 255) / 256) * 256);
  if (plain && tile32 && g_use_nt512 && K >= 128 && N >= 512 && (!big || fill512 * 1.35 >= fill256)) {
    p.max_tiles_m = (M + 255) / 256;
    p.n_tiles_n = (N + 255) / 256;
    const int grid = min(p.max_tiles_m * p.n_tiles_n, g_nt_max_grid);     // 1 resident block per CU (128 KB LDS)
    int spec = -1;
    if ((N & 7) == 0 && alpha == 1.f && epi != EPI_RELU && epi != EPI_MUL_DRELU) spec = NT_SPEC(epi, bias ? 1 : 0, residual ? 1 : 0, aux ? 1 : 0) | (out_f32 ? 64 : 0);
    if (g_use_nt4w && spec >= 0 && pitch24) {                 // four waves, 128x128 wave tiles (see gemm_nt4w_kernel): the same tiles, 5-15 % faster
      bool done = true;
      switch (spec) {
#define NT_CASE(s) case s: hipLaunchKernelGGL(gemm_nt4w_kernel<s>, dim3(grid), dim3(256), 0, stream, p); break;
        NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0))
        NT_CASE(NT_SPEC(EPI_NONE, 1, 0, 0))
        NT_CASE(NT_SPEC(EPI_NONE, 1, 1, 0))
        NT_CASE(NT_SPEC(EPI_NONE, 0, 1, 0))
        NT_CASE(NT_SPEC(EPI_GELU, 1, 0, 1))
        NT_CASE(NT_SPEC(EPI_GELU, 1, 0, 0))
        NT_CASE(NT_SPEC(EPI_MUL_DGELU, 0, 0, 1))
        NT_CASE(NT_SPEC(EPI_GELU_DAUX, 1, 0, 1))
        NT_CASE(NT_SPEC(EPI_MUL_AUX, 0, 0, 1))
        NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0) | 64)
#undef NT_CASE
        default: done = false; break;
      }
      if (done) { g_last_nt_kernel = 4; return mm_check_launch(); }
    }
    g_last_nt_kernel = 2;
    switch (spec) {
#define NT_CASE(s) case s: hipLaunchKernelGGL((gemm_nt512_kernel<s, false>), dim3(grid), dim3(512), 0, stream, p); break;
      NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0))
      NT_CASE(NT_SPEC(EPI_NONE, 1, 0, 0))
      NT_CASE(NT_SPEC(EPI_NONE, 1, 1, 0))
      NT_CASE(NT_SPEC(EPI_NONE, 0, 1, 0))
      NT_CASE(NT_SPEC(EPI_GELU, 1, 0, 1))
      NT_CASE(NT_SPEC(EPI_GELU, 1, 0, 0))
      NT_CASE(NT_SPEC(EPI_MUL_DGELU, 0, 0, 1))
      NT_CASE(NT_SPEC(EPI_GELU_DAUX, 1, 0, 1))     // FC1 forward, keeps GELU'(z)
      NT_CASE(NT_SPEC(EPI_MUL_AUX, 0, 0, 1))       // FC2 dgrad x stored GELU'
      NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0) | 64)     // fp32 C (local-loss context gradient)
#undef NT_CASE
      default: hipLaunchKernelGGL((gemm_nt512_kernel<-1, false>), dim3(grid), dim3(512), 0, stream, p); break;
    }
    return mm_check_launch();
  }
  if (big) {
    g_last_nt_kernel = 1;
    p.max_tiles_m = (M + BM2 - 1) / BM2;
    const int grid = min(p.max_tiles_m * p.n_tiles_n, 256);     // 1 resident block per CU (144 KB LDS)
    int spec = -1;
    if ((N & 7) == 0 && alpha == 1.f && epi != EPI_RELU && epi != EPI_MUL_DRELU) spec = NT_SPEC(epi, bias ? 1 : 0, residual ? 1 : 0, aux ? 1 : 0) | (out_f32 ? 64 : 0);
    switch (spec) {
#define NT_CASE(s) case s: hipLaunchKernelGGL(gemm_nt256_kernel<s>, dim3(grid), dim3(512), 0, stream, p); break;
      NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0))          // dgrad
      NT_CASE(NT_SPEC(EPI_NONE, 1, 0, 0))          // QKV
      NT_CASE(NT_SPEC(EPI_NONE, 1, 1, 0))          // out-proj, FC2, patch embedding: + bias + residual
      NT_CASE(NT_SPEC(EPI_NONE, 0, 1, 0))
      NT_CASE(NT_SPEC(EPI_GELU, 1, 0, 1))          // FC1 forward, keeps the pre-activation
      NT_CASE(NT_SPEC(EPI_GELU, 1, 0, 0))          // frozen text tower FC1
      NT_CASE(NT_SPEC(EPI_MUL_DGELU, 0, 0, 1))     // FC2 dgrad x GELU'
      NT_CASE(NT_SPEC(EPI_GELU_DAUX, 1, 0, 1))
      NT_CASE(NT_SPEC(EPI_MUL_AUX, 0, 0, 1))
#undef NT_CASE
      default: hipLaunchKernelGGL(gemm_nt256_kernel<-1>, dim3(grid), dim3(512), 0, stream, p); break;
    }
    return mm_check_launch();
  }
  if (g_use_nt_direct && !tiles && !a_rowmap && !c_rowmap && !col_perm && !m_dev && (K & 63) == 32 && K <= 288 && M >= 4096) {
    const long long waves = (long long)((M + 63) / 64) * ((N + 63) / 64);
    const dim3 dgrid((unsigned)((waves + 3) / 4));
    g_last_nt_kernel = 6;
    int spec = -1;
    if ((N & 7) == 0 && alpha == 1.f && !out_f32 && epi != EPI_RELU && epi != EPI_MUL_DRELU) spec = NT_SPEC(epi, bias ? 1 : 0, residual ? 1 : 0, aux ? 1 : 0);
    switch (spec) {
#define NT_CASE(s) case s: hipLaunchKernelGGL(gemm_nt_direct_kernel<s>, dgrid, dim3(256), 0, stream, p); break;
      NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0))          // dgrads
      NT_CASE(NT_SPEC(EPI_NONE, 1, 0, 0))          // QKV, o_proj under stochastic depth
      NT_CASE(NT_SPEC(EPI_NONE, 1, 1, 0))          // o_proj / FC2 + residual
      NT_CASE(NT_SPEC(EPI_GELU_DAUX, 1, 0, 1))     // FC1 forward, keeps GELU'(z)
      NT_CASE(NT_SPEC(EPI_MUL_AUX, 0, 0, 1))       // FC2 dgrad x stored GELU'
#undef NT_CASE
      default: hipLaunchKernelGGL(gemm_nt_direct_kernel<-1>, dgrid, dim3(256), 0, stream, p); break;
    }
    return mm_check_launch();
  }
  p.max_tiles_m = tiles ? max_tiles : (M + BM - 1) / BM;
  const int grid = min(p.max_tiles_m * p.n_tiles_n, 2 * 256);   // 2 resident blocks per CU (64 KB LDS each)
  g_last_nt_kernel = 0;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(grid), dim3(256), 0, stream, p);
  return mm_check_launch();
}

// --------------------------------------------------------------------------------------------
// gemm_tn (wgrad):  dW[g][n][k] += sum_m G[m][n] * X[xmap(m)][k],  db[g][n] += sum_m G[m][n]
// --------------------------------------------------------------------------------------------
struct GemmTNArgs {
  const bf16_t* G; const bf16_t* X; float* dW; float* db;
  const int* x_rowmap; const int* g_rowmap; const int* row_off;
  long long strideW; long long strideDb;
  int M, Nn, Kk, ldg, ldx, ldw;
  int tiles_n, tiles_k, nsplit, n_groups;
  long long gcol_stride, xcol_stride;      // COLG build: group g reads the columns G + g * gcol_stride, X + g * xcol_stride (all M rows)
  long long g_chunk_stride; int g_chunk_w;  // COLG build, g_chunk_w > 0: G column c lives at (c / g_chunk_w) * g_chunk_stride + c % g_chunk_w
  const float* gscale; long long gscale_gstride; int gscale_ld;   // SCALE build: G row m of group g is multiplied by gscale[g * gscale_gstride + m * gscale_ld]
  float* partial;   // plain gemm_tn4w build, non-null: every workgroup STORES its 256 x 256 fp32 tile at partial + id * 65536 (accumulator order) instead of adding it to dW atomically; tn_reduce_kernel sums the row ranges
};

template <bool MAPPED>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 32768];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wn = wid >> 1, wk = wid & 1;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = id % p.tiles_k; id /= p.tiles_k;
  const int tile_n = id % p.tiles_n; id /= p.tiles_n;
  const int split = id % p.nsplit;
  const int group = id / p.nsplit;
  int r0 = 0, r1 = p.M;
  if (p.row_off) { r0 = p.row_off[group]; r1 = p.row_off[group + 1]; }
  const int chunk = (((r1 - r0 + p.nsplit - 1) / p.nsplit) + BK - 1) / BK * BK;
  const int ms = r0 + split * chunk;
  const int me = min(r1, ms + chunk);
  if (ms >= me) return;
  const int n0 = tile_n * 128, k0 = tile_k * 128;

  // staging (register-staged, cdna guide T14): each thread loads 16 B of 4 rows of both tiles into
  // registers BEFORE the MFMA phase and writes them to the other LDS buffer AFTER it, so the HBM
  // latency hides under the MFMAs (an LDS-DMA + transposed-read mix makes hipcc drain the DMA with
  // vmcnt(0) ahead of every ds_read_b64_tr_b16).  Tile = 64 rows x 128 cols (256-B rows);
  // chunk' = chunk ^ ((row&3)<<2) keeps the transposed reads conflict-free; OOB rows are zeros.
  const int srow = wid * 4 + (lane >> 4);
  const int schunk = lane & 15;
  const int lds_off = srow * 256 + ((schunk ^ ((srow & 3) << 2)) << 4);
  const bool g_col_ok = (n0 + schunk * 8) < p.Nn;
  const bool x_col_ok = (k0 + schunk * 8) < p.Kk;
  const bf16_t* gbase = p.G + (g_col_ok ? n0 + schunk * 8 : 0);     // always a valid address; masked below
  const bf16_t* xbase = p.X + (x_col_ok ? k0 + schunk * 8 : 0);
  int gidx[4], xidx[4];
  auto load_idx = [&](int mbase) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mbase + i * 16 + srow;
      const bool ok = m < me;
      const int mc = ok ? m : ms;
      if constexpr (MAPPED) {
        const int gi = p.g_rowmap ? p.g_rowmap[mc] : mc;
        const int xi = p.x_rowmap ? p.x_rowmap[mc] : mc;
        gidx[i] = ok ? gi : -1; xidx[i] = ok ? xi : -1;   // NOTE: physical row 0 must exist (it is read, then masked)
      } else {
        gidx[i] = ok ? m : -1; xidx[i] = gidx[i];
      }
    }
  };
  uint4 rg[4], rx[4];
  bool okg[4], okx[4];
  auto gload = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rg[i] = *(const uint4*)(gbase + (long long)max(gidx[i], 0) * p.ldg);
      rx[i] = *(const uint4*)(xbase + (long long)max(xidx[i], 0) * p.ldx);
      okg[i] = gidx[i] >= 0 && g_col_ok; okx[i] = xidx[i] >= 0 && x_col_ok;
    }
  };
  auto lds_write = [&](int buf) {
    char* sG = smem + buf * 32768 + lds_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // masking happens here (after the MFMA phase) so the loads stay in flight across it
      *(uint4*)(sG + i * 4096) = okg[i] ? rg[i] : make_uint4(0, 0, 0, 0);
      *(uint4*)(sG + 16384 + i * 4096) = okx[i] ? rx[i] : make_uint4(0, 0, 0, 0);
    }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[2] = {0.f, 0.f};
  const bool do_db = (p.db != nullptr) && tile_k == 0 && wk == 0;

  // transposed-read addressing (cdna guide T10): in each 16-lane group lane 4q+p supplies
  // row q, columns 4p..4p+3 of a 4x16 block and receives column (lane&15).
  const int h = lane >> 5, gam = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  const int tr_row = 8 * h + q;                        // + 16*ks (+4 for the second read)
  const int tr_colg = (wn * 64 + gam * 16 + 4 * pp);   // + tn*32   (G tile column, elements)
  const int tr_colx = (wk * 64 + gam * 16 + 4 * pp);

  auto tr_addr = [&](const char* base, int row, int col) -> const char* {
    const int c = col >> 3;
    return base + row * 256 + ((c ^ ((row & 3) << 2)) << 4) + ((col & 7) << 1);
  };

  auto compute = [&](int buf) {
    const char* sG = smem + buf * 32768;
    const char* sX = sG + 16384;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8_t gf[2], xf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = ks * 16 + tr_row;
        bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4_t*)tr_addr(sG, row, tr_colg + t * 32));
        bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4_t*)tr_addr(sG, row + 4, tr_colg + t * 32));
        gf[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4_t*)tr_addr(sX, row, tr_colx + t * 32));
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (__attribute__((address_space(3))) bf16x4_t*)tr_addr(sX, row + 4, tr_colx + t * 32));
        xf[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      if (do_db) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 8; ++e) colsum[t] += (float)gf[t][e];
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
          acc[tn][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gf[tn], xf[tk], acc[tn][tk], 0, 0, 0);
    }
  };

  const int nt = (me - ms + BK - 1) / BK;
  load_idx(ms);
  gload();
  load_idx(ms + BK);
  lds_write(0);
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const bool more = t + 1 < nt;
    if (more) {
      gload();                              // stage t+1 -> registers (indices fetched one iteration ago)
      load_idx(ms + (t + 2) * BK);
    }
    compute(cur);
    if (more) lds_write(cur ^ 1);           // buffer cur^1 was last read before the previous barrier
    __syncthreads();
    cur ^= 1;
  }

  float* dW = p.dW + (long long)group * p.strideW;
  const int kcol = lane & 31;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      const int k = k0 + wk * 64 + tk * 32 + kcol;
      if (k >= p.Kk) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < p.Nn) atomicAdd(dW + (long long)n * p.ldw + k, acc[tn][tk][r]);
      }
    }
  if (do_db) {
    float* db = p.db + (long long)group * p.strideDb;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v = colsum[t] + __shfl_xor(colsum[t], 32, 64);
      const int n = n0 + wn * 64 + t * 32 + kcol;
      if (h == 0 && n < p.Nn) atomicAdd(db + n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// scores512: the local-loss score GEMM S = ctx . words^T of ONE caption length class with the word-softmax fused
// (losses.py:713-716), on the gemm_nt512 structure: 256 region rows x (whole captions) per workgroup, sub-steps of
// 32, ring of four LDS sub-stages, ping-pong wave groups.  The older 128-row kernel (loss.hip) runs a two-stage
// lock-step pipeline at ~680 TFLOP/s.  A wave owns whole captions, so the softmax over a caption's words needs only the
// two cross-lane-group shuffles of the 16x16 accumulator layout (lane: region row fr, words 4g..4g+3 of every 16-tile).
//   NTT 1, 2, 4: waves 2 (rows) x 4 (cols), wave tile 128 x 64 = 4 / 2 / 1 captions;  NTT 3: 128 x 48 = 1 caption;
//   NTT 5: waves 4 x 2, wave tile 64 x 80 = 1 caption.
// Output: the word-softmax as fp16 LOG-probabilities S - lse at the caption's columns of the ragged pair matrix + the row
// log-sum-exp (fp32).  Both a1 = exp(.) and S = . + lse are recovered to 2^-11 relative: bf16 probabilities lost S for
// every word whose probability underflowed, and kept only 2^-9 absolute on the rest (profiles/r02_notes.md).
// ---------------------------------------------------------------------------------------------
struct ScoresArgs {
  const bf16_t* ctx; const bf16_t* words; const int* cap_lens; const int* cap_list;
  bf16_t* a1; float* lse;
  int M, HW, HWP, Bc, T, D, n_cap;
  long long col_base, ldp, bstride;      // bstride: TR output only, elements between two images' blocks
  int skip_epi; float inv_hw;
};

// TR: the TRANSPOSED pair matrix of pair3.hip - rows = caption words (row col_base + cj*TP + t), columns = image regions
// (b*HWP + hw), ldp columns per row, values = fp16 LOG2-probabilities (S - lse) / ln 2.  The MFMA operands swap roles, so a lane
// holds four consecutive regions of one word (8-byte stores; HW % 4 == 0 keeps them inside one image) and the softmax over a
// caption's words runs across the 16 lanes of a DPP row.
template <int CTRL>
__device__ __forceinline__ float sc_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sc_row16_sum(float v) {
  v += sc_dpp<0xB1>(v); v += sc_dpp<0x4E>(v); v += sc_dpp<0x141>(v); v += sc_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float sc_row16_max(float v) {
  v = fmaxf(v, sc_dpp<0xB1>(v)); v = fmaxf(v, sc_dpp<0x4E>(v)); v = fmaxf(v, sc_dpp<0x141>(v)); v = fmaxf(v, sc_dpp<0x140>(v));
  return v;
}

// Four independent row reductions interleaved: a DPP operand needs two wait states behind the VALU write of its register, and three
// other instructions sit between two uses of one register here - no s_nop, no separate v_mov_b32_dpp (the compiler emits mov + op + nop
// per step: 12 issue slots per value against 4).
#define SC_DPP4(OP, CTRL)                                                                                            \
  "v_" OP "_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_" OP "_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
  "v_" OP "_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_" OP "_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void sc_row16_max4(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t" SC_DPP4("max", "quad_perm:[1,0,3,2]") SC_DPP4("max", "quad_perm:[2,3,0,1]") SC_DPP4("max", "row_half_mirror")
               SC_DPP4("max", "row_mirror") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void sc_row16_sum4(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t" SC_DPP4("add", "quad_perm:[1,0,3,2]") SC_DPP4("add", "quad_perm:[2,3,0,1]") SC_DPP4("add", "row_half_mirror")
               SC_DPP4("add", "row_mirror") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

template <int NTT, bool TR = false>
__global__ __launch_bounds__(512, 2) void scores512_kernel(ScoresArgs p) {
  constexpr int TP = NTT * 16;
  constexpr int WNN = (NTT == 5) ? 2 : 4;                     // waves along the columns
  constexpr int WMM = 8 / WNN;
  constexpr int TMW = 16 / WMM;                               // 16-row tiles per wave (256 rows per workgroup)
  constexpr int CW = (NTT == 1) ? 4 : (NTT == 2) ? 2 : 1;     // captions per wave
  constexpr int TNW = CW * NTT;                               // 16-col tiles per wave
  constexpr int BNS = WNN * TNW * 16;                          // word rows (= score columns) per workgroup
  constexpr int CAPB = WNN * CW;                              // captions per workgroup
  constexpr int NPIECE = (256 + BNS) / 16;                     // 1-KB DMA pieces per sub-stage
  __shared__ __attribute__((aligned(16))) char smem[NRING * SUB3];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid % WMM, wn = wid / WMM;                   // the two waves of a SIMD (w, w+4) take different captions
  const int fr = lane & 15, g = lane >> 4;
  const int G = gridDim.x;
  const int my = xcd_remap(blockIdx.x, G);
  const int tiles_m = (p.M + 255) / 256, tiles_n = (p.n_cap + CAPB - 1) / CAPB;
  const int total = tiles_m * tiles_n;
  const int nu = p.D / 32;
  struct Tile { int m0, c0; };
  // column tiles of one row tile are consecutive: the 32 workgroups of an XCD share a few ctx panels and the words
  auto decode = [&](int id) -> Tile { Tile t; t.m0 = (id / tiles_n) * 256; t.c0 = (id % tiles_n) * CAPB; return t; };

  unsigned src[4];
  bool have[4];
  auto setup = [&](const Tile& t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = i * 8 + wid;
      have[i] = piece < NPIECE;
      const int line = piece * 8 + (lane >> 3);
      const int lc = (lane & 7) ^ (line & 7);
      const int row = 2 * line + (lc >> 2);                   // 0..255 ctx rows, 256.. word rows
      if (i < 2) src[i] = (unsigned)min(t.m0 + row, p.M - 1) * (unsigned)(p.D * 2) + (lc & 3) * 16;
      else {
        const int n = min(row - 256, BNS - 1);
        const int cj = min(t.c0 + n / TP, p.n_cap - 1), tw = min(n % TP, p.T - 1);
        src[i] = (unsigned)(p.cap_list[cj] * p.T + tw) * (unsigned)(p.D * 2) + (lc & 3) * 16;
      }
    }
  };
  auto stage = [&](int buf, int k0) {
    char* sb = smem + buf * SUB3 + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (have[i])
        __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)(i < 2 ? p.ctx : p.words) + k0 * 2 + src[i]), LDS_PTR(sb + i * 8192), 16, 0, 0);
  };
  f32x4_t acc[TMW][TNW];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TNW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  const int lane_off = (fr >> 1) * 128 + (((((fr & 1) << 2) | g) ^ ((fr >> 1) & 7)) << 4);
  const int offA_nat = (wm * TMW * 16) * 64 + lane_off;
  // word rows are read sigma-permuted (as the B rows of gemm_nt512): lane (fr, g) then owns words pg*4 .. pg*4+3 of every
  // 16-word tile, and one permlane32_swap per tile pair gives each lane 8 consecutive words (16-byte stores)
  const int sig = ((((fr >> 2) & 1) << 1 | (fr >> 3)) << 2) | (fr & 3);
  const int pg = ((g & 1) << 1) | (g >> 1);
  const int sig_off = (sig >> 1) * 128 + (((((sig & 1) << 2) | g) ^ ((sig >> 1) & 7)) << 4);
  // TR: the sigma permutation moves to the ctx rows (they become the accumulator rows), the word rows are read in natural order
  const int offA = TR ? (wm * TMW * 16) * 64 + sig_off : offA_nat;
  const int offB = (256 + wn * TNW * 16) * 64 + (TR ? lane_off : sig_off);
  bf16x8_t af[TMW], bf[TNW];
  auto read_frags = [&](int buf) __attribute__((always_inline)) {
    const char* sA = smem + buf * SUB3 + offA;
    const char* sB = smem + buf * SUB3 + offB;
#pragma unroll
    for (int t = 0; t < TMW; ++t) af[t] = *(const bf16x8_t*)(sA + t * 1024);
#pragma unroll
    for (int t = 0; t < TNW; ++t) bf[t] = *(const bf16x8_t*)(sB + t * 1024);
  };
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn)
        acc[tm][tn] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tm], bf[tn], acc[tm][tn], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tm][tn], 0, 0, 0);
  };
  // TR epilogue: lane (fr, g) holds word tn*16 + fr of the caption and regions pg*4 .. pg*4+3 of region tile tm (the ctx rows are read
  // sigma-permuted); one permlane32_swap per pair of region tiles then gives each lane 8 consecutive regions (16-byte stores).
  // The epilogue is instruction-bound (it was 41 % of the kernel: 3000 vector + 1200 scalar instructions per wave and tile), so:
  // the padding words of a caption are set to -inf ONCE (class NTT holds captions of 16 (NTT-1) < len <= 16 NTT words: only the last
  // word tile can have padding) and the max / exp / clamp then need no selects; row -> (image, region) by a reciprocal multiply with one
  // correction step instead of an integer division; store predicates hoisted out of the word-tile loop.
  auto div_hw = [&](int m, int& q, int& r) __attribute__((always_inline)) {
    q = (int)(((float)m + 0.5f) * p.inv_hw);
    r = m - q * p.HW;
    if (r < 0) { --q; r += p.HW; }
    if (r >= p.HW) { ++q; r -= p.HW; }
  };
  auto epilogue_t = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int cj = t.c0 + wn * CW + c;
      const bool cap_ok = cj < p.n_cap && p.skip_epi != 2;          // skip_epi 2 (measurement): the whole epilogue except its stores
      const int cap_i = p.cap_list[min(cj, p.n_cap - 1)];
      const int cap = max(1, min(min(p.cap_lens[cap_i], p.T), TP));
      if ((NTT - 1) * 16 + fr >= cap) {
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[tm][c * NTT + NTT - 1][r] = -INFINITY;
      }
      if (NTT > 1 && cap <= (NTT - 1) * 16) {                      // a caption shorter than its class (not what the engine's tables build): every tile
#pragma unroll
        for (int tn = 0; tn < NTT - 1; ++tn)
          if (tn * 16 + fr >= cap) {
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[tm][c * NTT + tn][r] = -INFINITY;
          }
      }
      // Two region-tile pairs (jp = 2 J, 2 J + 1: 64 consecutive regions) per round.  After the permlane32_swap a lane holds 8 consecutive
      // regions of word fr, and one store instruction would write 16 word rows x 64 bytes - the shape a CU's store path drains at 12.5 B/clk
      // (profiles/r02_notes.md, tools/store_bw.hip): the stores, not the softmax arithmetic, set the epilogue's length.  One more exchange, a
      // bank-masked row_ror:8 DPP move per dword, hands the lanes fr >= 8 the NEXT 32 regions of word fr - 8 (and the lanes fr < 8 the first 32
      // regions of word fr + 8): every store instruction then writes 8 word rows x 128 contiguous bytes (39 B/clk).
#pragma unroll
      for (int J = 0; J < TMW / 4; ++J) {
        uint4 vv[2][NTT];
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
          const int jp = 2 * J + jq;
          uint2 o[2][NTT];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int tm = 2 * jp + q;
            // log2 domain: t = S log2 e, lse2 = max + log2 sum 2^(t - max), output t - lse2, stored lse = lse2 ln 2.  The scale rides in the
            // two fused multiply-adds (the maximum is taken on the raw scores: the scale is positive), and the clamp to LOGP_MIN runs on the
            // packed halves (-60000 is an fp16 value; a log-probability below the fp16 range converts to -inf and is clamped the same):
            // 192 of the epilogue's 1850 vector instructions per wave and tile gone.
            constexpr float L2E = 1.44269504088896f;
            float mx[4], sm[4], lse2[4];
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              mx[r] = acc[tm][c * NTT][r];
  #pragma unroll
              for (int tn = 1; tn < NTT; ++tn) mx[r] = fmaxf(mx[r], acc[tm][c * NTT + tn][r]);
            }
            sc_row16_max4(mx[0], mx[1], mx[2], mx[3]);             // word 0 of every caption is real: finite
  #pragma unroll
            for (int r = 0; r < 4; ++r) {
              mx[r] *= L2E;
              sm[r] = 0.f;
  #pragma unroll
              for (int tn = 0; tn < NTT; ++tn) sm[r] += __builtin_amdgcn_exp2f(__builtin_fmaf(acc[tm][c * NTT + tn][r], L2E, -mx[r]));
            }
            sc_row16_sum4(sm[0], sm[1], sm[2], sm[3]);
            float lse4[4];
  #pragma unroll
            for (int r = 0; r < 4; ++r) { lse2[r] = mx[r] + __builtin_amdgcn_logf(sm[r]); lse4[r] = lse2[r] * 0.6931471805599453f; }
            typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
            const h2_t lo2 = __builtin_bit_cast(h2_t, LOGP_MIN_BITS2);
  #pragma unroll
            for (int tn = 0; tn < NTT; ++tn) {
              float e[4];
  #pragma unroll
              for (int r = 0; r < 4; ++r) e[r] = __builtin_fmaf(acc[tm][c * NTT + tn][r], L2E, -lse2[r]);
              o[q][tn].x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h2_t, pack2h(e[0], e[1])), lo2));
              o[q][tn].y = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h2_t, pack2h(e[2], e[3])), lo2));
            }
            const int m4 = t.m0 + wm * TMW * 16 + tm * 16 + 4 * pg;          // this lane's four regions of tile tm (one image: HW % 4 == 0)
            if (fr == 0 && m4 < p.M && cap_ok) {
              int mb, hw;
              div_hw(m4, mb, hw);
              *(float4*)(p.lse + ((long long)mb * p.Bc + cap_i) * p.HWP + hw) = make_float4(lse4[0], lse4[1], lse4[2], lse4[3]);
            }
          }
  
#pragma unroll
          for (int tn = 0; tn < NTT; ++tn) {
            auto r0 = __builtin_amdgcn_permlane32_swap(o[0][tn].x, o[1][tn].x, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(o[0][tn].y, o[1][tn].y, false, false);
            vv[jq][tn] = make_uint4(r0[0], r1[0], r0[1], r1[1]);
          }
        }
        // lane (fr, g): word row (fr & 7) + 8 i of instruction i, regions m8 .. m8 + 7
        const int m8 = t.m0 + wm * TMW * 16 + (4 * J + (g >> 1)) * 16 + (g & 1) * 8 + 32 * (fr >> 3);
        int mb, hw;
        div_hw(min(m8, p.M - 4), mb, hw);
        const bool ok = m8 < p.M && cap_ok;
        const bool whole = hw + 8 <= p.HW;                                  // else the image ends after four of them
        const bool ok2 = m8 + 4 < p.M;
        bf16_t* dst0 = p.a1 + (p.col_base + (long long)cj * TP + (fr & 7)) * p.ldp + (long long)mb * p.bstride + hw;
        bf16_t* dst2 = p.a1 + (p.col_base + (long long)cj * TP + (fr & 7)) * p.ldp + (long long)(mb + 1) * p.bstride;
#pragma unroll
        for (int tn = 0; tn < NTT; ++tn) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const uint4 a = vv[0][tn], b = vv[1][tn];
            uint4 y;
            if (i == 0) {           // lanes 8..15 of every row take the second pair's piece of lane - 8
              y.x = __builtin_amdgcn_update_dpp(a.x, b.x, 0x128, 0xf, 0xC, false); y.y = __builtin_amdgcn_update_dpp(a.y, b.y, 0x128, 0xf, 0xC, false);
              y.z = __builtin_amdgcn_update_dpp(a.z, b.z, 0x128, 0xf, 0xC, false); y.w = __builtin_amdgcn_update_dpp(a.w, b.w, 0x128, 0xf, 0xC, false);
            } else {                // lanes 0..7 take the first pair's piece of lane + 8
              y.x = __builtin_amdgcn_update_dpp(b.x, a.x, 0x128, 0xf, 0x3, false); y.y = __builtin_amdgcn_update_dpp(b.y, a.y, 0x128, 0xf, 0x3, false);
              y.z = __builtin_amdgcn_update_dpp(b.z, a.z, 0x128, 0xf, 0x3, false); y.w = __builtin_amdgcn_update_dpp(b.w, a.w, 0x128, 0xf, 0x3, false);
            }
            const long long ro = (long long)(tn * 16 + 8 * i) * p.ldp;
            if (ok && whole) *(uint4*)(dst0 + ro) = y;
            else if (ok) {
              *(uint2*)(dst0 + ro) = make_uint2(y.x, y.y);
              if (ok2) *(uint2*)(dst2 + ro) = make_uint2(y.z, y.w);
            }
          }
        }
      }
    }
  };
  // word softmax per region row and caption: the row's words are (tn, r) in this lane and the 4 lane groups g
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) -> int {
    int n_st = 0;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      const int cj = t.c0 + wn * CW + c;
      const bool cap_ok = cj < p.n_cap;
      const int cap_i = p.cap_list[min(cj, p.n_cap - 1)];
      const int cap = min(min(p.cap_lens[cap_i], p.T), TP);
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) {
        float mx = -INFINITY;
#pragma unroll
        for (int tn = 0; tn < NTT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (tn * 16 + pg * 4 + r < cap) mx = fmaxf(mx, acc[tm][c * NTT + tn][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sm = 0.f;
#pragma unroll
        for (int tn = 0; tn < NTT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sm += (tn * 16 + pg * 4 + r < cap) ? __expf(acc[tm][c * NTT + tn][r] - mx) : 0.f;
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        const float lse = mx + __logf(sm);
        const int m = t.m0 + wm * TMW * 16 + tm * 16 + fr;
        const bool ok = m < p.M && cap_ok;
        const int mb = min(m, p.M - 1) / p.HW, hw = min(m, p.M - 1) - mb * p.HW;
        if (ok && g == 0) p.lse[((long long)mb * p.Bc + cap_i) * p.HWP + hw] = lse;       // [image][caption][region]
        bf16_t* dst = p.a1 + ((long long)mb * p.HWP + hw) * p.ldp + p.col_base + (long long)cj * TP;
        // the tile holds the word-softmax as fp16 LOG-probabilities S - lse (masked words: LOGP_MIN)
        float e[NTT][4];
#pragma unroll
        for (int tn = 0; tn < NTT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            e[tn][r] = (tn * 16 + pg * 4 + r < cap) ? fmaxf(acc[tm][c * NTT + tn][r] - lse, LOGP_MIN) : LOGP_MIN;
        uint2 o[NTT];
#pragma unroll
        for (int tn = 0; tn < NTT; ++tn) {
          o[tn].x = pack2h(e[tn][0], e[tn][1]);
          o[tn].y = pack2h(e[tn][2], e[tn][3]);
        }
#pragma unroll
        for (int j = 0; j < NTT / 2; ++j) {          // tile pairs: 16-byte stores of 8 consecutive words
          auto r0 = __builtin_amdgcn_permlane32_swap(o[2 * j].x, o[2 * j + 1].x, false, false);
          auto r1 = __builtin_amdgcn_permlane32_swap(o[2 * j].y, o[2 * j + 1].y, false, false);
          if (ok) *(uint4*)(dst + (2 * j + (g >> 1)) * 16 + (g & 1) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
        if ((NTT & 1) && ok) *(uint2*)(dst + (NTT - 1) * 16 + pg * 4) = o[NTT - 1];
        n_st += NTT;
      }
    }
    return n_st + TMW * CW;         // upper bound of the store instructions issued (A1 pieces + lse): see wait below
  };

  if (my >= total) return;
  const int my_tiles = (total - my + G - 1) / G;
  // Where the time goes (tools/scores_probe.py, batch 1024, all five classes): k-loops 15.0 ms, + epilogue arithmetic 3.6 ms, + its stores
  // 4.8 ms (128 KB per tile leave a CU at ~7 B/clk).  Measured and without effect on that: 10 % fewer vector instructions in the epilogue, store
  // instructions of 8 rows x 128 bytes instead of 16 x 64, workgroups started in 2..16 phases a fraction of a tile apart (the store cost is
  // per CU, not a chip-wide burst).
  Tile ct = decode(my);
  setup(ct);
  int lid = my, lk = 0, wb = 0, rb = 0;
  bool lmore = true;
  // The epilogue's stores sit between DMA stages in the vmcnt order; their number depends on predicates, so after a
  // seam the counted wait is replaced by a full drain for the three waits that could still see them (once per tile).
  int drain = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    if (lmore) {
      stage(wb, lk * 32);
      wb = (wb + 1) & 3;
      if (++lk == nu) {
        lk = 0; lid += G;
        if (lid < total) { const Tile lt = decode(lid); setup(lt); } else lmore = false;
      }
    }
  };
  // pieces per stage differ per wave when NPIECE < 32: the wave's own count
  const int my_pieces = (have[0] ? 1 : 0) + (have[1] ? 1 : 0) + (have[2] ? 1 : 0) + (have[3] ? 1 : 0);
  auto wait_third_newest = [&]() __attribute__((always_inline)) {
    if (drain > 0 || !lmore) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (drain > 0) --drain; }
    else if (my_pieces == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (my_pieces == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  const int grp = __builtin_amdgcn_readfirstlane(wid >> 2);
  zero_acc();
  issue(); issue(); issue();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  auto seg_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  auto load_seg = [&]() __attribute__((always_inline)) {
    read_frags(rb);
    rb = (rb + 1) & 3;
    issue();
    if (grp == 1) wait_third_newest();
    seg_barrier();
  };
  auto body = [&](bool last) __attribute__((always_inline)) {
    compute();
    if (grp == 0) { wait_third_newest(); seg_barrier(); }
    if (last) {
      if (p.skip_epi != 1) { if constexpr (TR) epilogue_t(ct); else epilogue(ct); }
      zero_acc();
      drain = 3;
    }
    if (grp == 1) seg_barrier();
    load_seg();
  };
  if (grp == 1) seg_barrier();
  load_seg();
  for (int ti = 0; ti < my_tiles; ++ti) {
    for (int k = 1; k < nu; ++k) body(false);
    body(true);
    if (ti + 1 < my_tiles) ct = decode(my + (ti + 1) * G);
  }
  if (grp == 0) seg_barrier();
}

// TRANSPOSED output (see the TR note above scores512_kernel): lpT[(row_base + j*16*ntt + t) * ld + b*bstride + hw] = log2-probability of
// word t of the class's j-th caption at region hw of image b (ld = B*HWP, bstride = HWP: [word rows][image region columns]; ld = HWP,
// bstride = rows*HWP: image-major); lse as in the untransposed kernel.  Any M (rows are clamped in the loads).
extern "C" int medmoe_local_scores_t(const void* ctx, const void* words, const int* cap_lens, void* lpT, float* lse, int B, int Bc,
                                     int HW, int T, int D, const int* cap_list, int n_cap, int ntt, long long row_base, long long ld,
                                     long long bstride, hipStream_t stream) {
  if (!ctx || !words || !cap_lens || !lpT || !lse || !cap_list) return MM_ERR_ARG;
  if (B <= 0 || Bc <= 0 || HW < 4 || (HW % 4) || T <= 0 || n_cap <= 0 || n_cap > Bc || ntt < 1 || ntt > 5 || row_base < 0) return MM_ERR_SHAPE;
  const long long M = (long long)B * HW;
  const int HWP = ((HW + 15) / 16) * 16;
  if ((D % 32) || D < 128 || M * D * 2 >= (1ll << 32) || (long long)Bc * T * D * 2 >= (1ll << 32) || (ld % 4) || (bstride % 4) || ld < HWP || bstride < HWP) return MM_ERR_SHAPE;
  ScoresArgs p;
  p.ctx = (const bf16_t*)ctx; p.words = (const bf16_t*)words; p.cap_lens = cap_lens; p.cap_list = cap_list;
  p.bstride = bstride; p.skip_epi = g_scores_skip_epi; p.inv_hw = 1.0f / (float)HW;
  p.a1 = (bf16_t*)lpT; p.lse = lse;
  p.M = (int)M; p.HW = HW; p.HWP = HWP; p.Bc = Bc; p.T = T; p.D = D; p.n_cap = n_cap;
  p.col_base = row_base; p.ldp = ld;
  const int tiles_m = (int)((M + 255) / 256);
#define SC(N_, CAPB_) { const int grid = min(tiles_m * ((n_cap + CAPB_ - 1) / CAPB_), 256); \
                        hipLaunchKernelGGL((scores512_kernel<N_, true>), dim3(grid), dim3(512), 0, stream, p); }
  switch (ntt) { case 1: SC(1, 16) break; case 2: SC(2, 8) break; case 3: SC(3, 4) break; case 4: SC(4, 4) break; default: SC(5, 2) break; }
#undef SC
  return mm_check_launch();
}

// launcher used by medmoe_local_scores_ragged (loss.hip); returns false when the shape is not taken
bool mm_launch_scores512(const void* ctx, const void* words, const int* cap_lens, void* a1, float* lse, int B, int Bc, int HW, int T,
                         int D, const int* cap_list, int n_cap, int ntt, long long col_base, long long ldp, hipStream_t stream) {
  const long long M = (long long)B * HW;
  if (!g_use_scores512 || (D % 32) || D < 128 || M < 1024 || M * D * 2 >= (1ll << 32) || (long long)Bc * T * D * 2 >= (1ll << 32)) return false;
  ScoresArgs p;
  p.ctx = (const bf16_t*)ctx; p.words = (const bf16_t*)words; p.cap_lens = cap_lens; p.cap_list = cap_list;
  p.a1 = (bf16_t*)a1; p.lse = lse;
  p.M = (int)M; p.HW = HW; p.HWP = ((HW + 15) / 16) * 16; p.Bc = Bc; p.T = T; p.D = D; p.n_cap = n_cap;
  p.col_base = col_base; p.ldp = ldp; p.bstride = 0; p.skip_epi = g_scores_skip_epi; p.inv_hw = 1.0f / (float)HW;
  const int tiles_m = (int)((M + 255) / 256);
#define SC(N_, CAPB_) { const int grid = min(tiles_m * ((n_cap + CAPB_ - 1) / CAPB_), 256); \
                        hipLaunchKernelGGL((scores512_kernel<N_>), dim3(grid), dim3(512), 0, stream, p); }
  switch (ntt) { case 1: SC(1, 16) break; case 2: SC(2, 8) break; case 3: SC(3, 4) break; case 4: SC(4, 4) break; default: SC(5, 2) break; }
#undef SC
  return true;
}

// Grouped / row-mapped GEMM on the 256x256 kernel: same arguments as medmoe_gemm_nt, but `tiles` holds 256-ROW tiles
// (medmoe_dispatch writes that table behind the 128-row one).  Shapes the kernel does not take return MM_ERR_SHAPE.
extern "C" int medmoe_gemm_nt_tiles256(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                       int M, int N, int K, const float* bias, const void* residual, int ldr,
                                       void* aux, int ldaux, const int* a_rowmap, const int* c_rowmap,
                                       const int* tiles, const int* tile_count, int max_tiles,
                                       long long strideB, long long strideBias, float alpha, int epi,
                                       int out_f32, int col_perm, hipStream_t stream) {
  if (!A || !B || !C || !tiles || !tile_count || max_tiles <= 0) return MM_ERR_ARG;
  if (M <= 0 || N <= 0 || K < 128 || (K % BK) != 0 || (N % 8) != 0 || col_perm || out_f32 || alpha != 1.f) return MM_ERR_SHAPE;
  if ((lda % 8) || (ldb % 8) || (ldc % 8) || (residual && (ldr % 8)) || (aux && (ldaux % 8))) return MM_ERR_SHAPE;
  if (epi < 0 || epi > EPI_MUL_AUX) return MM_ERR_ARG;
  if ((epi == EPI_MUL_DGELU || epi == EPI_MUL_DRELU || epi == EPI_MUL_AUX) && !aux) return MM_ERR_ARG;
  if (256ll * lda * 2 >= (1ll << 32) || (long long)N * ldb * 2 >= (1ll << 32)) return MM_ERR_SHAPE;
  if (a_rowmap && (long long)M * lda * 2 >= (1ll << 32)) return MM_ERR_SHAPE;      // gathered rows: offsets from A itself
  GemmNTArgs p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
  p.bias = bias; p.residual = (const bf16_t*)residual; p.aux = (bf16_t*)aux;
  p.a_rowmap = a_rowmap; p.c_rowmap = c_rowmap; p.tiles = (const int4*)tiles; p.tile_count = tile_count;
  p.strideB = strideB; p.strideBias = strideBias;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldaux = ldaux;
  p.n_tiles_n = (N + 255) / 256; p.max_tiles_m = max_tiles;
  p.alpha = 1.f; p.epi = epi; p.out_f32 = 0; p.col_perm = 0; p.m_dev = nullptr;
  const int grid = min(max_tiles * p.n_tiles_n, 256);
  if (g_use_nt4w && 2ll * lda < (1ll << 24) && 2ll * ldb < (1ll << 24)) {
    bool done = true;
    switch (NT_SPEC(epi, bias ? 1 : 0, residual ? 1 : 0, aux ? 1 : 0)) {
#define NT_CASE(s) case s: hipLaunchKernelGGL((gemm_nt4w_kernel<s, true>), dim3(grid), dim3(256), 0, stream, p); break;
      NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0))
      NT_CASE(NT_SPEC(EPI_RELU, 1, 0, 0))
      NT_CASE(NT_SPEC(EPI_MUL_DRELU, 0, 1, 1))
      NT_CASE(NT_SPEC(EPI_NONE, 0, 1, 0))          // pyramid experts: dG += dH1 . W0 (the ReLU' follows the interpolation, swin.py:41-42)
#undef NT_CASE
      default: done = false; break;
    }
    if (done) { g_last_nt_kernel = 5; return mm_check_launch(); }
  }
  g_last_nt_kernel = 3;
  switch (NT_SPEC(epi, bias ? 1 : 0, residual ? 1 : 0, aux ? 1 : 0)) {
#define NT_CASE(s) case s: hipLaunchKernelGGL((gemm_nt512_kernel<s, true>), dim3(grid), dim3(512), 0, stream, p); break;
    NT_CASE(NT_SPEC(EPI_NONE, 0, 0, 0))            // expert dgrad, dGm
    NT_CASE(NT_SPEC(EPI_RELU, 1, 0, 0))            // expert projections / scale-attention hidden layer
    NT_CASE(NT_SPEC(EPI_MUL_DRELU, 0, 1, 1))       // gradient through the projection ReLU, accumulated
#undef NT_CASE
    default: hipLaunchKernelGGL((gemm_nt512_kernel<-1, true>), dim3(grid), dim3(512), 0, stream, p); break;
  }
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// gemm_tn512: wgrad dW[Nn,Kk] += G[m0:m1, :]^T X[m0:m1, :] with a 256x256 output tile per workgroup, 8 waves
// (2x4 of 128x64), the gemm_nt512 structure: sub-steps of 32 token rows, ring of four 32 KB LDS sub-stages filled
// by LDS-DMA, ping-pong between the two waves of a SIMD.  Half the DMA bytes per flop of the 128x128 kernel.
//   * both operands are reduction-major in memory, so fragments come from ds_read_b64_tr_b16.  hipcc drains
//     every LDS-DMA with vmcnt(0) in front of the tr-read BUILTIN, so the reads are inline asm: the fragment
//     registers pass through the s_waitcnt asm as "+v" operands, which is what orders the MFMAs behind it.
//   * LDS image of a sub-stage: [G: 32 rows x 512 B][X: 32 rows x 512 B], 16-B chunk c of row r stored at
//     chunk c ^ ((r & 3) << 2) (the four rows one transposed read touches land in four different 64-B bank groups).
//   * the M range is split over gridDim so that ~256 workgroups exist; partial sums meet in fp32 atomics.
//   * db = column sums of G: v_dot2_f32_bf16 against ones on the fragments, the four waves that hold the same
//     G columns take turns (sub-step & 3 == wn), only in the k-tile-0 workgroups.
// Requires M % 32 == 0, Nn % 256 == 0, Kk % 256 == 0, no row maps / groups (the 128x128 kernel takes the rest).
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
#define TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))

template <bool MAPPED>
__global__ __launch_bounds__(512, 2) void gemm_tn512_kernel(GemmTNArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * SUB3 + (MAPPED ? 32768 : 0)];     // ring (+ the row-map slice)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;          // wave tile: G columns wm*128.., X columns wn*64..
  // split-major ids: the tiles of one M range (same G / X rows) sit on one XCD
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_n * p.tiles_k;
  // groups vary fastest: unequal groups (an unbalanced router leaves some empty) still spread over every XCD
  const int group = id % p.n_groups; id /= p.n_groups;
  const int split = id / ntile; id -= split * ntile;
  const int tile_n = id / p.tiles_k, tile_k = id - tile_n * p.tiles_k;
  int r0 = 0, r1 = p.M;
  if (MAPPED && p.row_off) { r0 = p.row_off[group]; r1 = p.row_off[group + 1]; }
  const int chunk = (((r1 - r0 + 31) / 32 + p.nsplit - 1) / p.nsplit) * 32;
  const int ms = r0 + split * chunk, me = min(r1, ms + chunk);
  if (ms >= me) return;
  const int U = (me - ms + 31) / 32;              // sub-steps; the last one may hold fewer than 32 valid rows
  const int n0 = tile_n * 256, k0 = tile_k * 256;

  // DMA piece i of wave w fills LDS bytes [(i*8 + w) * 1024, +1024) of the sub-stage: two 512-B rows.
  // MAPPED: rows come through g_rowmap / x_rowmap, rows past the range are clamped
  // (their G rows are zeroed in LDS before the fragment reads), G columns past Nn are clamped (never written back).
  const int prow = (wid << 1) + (lane >> 5);                               // row of pieces 0 / 2; pieces 1 / 3: +16
  unsigned col_g, col_x;
  {
    const int lc0 = (lane & 31) ^ ((prow & 3) << 2);                       // (prow + 16) & 3 == prow & 3
    const int gcol = n0 + lc0 * 8;
    col_g = (unsigned)((MAPPED && gcol >= p.Nn) ? n0 : gcol) * 2u;
    col_x = (unsigned)(k0 + lc0 * 8) * 2u;
  }
  const unsigned ldg2 = (unsigned)p.ldg * 2u, ldx2 = (unsigned)p.ldx * 2u;
  // A row map (g_rowmap or x_rowmap, at most one) is staged through LDS: a ring of 8192 entries (256 sub-steps),
  // filled up front and refilled half a ring at a time every 128 sub-steps.  Loading it from global memory inside
  // the loop makes hipcc drain every LDS-DMA in flight (vmcnt(0)) in front of the first use, which collapses the
  // DMA ring to one stage (measured 113 TF/s); LDS reads are counted by lgkmcnt instead, and the refill's drain
  // happens once per 128 sub-steps.
  const int* rmap = MAPPED ? (p.g_rowmap ? p.g_rowmap : p.x_rowmap) : nullptr;
  int* lmap = (int*)(smem + 4 * SUB3);
  if (MAPPED && rmap) {
    for (int i = tid; i < min(me - ms, 8192); i += 512) lmap[i] = rmap[ms + i];
    __syncthreads();
  }
  int wb = 0, rb = 0, lk = 0;
  int d2 = 0, d1 = 0, d0 = 0;                     // pieces of the three newest sub-stages (no stores in the loop)
  auto issue = [&]() __attribute__((always_inline)) {
    d2 = d1; d1 = d0; d0 = 0;
    if (lk < U) {
      char* sb = smem + wb * SUB3 + wid * 1024;
      unsigned so[4];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ml = min(lk * 32 + prow + j * 16, me - ms - 1);       // row inside the range (clamped past the end)
        const int r = (MAPPED && rmap) ? lmap[ml & 8191] : 0;
        so[j] = (unsigned)((MAPPED && p.g_rowmap) ? r : ms + ml) * ldg2 + col_g;
        so[2 + j] = (unsigned)((MAPPED && p.x_rowmap) ? r : ms + ml) * ldx2 + col_x;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)(i < 2 ? p.G : p.X) + so[i]), LDS_PTR(sb + i * 8192), 16, 0, 0);
      d0 = 4; ++lk;
      wb = (wb + 1) & 3;
    }
  };
  auto wait_third_newest = [&]() { wait_vmcnt_ring4(__builtin_amdgcn_readfirstlane(d1 + d0)); };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_db = (p.db != nullptr) && tile_k == 0;

  // transposed-read addressing (cdna guide T10): in each 16-lane group lane 4q+p supplies row q, columns 4p..4p+3
  // of a 4x16 block and receives column (lane & 15)
  const int h = lane >> 5, gam = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  unsigned ag[4], ax[2];                          // LDS byte address of (row 8h+q, this lane's columns) per 32-column tile
#pragma unroll
  for (int tn = 0; tn < 4; ++tn) {
    const int c = wm * 16 + tn * 4 + gam * 2 + (pp >> 1);
    ag[tn] = (unsigned)(size_t)LDS_PTR(smem) + (8 * h + q) * 512 + ((c ^ (q << 2)) << 4) + ((pp & 1) << 3);
  }
#pragma unroll
  for (int tk = 0; tk < 2; ++tk) {
    const int c = wn * 8 + tk * 4 + gam * 2 + (pp >> 1);
    ax[tk] = (unsigned)(size_t)LDS_PTR(smem) + 16384 + (8 * h + q) * 512 + ((c ^ (q << 2)) << 4) + ((pp & 1) << 3);
  }
  u32x2_t gl[2][4], gh[2][4], xl[2][2], xh[2][2];  // [16-row k sub-step][tile]: rows 8h+q and 8h+q+4
  auto read_frags = [&](int buf) __attribute__((always_inline)) {
    const unsigned bo = buf * SUB3;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const unsigned a = ag[tn] + bo;
      TR_READ(gl[0][tn], a, 0); TR_READ(gh[0][tn], a, 2048); TR_READ(gl[1][tn], a, 8192); TR_READ(gh[1][tn], a, 10240);
    }
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      const unsigned a = ax[tk] + bo;
      TR_READ(xl[0][tk], a, 0); TR_READ(xh[0][tk], a, 2048); TR_READ(xl[1][tk], a, 8192); TR_READ(xh[1][tk], a, 10240);
    }
  };
  auto fence_frags = [&]() __attribute__((always_inline)) {      // data of every tr-read above has arrived
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(gl[0][0]), "+v"(gl[0][1]), "+v"(gl[0][2]), "+v"(gl[0][3]), "+v"(gh[0][0]), "+v"(gh[0][1]), "+v"(gh[0][2]), "+v"(gh[0][3]),
                   "+v"(gl[1][0]), "+v"(gl[1][1]), "+v"(gl[1][2]), "+v"(gl[1][3]), "+v"(gh[1][0]), "+v"(gh[1][1]), "+v"(gh[1][2]), "+v"(gh[1][3]),
                   "+v"(xl[0][0]), "+v"(xl[0][1]), "+v"(xh[0][0]), "+v"(xh[0][1]), "+v"(xl[1][0]), "+v"(xl[1][1]), "+v"(xh[1][0]), "+v"(xh[1][1])
                 :: "memory");
  };
  auto frag = [&](u32x2_t lo, u32x2_t hi) -> bf16x8_t {
    const uint4 v = make_uint4(lo[0], lo[1], hi[0], hi[1]);
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
          acc[tn][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(gl[ks][tn], gh[ks][tn]), frag(xl[ks][tk], xh[ks][tk]), acc[tn][tk], 0, 0, 0);
  };
  auto add_colsum = [&]() __attribute__((always_inline)) {
    const s16x2_t ones = {(short)0x3F80, (short)0x3F80};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const uint4 v = __builtin_bit_cast(uint4, frag(gl[ks][tn], gh[ks][tn]));
        float c = colsum[tn];
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.x), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.y), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.z), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.w), ones, c, false);
        colsum[tn] = c;
      }
  };

  const int grp = __builtin_amdgcn_readfirstlane(wid >> 2);
  auto seg_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  issue(); issue(); issue();
  wait_third_newest();
  __builtin_amdgcn_s_barrier();                   // sub-stage 0 landed for every wave
  asm volatile("" ::: "memory");
  int u_load = 0;
  const int tail = (me - ms) - (U - 1) * 32;       // valid rows of the last sub-step
  auto load_seg = [&]() __attribute__((always_inline)) {
    if (MAPPED && rmap && u_load >= 128 && (u_load & 127) == 0) {
      // sub-steps [u-128, u) are behind every cursor: their ring slots take the rows of sub-steps [u+128, u+256)
      for (int i = tid; i < 4096; i += 512) {
        const int r = (u_load + 128) * 32 + i;
        if (r < me - ms) lmap[r & 8191] = rmap[ms + r];
      }
    }
    if (MAPPED && u_load == U - 1 && tail < 32) {
      // rows past the range hold clamped copies of real rows: zero their G half (every wave zeroes all of them
      // itself, so its own reads below are ordered behind its own writes; zero G rows add nothing to dW or db)
      char* zb = smem + rb * SUB3;
      for (int z = tail * 32 + lane; z < 32 * 32; z += 64) *(uint4*)(zb + z * 16) = make_uint4(0, 0, 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    read_frags(rb);
    rb = (rb + 1) & 3;
    issue();
    fence_frags();
    if (do_db && (u_load & 3) == wn && u_load < U) add_colsum();
    ++u_load;
    if (grp == 1) wait_third_newest();
    seg_barrier();
  };
  if (grp == 1) seg_barrier();
  load_seg();
  for (int u = 0; u < U; ++u) {
    compute();
    if (grp == 0) { wait_third_newest(); seg_barrier(); }
    if (grp == 1) seg_barrier();
    load_seg();                                   // past the end: reads a stale buffer, issues nothing
  }
  if (grp == 0) seg_barrier();

  float* dW = p.dW + (MAPPED ? (long long)group * p.strideW : 0ll);
  const int kcol = lane & 31;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) {
      const int k = k0 + wn * 64 + tk * 32 + kcol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 128 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!MAPPED || n < p.Nn) atomicAdd(dW + (long long)n * p.ldw + k, acc[tn][tk][r]);
      }
    }
  if (do_db) {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const float v = colsum[tn] + __shfl_xor(colsum[tn], 32, 64);
      const int n = n0 + wm * 128 + tn * 32 + kcol;
      if (h == 0 && (!MAPPED || n < p.Nn)) atomicAdd(p.db + (MAPPED ? (long long)group * p.strideDb : 0ll) + n, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gemm_tn4w: the plain wgrad (no row maps / groups) on the gemm_nt4w recipe: 256x256 fp32 output tile, FOUR waves (one per
// SIMD) owning 128x128 of it = 16 accumulators of v_mfma_f32_32x32x16_bf16 (256 registers), two fragment sets of 32
// transposed 8-byte reads.  LDS image and DMA as gemm_tn512 (32-row sub-stages of [G 32 x 512 B][X 32 x 512 B], ring of four;
// rows are whole 128-B lines already).  Sub-step u (32 MFMAs of 32 clk): the 32 ds_read_b64_tr_b16 of sub-step u+1 into the
// other fragment set (two per MFMA gap for the first 8 gaps, then one), the 8 DMA pieces of sub-stage u+4 into buffer
// u % 4 (one per 3 MFMAs), lgkmcnt(0) + vmcnt(16) + barrier after MFMA 27.  The DMA runs four sub-stages past the end of
// the range with clamped rows (nothing is computed from them).
// Requires M % 32 == 0, Nn % 256 == 0, Kk % 256 == 0.
// ---------------------------------------------------------------------------------------------
// MAPPED build: expert groups (row_off; ids with the group fastest), ONE gathered operand (row map staged through an 8192-entry LDS
// ring as in gemm_tn512), ragged ranges (rows past the end: G rows zeroed in LDS), Nn a multiple of 128.
// COLG build (plain rows only): n_groups COLUMN groups - group g multiplies the column blocks G[:, g*gcol_stride + (0..Nn)] and
// X[:, g*xcol_stride + (0..Kk)] over all M rows into dW + g*strideW (the per-image U_b^T A_b of the transposed local loss); Nn and Kk
// need not be multiples of 256 (columns past them are fetched from valid memory and never written back).
// Plain and COLG builds address rows through a 64-bit scalar base per sub-stage (per-lane offsets stay below 32 rows), so M * ld may
// exceed 4 GB (the transposed pair matrices: 45k rows of 426 KB).
// SCALE build (COLG only): dW_g += G_g^T diag(w_g) X_g - every G fragment is multiplied by its rows' weights (fp32 product, rounded to bf16
// as a separately stored w * G would be) between the transposed read and the MFMA: 16 VALU per fragment of eight, placed under the four
// MFMAs of the fragment before it.  The 32 weights of a sub-stage travel with it as one more (4-byte) DMA piece into a private 256-byte
// slot per wave and buffer.  With G == X (the Gram products of the local loss, dGm_b = A_b^T diag(d2) A_b) the second operand's DMA hits
// the lines the first one just fetched, and the U = d2 * A matrix is never stored.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define W_READ(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
template <bool MAPPED, bool COLG = false, bool SCALE = false>
__global__ __launch_bounds__(256) void gemm_tn4w_kernel(GemmTNArgs p) {
  static_assert(!(MAPPED && COLG), "column groups only in the plain build");
  static_assert(!SCALE || COLG, "row weights only in the column-group build");
  __shared__ __attribute__((aligned(128))) char smem[4 * SUB3 + (MAPPED ? 32768 : 0) + (SCALE ? 4096 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;          // wave tile: G columns wm*128.., X columns wn*128..
  int id = xcd_remap(blockIdx.x, gridDim.x);      // split-major ids: the tiles of one M range (same G / X rows) sit on one XCD
  const int ntile = p.tiles_n * p.tiles_k;
  int group = 0, ms, me, pid = 0;
  if (MAPPED && p.row_off) {
    // ranges of p.nsplit ROWS each, enumerated over the groups on the device (the group sizes are only known here): every
    // workgroup gets the same amount of work however unequal the groups are; the tiles of one range are adjacent ids
    const int tile = id % ntile;
    int rid = id / ntile;
    const int R = p.nsplit;
    ms = -1; me = -1;
    for (int gq = 0; gq < p.n_groups; ++gq) {
      const int a = p.row_off[gq], b = p.row_off[gq + 1];
      const int nr = (b - a + R - 1) / R;
      if (rid < nr) { group = gq; ms = a + rid * R; me = min(b, ms + R); break; }
      rid -= nr;
    }
    if (ms < 0) return;
    id = tile;
  } else {
    if (COLG) { group = id / (ntile * p.nsplit); id -= group * ntile * p.nsplit; }
    pid = id;                                     // = split * ntile + tile: the slot of this workgroup's partial tile (staged plain build)
    const int split = id / ntile; id -= split * ntile;
    const int chunk = (((p.M + 31) / 32 + p.nsplit - 1) / p.nsplit) * 32;
    ms = split * chunk; me = min(p.M, ms + chunk);
  }
  const int tile_n = id / p.tiles_k, tile_k = id - tile_n * p.tiles_k;
  if (ms >= me) return;
  const int U = (me - ms + 31) / 32;              // sub-steps; MAPPED: the last one may hold fewer than 32 valid rows
  const int tail = (me - ms) - (U - 1) * 32;
  const int n0 = tile_n * 256, k0 = tile_k * 256;

  // DMA piece i (0..7) of wave w fills LDS bytes [(i * 4 + w) * 1024, +1024) of the sub-stage: two 512-B rows.
  const int prow = (wid << 1) + (lane >> 5);      // row of piece 0 inside its 32-row half; piece i: + 8 * (i & 3)
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned col_g, col_x;
  {
    const int lc0 = (lane & 31) ^ ((prow & 3) << 2);
    const int gcol = n0 + lc0 * 8;
    col_g = (unsigned)(((MAPPED || COLG) && gcol >= p.Nn) ? n0 : gcol) * 2u;       // columns past Nn: fetched from valid memory, never written back
    if (COLG && p.g_chunk_w > 0) {
      // chunked columns (the image-major pair matrices: 208 columns per image, images g_chunk_stride apart): a 256-column tile spans at
      // most a few chunks; the first one's base is scalar (gbase below), the lane keeps its distance from it (< 2^32 bytes, host check)
      const int gc = (gcol >= p.Nn) ? n0 : gcol;
      const int c0 = n0 / p.g_chunk_w, cc = gc / p.g_chunk_w;
      col_g = (unsigned)((long long)(cc - c0) * p.g_chunk_stride + (gc - cc * p.g_chunk_w)) * 2u;
    }
    const int xcol = k0 + lc0 * 8;
    col_x = (unsigned)((COLG && xcol >= p.Kk) ? k0 : xcol) * 2u;
  }
  const int* rmap = MAPPED ? (p.g_rowmap ? p.g_rowmap : p.x_rowmap) : nullptr;
  int* lmap = (int*)(smem + 4 * SUB3);
  if (MAPPED && rmap) {
    for (int i = tid; i < min(me - ms, 8192); i += 256) lmap[i] = rmap[ms + i];
    __syncthreads();
  }
  const unsigned ldg2 = (unsigned)p.ldg * 2u, ldx2 = (unsigned)p.ldx * 2u;
  // per-lane byte offsets of the current sub-stage's piece 0 rows; pieces 1..3 add 8 rows each (scalar); the next
  // sub-stage adds 32 rows.  Rows are clamped to the range's last row (only past the end of the range).
  int lk = 0, wb = 0, rb = 0;
  unsigned sg[4], sx[4];
  // plain / COLG builds: whole sub-stages only (M % 32 == 0), per-lane offsets are those of the sub-stage's 32 rows and the sub-stage
  // itself is a scalar 64-bit base (clamped to the range's last sub-stage for the four the DMA runs ahead)
  const char* gbase = (const char*)p.G + (COLG ? (long long)group * p.gcol_stride * 2 : 0ll)
                      + ((COLG && p.g_chunk_w > 0) ? (long long)(n0 / p.g_chunk_w) * p.g_chunk_stride * 2 : 0ll);
  const char* xbase = (const char*)p.X + (COLG ? (long long)group * p.xcol_stride * 2 : 0ll);
  const char* gsub = gbase;
  const char* xsub = xbase;
  const char* wbase = SCALE ? (const char*)(p.gscale + (long long)group * p.gscale_gstride) : nullptr;
  const char* wsub = wbase;
  int wr0 = 0;
  const unsigned wl0 = lds0 + 4 * SUB3 + wid * 1024;              // SCALE: this wave's four 256-byte weight slots
  auto setup_rows = [&]() __attribute__((always_inline)) {
    if constexpr (!MAPPED) {
      // wave-uniform by construction; readfirstlane makes it provably scalar, so every DMA piece is `global_load_lds v_off, s[base]`
      // with the persistent per-lane offsets (a 64-bit VGPR address would be a TEMPORARY register pair: see keep_frags below)
      const long long r0 = ms + min(lk, (me - ms) / 32 - 1) * 32;
      auto uni = [](const char* q) -> const char* {
        const unsigned long long v = (unsigned long long)q;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const char*)(((unsigned long long)hi << 32) | lo);
      };
      gsub = uni(gbase + r0 * (long long)ldg2);
      xsub = uni(xbase + r0 * (long long)ldx2);
      if constexpr (SCALE) { wr0 = (int)r0; wsub = uni(wbase + r0 * (long long)p.gscale_ld * 4); }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ml = min(lk * 32 + prow + 8 * j, me - ms - 1);         // row inside the range (clamped past the end)
      const unsigned rm = (MAPPED && rmap) ? (unsigned)lmap[ml & 8191] : 0u;
      sg[j] = ((MAPPED && p.g_rowmap) ? rm : (unsigned)(ms + ml)) * ldg2 + col_g;
      sx[j] = ((MAPPED && p.x_rowmap) ? rm : (unsigned)(ms + ml)) * ldx2 + col_x;
    }
  };
  if constexpr (!MAPPED) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { sg[j] = (unsigned)(prow + 8 * j) * ldg2 + col_g; sx[j] = (unsigned)(prow + 8 * j) * ldx2 + col_x; }
  }
  unsigned ldsW = lds0 + wid * 1024;
  auto advance = [&]() __attribute__((always_inline)) {
    ++lk; wb = (wb + 1) & 3;
    setup_rows();
    ldsW = lds0 + wb * SUB3 + wid * 1024;
  };
  auto piece = [&](int i) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds(GLB_PTR((i < 4 ? gsub : xsub) + (i < 4 ? sg[i & 3] : sx[i & 3])),
                                     (__attribute__((address_space(3))) void*)(size_t)(ldsW + i * 4096), 16, 0, 0);
  };
  // SCALE: the weights of rows wr0 .. wr0 + 63 (this sub-stage's 32 and, unused, the next 32; clamped to the last row) into slot wb
  auto wpiece = [&]() __attribute__((always_inline)) {
    if constexpr (SCALE) {
      const unsigned wo = (unsigned)min(lane, p.M - 1 - wr0) * (unsigned)p.gscale_ld * 4u;
      __builtin_amdgcn_global_load_lds(GLB_PTR(wsub + wo), (__attribute__((address_space(3))) void*)(size_t)(wl0 + wb * 256), 4, 0, 0);
    }
  };

  f32x16_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_db = (p.db != nullptr) && tile_k == 0;

  // transposed-read addressing as in gemm_tn512
  const int h = lane >> 5, gam = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
  unsigned aa[8];                                 // G tiles 0..3, X tiles 0..3
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cg = wm * 16 + t * 4 + gam * 2 + (pp >> 1), cx = wn * 16 + t * 4 + gam * 2 + (pp >> 1);
    aa[t] = lds0 + (8 * h + q) * 512 + ((cg ^ (q << 2)) << 4) + ((pp & 1) << 3);
    aa[4 + t] = lds0 + 16384 + (8 * h + q) * 512 + ((cx ^ (q << 2)) << 4) + ((pp & 1) << 3);
  }
  struct Frags { u32x2_t gl[2][4], gh[2][4], xl[2][4], xh[2][4]; };
  Frags f0, f1;
  // SCALE: ONE set of row weights wv[ks][rows +0 / +4] - the current set's until its last G fragment is scaled (MFMA 24), then the next
  // set's (read there, landed at the wait after MFMA 27, first used under MFMA 28)
  u32x4_t wv[2][2];
  const unsigned wa = wl0 + 32 * (lane >> 5);     // this lane's eight rows of a 16-row half start at row 8 h
  auto read_w = [&](int buf) __attribute__((always_inline)) {
    if constexpr (SCALE) {
      const unsigned a = wa + buf * 256;
      W_READ(wv[0][0], a, 0); W_READ(wv[0][1], a, 16); W_READ(wv[1][0], a, 64); W_READ(wv[1][1], a, 80);
    }
  };
  // G fragment number i = ks * 4 + tn of a set times its rows' weights
  auto scale_frag = [&](Frags& f, auto ic) __attribute__((always_inline)) {
    if constexpr (SCALE) {
      constexpr int i = decltype(ic)::value, ks = i >> 2, tn = i & 3;
      auto sc = [](unsigned v, unsigned w0, unsigned w1) -> unsigned {
        return pack2bf(__uint_as_float(v << 16) * __uint_as_float(w0), __uint_as_float(v & 0xffff0000u) * __uint_as_float(w1));
      };
      const u32x4_t wl = wv[ks][0], wh = wv[ks][1];
      f.gl[ks][tn][0] = sc(f.gl[ks][tn][0], wl[0], wl[1]); f.gl[ks][tn][1] = sc(f.gl[ks][tn][1], wl[2], wl[3]);
      f.gh[ks][tn][0] = sc(f.gh[ks][tn][0], wh[0], wh[1]); f.gh[ks][tn][1] = sc(f.gh[ks][tn][1], wh[2], wh[3]);
    }
  };
  // read number g (0..31) of a sub-stage: operand (G / X), tile t, 16-row half ks, rows +0 / +4
  auto read_one = [&](Frags& f, unsigned bo, auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, t = (g >> 2) & 3, ks = (g >> 1) & 1, hi = g & 1;
    constexpr int off = ks * 8192 + hi * 2048;
    const unsigned a = aa[(g < 16 ? 0 : 4) + t] + bo;
    if constexpr (g < 16) { if constexpr (hi) TR_READ(f.gh[ks][t], a, off); else TR_READ(f.gl[ks][t], a, off); }
    else { if constexpr (hi) TR_READ(f.xh[ks][t], a, off); else TR_READ(f.xl[ks][t], a, off); }
  };
  auto frag = [&](u32x2_t lo, u32x2_t hi) -> bf16x8_t {
    const uint4 v = make_uint4(lo[0], lo[1], hi[0], hi[1]);
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto add_colsum = [&](Frags& f) __attribute__((always_inline)) {
    const s16x2_t ones = {(short)0x3F80, (short)0x3F80};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        const uint4 v = __builtin_bit_cast(uint4, frag(f.gl[ks][tn], f.gh[ks][tn]));
        float c = colsum[tn];
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.x), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.y), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.z), ones, c, false);
        c = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2_t, v.w), ones, c, false);
        colsum[tn] = c;
      }
  };
  int u_now = 0;
  // rows past the range hold clamped copies of real rows: zero their G half (every wave zeroes all of them itself, so its own
  // reads are ordered behind its own writes; zero G rows add nothing to dW or db)
  auto zero_tail = [&](int buf) __attribute__((always_inline)) {
    char* zb = smem + buf * SUB3;
    for (int z = tail * 32 + lane; z < 32 * 32; z += 64) *(uint4*)(zb + z * 16) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto substep = [&](Frags& c, Frags& n) __attribute__((always_inline)) {
    const unsigned bo = rb * SUB3;
    if (MAPPED && rmap && u_now >= 128 && (u_now & 127) == 0) {
      // sub-steps [u-128, u) are behind every cursor: their ring slots take the rows of sub-steps [u+128, u+256)
      for (int i = tid; i < 4096; i += 256) {
        const int r = (u_now + 128) * 32 + i;
        if (r < me - ms) lmap[r & 8191] = rmap[ms + r];
      }
    }
    if (MAPPED && u_now + 1 == U - 1 && tail < 32) zero_tail(rb);  // the sub-stage read during this sub-step is the ragged last one
    if (do_db && (u_now & 1) == wn) add_colsum(c);               // the two waves that hold the same G columns take turns
    auto one = [&](auto qc) __attribute__((always_inline)) {
      constexpr int qq = decltype(qc)::value, ks = qq >> 4, tn = (qq >> 2) & 3, tk = qq & 3;
      acc[tn][tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(c.gl[ks][tn], c.gh[ks][tn]), frag(c.xl[ks][tk], c.xh[ks][tk]), acc[tn][tk], 0, 0, 0);
      if constexpr (qq < 24) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (qq < 8) { read_one(n, bo, std::integral_constant<int, 2 * qq>{}); read_one(n, bo, std::integral_constant<int, 2 * qq + 1>{}); }
        else read_one(n, bo, std::integral_constant<int, qq + 8>{});
        if constexpr (qq % 3 == 2) piece(qq / 3);
        if constexpr (SCALE && qq == 0) wpiece();
        if constexpr (SCALE && qq % 4 == 0) scale_frag(c, std::integral_constant<int, qq / 4 + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SCALE && qq == 24) {
        __builtin_amdgcn_sched_barrier(0);
        scale_frag(c, std::integral_constant<int, 7>{});
        read_w(rb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (SCALE && qq == 28) {           // the next set's reads have landed (wait after MFMA 27): its first fragment
        __builtin_amdgcn_sched_barrier(0);
        scale_frag(n, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (qq == 25) { advance(); rb = (rb + 1) & 3; ++u_now; }
      if constexpr (qq == 27) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
      }
    };
    static_for<32>(one);
  };

  setup_rows();
  for (int v = 0; v < 4; ++v) {                   // sub-stages 0..3
    wpiece();
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(i);
    advance();
  }
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // (SCALE: nine pieces per sub-stage - the same counts wait for a little more)
  __builtin_amdgcn_s_barrier();                   // sub-stage 0 landed for every wave
  asm volatile("" ::: "memory");
  if (MAPPED && U == 1 && tail < 32) zero_tail(0);
  {
    auto rd = [&](auto gc) __attribute__((always_inline)) { read_one(f0, 0u, gc); };
    static_for<32>(rd);
    read_w(0);
  }
  rb = 1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // every wave has read buffer 0; sub-stage 1 landed
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  scale_frag(f0, std::integral_constant<int, 0>{});
  int u = 0;
  for (; u + 1 < U; u += 2) { substep(f0, f1); substep(f1, f0); }
  if (u < U) substep(f0, f1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA ran four sub-stages past the end
  // The last sub-step's transposed reads fetch a sub-stage nobody multiplies: to the compiler their destination registers are dead
  // from the asm statement on, but the LDS writes them when the data arrives.  Keep both fragment sets allocated up to here, so that no
  // temporary (an address pair of a DMA piece, say) is placed in a register with a read still in flight.
  auto keep_frags = [&](Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      asm volatile("" :: "v"(f.gl[ks][0]), "v"(f.gl[ks][1]), "v"(f.gl[ks][2]), "v"(f.gl[ks][3]), "v"(f.gh[ks][0]), "v"(f.gh[ks][1]), "v"(f.gh[ks][2]), "v"(f.gh[ks][3]));
      asm volatile("" :: "v"(f.xl[ks][0]), "v"(f.xl[ks][1]), "v"(f.xl[ks][2]), "v"(f.xl[ks][3]), "v"(f.xh[ks][0]), "v"(f.xh[ks][1]), "v"(f.xh[ks][2]), "v"(f.xh[ks][3]));
    }
  };
  if constexpr (SCALE) asm volatile("" :: "v"(wv[0][0]), "v"(wv[0][1]), "v"(wv[1][0]), "v"(wv[1][1]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  keep_frags(f0); keep_frags(f1);

  float* dW = p.dW + ((MAPPED || COLG) ? (long long)group * p.strideW : 0ll);
  const int kcol = lane & 31;
  bool staged = false;
  if constexpr (!MAPPED && !COLG) staged = p.partial != nullptr;
  if (staged) {
    // STAGED: 256 whole-wave 256-byte stores in accumulator order [wave][tn][tk][r][lane] - plain stores leave a CU five times faster than
    // the same bytes as fp32 atomics (which execute at the memory side, ~1.3 TB/s chip-wide: 64 MB per launch whatever the row count)
    float* part = p.partial + ((long long)pid << 16) + (wid << 14) + lane;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tk = 0; tk < 4; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((tn * 4 + tk) * 16 + r) * 64] = acc[tn][tk][r];
  } else {
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int tk = 0; tk < 4; ++tk) {
      const int k = k0 + wn * 128 + tk * 32 + kcol;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 128 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if ((!(MAPPED || COLG) || n < p.Nn) && (!COLG || k < p.Kk)) atomicAdd(dW + (long long)n * p.ldw + k, acc[tn][tk][r]);
      }
    }
  }
  if (do_db) {
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const float v = colsum[tn] + __shfl_xor(colsum[tn], 32, 64);
      const int n = n0 + wm * 128 + tn * 32 + kcol;
      if (h == 0 && (!(MAPPED || COLG) || n < p.Nn)) atomicAdd(p.db + ((MAPPED || COLG) ? (long long)group * p.strideDb : 0ll) + n, v);
    }
  }
}

extern "C" int medmoe_gemm_tn(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw, float* db, int M, int Nn, int Kk,
                              const int* x_rowmap, const int* g_rowmap, const int* row_off, int n_groups, long long strideW,
                              long long strideDb, int nsplit, hipStream_t stream);

// Second half of the STAGED plain wgrad: dW tile += sum over the row ranges of the partial tiles gemm_tn4w_kernel stored (fixed order:
// the result does not depend on which workgroup finished first, unlike the atomic form).  64 blocks per 256 x 256 tile, a float4 per thread.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int ldw, int ntile,
                                                        int tiles_k, int nvalid) {
  const int tile = blockIdx.x >> 6;
  const int e4 = (((blockIdx.x & 63) << 8) + threadIdx.x) << 2;
  float4 s = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < nvalid; ++sp) {
    const float4 v = *(const float4*)(partial + ((long long)(sp * ntile + tile) << 16) + e4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int lane = e4 & 63, r = (e4 >> 6) & 15, tk = (e4 >> 10) & 3, tn = (e4 >> 12) & 3, wid = e4 >> 14;
  const int wm = wid & 1, wn = wid >> 1, h = lane >> 5, kcol = lane & 31;
  const int tile_n = tile / tiles_k, tile_k = tile - tile_n * tiles_k;
  const int n = tile_n * 256 + wm * 128 + tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
  const int k = tile_k * 256 + wn * 128 + tk * 32 + kcol;
  float4* d = (float4*)(dW + (long long)n * ldw + k);
  float4 o = *d;
  o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
  *d = o;
}

// Plain wgrad dW[Nn, Kk] += G^T X (db += column sums of G) in the staged form when it applies - plain rows, Nn and Kk multiples of 256,
// M % 32 == 0 and >= 4096, `scratch` holding tiles x row ranges partial tiles of 65536 floats - else exactly medmoe_gemm_tn(nsplit 16).
// The caller owns `scratch` and must not share it between launches that may run concurrently (one per stream).
extern "C" int medmoe_gemm_tn_staged(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw, float* db, int M, int Nn, int Kk,
                                     float* scratch, long long scratch_floats, hipStream_t stream) {
  if (!G || !X || !dW) return MM_ERR_ARG;
  if (M <= 0 || Nn <= 0 || Kk <= 0 || (Nn % 8) || (Kk % 8) || (ldg % 8) || (ldx % 8)) return MM_ERR_SHAPE;
  const bool fit32 = (long long)M * ldg * 2 < (1ll << 32) && (long long)M * ldx * 2 < (1ll << 32);
  if (scratch && g_use_tn512 && g_use_tn4w && (M % 32) == 0 && (Nn % 256) == 0 && (Kk % 256) == 0 && M >= 4096 && fit32 && (ldw % 4) == 0) {
    GemmTNArgs p;
    p.partial = nullptr;
    p.G = (const bf16_t*)G; p.X = (const bf16_t*)X; p.dW = dW; p.db = db;
    p.x_rowmap = nullptr; p.g_rowmap = nullptr; p.row_off = nullptr; p.strideW = 0; p.strideDb = 0;
    p.M = M; p.Nn = Nn; p.Kk = Kk; p.ldg = ldg; p.ldx = ldx; p.ldw = ldw; p.gcol_stride = 0; p.xcol_stride = 0; p.g_chunk_w = 0; p.g_chunk_stride = 0;
    p.gscale = nullptr; p.gscale_gstride = 0; p.gscale_ld = 0;
    p.tiles_n = Nn / 256; p.tiles_k = Kk / 256;
    const int ntile = p.tiles_n * p.tiles_k;
    p.nsplit = max(1, min(256 / ntile, M / g_tn_min_rows));
    p.n_groups = 1;
    if (p.nsplit >= 2 && (long long)ntile * p.nsplit * 65536 <= scratch_floats) {
      p.partial = scratch;
      const int chunk = (((M + 31) / 32 + p.nsplit - 1) / p.nsplit) * 32;
      const int nvalid = (M + chunk - 1) / chunk;
      hipLaunchKernelGGL(gemm_tn4w_kernel<false>, dim3(ntile * p.nsplit), dim3(256), 0, stream, p);
      hipLaunchKernelGGL(tn_reduce_kernel, dim3(ntile * 64), dim3(256), 0, stream, scratch, dW, ldw, ntile, p.tiles_k, nvalid);
      return mm_check_launch();
    }
  }
  return medmoe_gemm_tn(G, ldg, X, ldx, dW, ldw, db, M, Nn, Kk, nullptr, nullptr, nullptr, 1, 0, 0, 16, stream);
}

// dW[g][Nn][Kk] += G_g^T X_g, G_g = G + g*gcol_stride, X_g = X + g*xcol_stride ([M][ld] views; column blocks of one matrix, separate
// matrices, or the same one for stride 0), over ALL M rows (fp32 atomics: zero dW first).  M % 32 == 0; one group's Nn x Kk block is tiled 256 x 256 (partial tiles masked).  With n_groups == 1 and strides 0
// this is the plain wgrad for operands whose M * ld exceeds 4 GB (the transposed local-loss matrices).
// g_chunk_w > 0: G's Nn columns are stored in chunks of g_chunk_w columns (a multiple of 8), chunk j at G + j * g_chunk_stride (the
// image-major pair matrices seen as ONE [M][B * HWp] operand: full 256-column tiles across images).
extern "C" int medmoe_gemm_tn_cols(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw, int M, int Nn, int Kk,
                                   int n_groups, long long gcol_stride, long long xcol_stride, long long strideW, int g_chunk_w,
                                   long long g_chunk_stride, hipStream_t stream) {
  if (!G || !X || !dW) return MM_ERR_ARG;
  if (M < 32 || (M % 32) || Nn <= 0 || Kk <= 0 || (Nn % 8) || (Kk % 8) || (ldg % 8) || (ldx % 8) || n_groups < 1) return MM_ERR_SHAPE;
  if ((gcol_stride % 8) || (xcol_stride % 8) || gcol_stride < 0 || xcol_stride < 0) return MM_ERR_SHAPE;
  if (32ll * ldg * 2 + 1024 >= (1ll << 32) || 32ll * ldx * 2 + 1024 >= (1ll << 32)) return MM_ERR_SHAPE;
  if ((g_chunk_w <= 0 && Nn > ldg) || Kk > ldx) return MM_ERR_SHAPE;      // a group's columns (and the clamped reads past them) stay inside its rows
  if (g_chunk_w > 0 && ((g_chunk_w % 8) || g_chunk_w > ldg || (g_chunk_stride % 8) || g_chunk_stride < 0 ||
                        (256 / g_chunk_w + 2) * g_chunk_stride * 2 + 32ll * ldg * 2 >= (1ll << 32))) return MM_ERR_SHAPE;
  GemmTNArgs p;
  p.partial = nullptr;
  p.G = (const bf16_t*)G; p.X = (const bf16_t*)X; p.dW = dW; p.db = nullptr;
  p.x_rowmap = nullptr; p.g_rowmap = nullptr; p.row_off = nullptr; p.strideW = strideW; p.strideDb = 0;
  p.M = M; p.Nn = Nn; p.Kk = Kk; p.ldg = ldg; p.ldx = ldx; p.ldw = ldw;
  p.tiles_n = (Nn + 255) / 256; p.tiles_k = (Kk + 255) / 256;
  p.n_groups = n_groups; p.gcol_stride = gcol_stride; p.xcol_stride = xcol_stride;
  p.g_chunk_w = g_chunk_w; p.g_chunk_stride = g_chunk_stride;
  p.gscale = nullptr; p.partial = nullptr; p.gscale_gstride = 0; p.gscale_ld = 0;
  const long long ntile = (long long)p.tiles_n * p.tiles_k * n_groups;
  p.nsplit = (int)max(1ll, min(256ll / ntile, (long long)M / g_tn_min_rows));
  hipLaunchKernelGGL((gemm_tn4w_kernel<false, true>), dim3((unsigned)(ntile * p.nsplit)), dim3(256), 0, stream, p);
  return mm_check_launch();
}

// dW[g][Nn][Nn] += A_g^T diag(w_g) A_g: weighted Gram products of the column blocks A_g = A + g*col_stride of one [M][lda] bf16 matrix,
// w_g[m] = w[g * w_gstride + m * w_ld] fp32 (every weight of rows 0..M-1 finite: it multiplies even all-zero rows).  The product
// w * A is rounded to bf16 before it is multiplied, as a stored copy would be.  fp32 atomics: zero dW first.  M % 32 == 0, Nn % 8 == 0.
// (The dGm of the GLoRIA local loss, sum over words of d2 a a^T per image, without the U = d2 * A matrix - losses.py:698-736 backward.)
extern "C" int medmoe_gemm_tn_gram(const void* A, int lda, const float* w, long long w_gstride, int w_ld, float* dW, int ldw, int M,
                                   int Nn, int n_groups, long long col_stride, long long strideW, hipStream_t stream) {
  if (!A || !w || !dW) return MM_ERR_ARG;
  if (M < 32 || (M % 32) || Nn <= 0 || (Nn % 8) || (lda % 8) || n_groups < 1 || Nn > lda || w_ld < 1 || w_gstride < 0) return MM_ERR_SHAPE;
  if ((col_stride % 8) || col_stride < 0 || 32ll * lda * 2 + 1024 >= (1ll << 32) || 64ll * w_ld * 4 >= (1ll << 32)) return MM_ERR_SHAPE;
  GemmTNArgs p;
  p.partial = nullptr;
  p.G = (const bf16_t*)A; p.X = (const bf16_t*)A; p.dW = dW; p.db = nullptr;
  p.x_rowmap = nullptr; p.g_rowmap = nullptr; p.row_off = nullptr; p.strideW = strideW; p.strideDb = 0;
  p.M = M; p.Nn = Nn; p.Kk = Nn; p.ldg = lda; p.ldx = lda; p.ldw = ldw;
  p.tiles_n = (Nn + 255) / 256; p.tiles_k = p.tiles_n;
  p.n_groups = n_groups; p.gcol_stride = col_stride; p.xcol_stride = col_stride;
  p.g_chunk_w = 0; p.g_chunk_stride = 0;
  p.gscale = w; p.gscale_gstride = w_gstride; p.gscale_ld = w_ld;
  const long long ntile = (long long)p.tiles_n * p.tiles_k * n_groups;
  p.nsplit = (int)max(1ll, min(256ll / ntile, (long long)M / g_tn_min_rows));
  hipLaunchKernelGGL((gemm_tn4w_kernel<false, true, true>), dim3((unsigned)(ntile * p.nsplit)), dim3(256), 0, stream, p);
  return mm_check_launch();
}

extern "C" int medmoe_gemm_tn(const void* G, int ldg, const void* X, int ldx, float* dW, int ldw,
                              float* db, int M, int Nn, int Kk, const int* x_rowmap,
                              const int* g_rowmap, const int* row_off, int n_groups, long long strideW,
                              long long strideDb, int nsplit, hipStream_t stream) {
  if (!G || !X || !dW) return MM_ERR_ARG;
  if (M <= 0 || Nn <= 0 || Kk <= 0 || (Nn % 8) || (Kk % 8) || (ldg % 8) || (ldx % 8)) return MM_ERR_SHAPE;
  if (n_groups < 1 || nsplit < 1) return MM_ERR_ARG;
  GemmTNArgs p;
  p.partial = nullptr;
  p.G = (const bf16_t*)G; p.X = (const bf16_t*)X; p.dW = dW; p.db = db;
  p.x_rowmap = x_rowmap; p.g_rowmap = g_rowmap; p.row_off = row_off; p.strideW = strideW; p.strideDb = strideDb;
  p.M = M; p.Nn = Nn; p.Kk = Kk; p.ldg = ldg; p.ldx = ldx; p.ldw = ldw; p.gcol_stride = 0; p.xcol_stride = 0; p.g_chunk_w = 0; p.g_chunk_stride = 0; p.gscale = nullptr; p.partial = nullptr; p.gscale_gstride = 0; p.gscale_ld = 0;
  const bool fit32 = (long long)M * ldg * 2 < (1ll << 32) && (long long)M * ldx * 2 < (1ll << 32);   // 32-bit DMA offsets
  if (g_use_tn512 && !x_rowmap && !g_rowmap && !row_off && n_groups == 1 && (M % 32) == 0 && (Nn % 256) == 0 && (Kk % 256) == 0 &&
      M >= 4096 && fit32) {
    p.tiles_n = Nn / 256; p.tiles_k = Kk / 256;
    const int ntile = p.tiles_n * p.tiles_k;
    p.nsplit = max(1, min(256 / ntile, M / g_tn_min_rows));   // ~256 workgroups, at least g_tn_min_rows / 32 sub-steps each
    p.n_groups = 1;
    if (g_use_tn4w) hipLaunchKernelGGL(gemm_tn4w_kernel<false>, dim3(ntile * p.nsplit), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_tn512_kernel<false>, dim3(ntile * p.nsplit), dim3(512), 0, stream, p);
    return mm_check_launch();
  }
  // grouped (row_off) and/or ONE row-mapped operand, ragged row counts, Nn a multiple of 128: the MAPPED build
  if (g_use_tn512 && !(x_rowmap && g_rowmap) && (Nn % 128) == 0 && (Kk % 256) == 0 && M / n_groups >= 4096 && fit32) {
    p.tiles_n = (Nn + 255) / 256; p.tiles_k = Kk / 256;
    const int ntile = p.tiles_n * p.tiles_k;
    // groups are unequal (router imbalance): ranges of ~g_tn_rows rows, several waves of workgroups, so that a large
    // group's work spreads over the chip (3 ranges per group measured slower than 50)
    p.n_groups = n_groups;
    if (g_use_tn4w && row_off) {
      // equal ranges of R rows over all groups, enumerated on the device.  tools/tn_rows_sweep4w.py (768x768 tiles, 8 groups,
      // us per launch; balanced / 2-of-8 groups): M = 401408: R 1024 887 / 860, 2048 635 / 651, 4096 540 / 547, 8192 502 / 629,
      // 16384 473 / 787;  M = 50176: 2048 95 / 143, 4096 104 / 109, 8192 149 / 182.  Long ranges lose on unequal groups: their few
      // output tiles take every range's fp32 atomics at the same time.
      long long R = g_tn_rows4w > 0 ? (g_tn_rows4w + 31) / 32 * 32 : 4096;
      p.nsplit = (int)R;
      const int max_ranges = (int)(M / R) + n_groups;
      hipLaunchKernelGGL(gemm_tn4w_kernel<true>, dim3(ntile * max_ranges), dim3(256), 0, stream, p);
      return mm_check_launch();
    }
    p.nsplit = max(1, M / n_groups / g_tn_rows);
    if (g_use_tn4w) hipLaunchKernelGGL(gemm_tn4w_kernel<true>, dim3(ntile * p.nsplit * n_groups), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(gemm_tn512_kernel<true>, dim3(ntile * p.nsplit * n_groups), dim3(512), 0, stream, p);
    return mm_check_launch();
  }
  p.tiles_n = (Nn + 127) / 128; p.tiles_k = (Kk + 127) / 128; p.nsplit = nsplit; p.n_groups = n_groups;
  const int grid = p.tiles_n * p.tiles_k * nsplit * n_groups;
  if (x_rowmap || g_rowmap) hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(grid), dim3(256), 0, stream, p);
  return mm_check_launch();
}
