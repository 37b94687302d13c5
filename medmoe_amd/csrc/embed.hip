// Token front-ends and text aggregation (all HBM-bound byte movers).
//   patchify        : [B,3,H,W] image -> [B*P, 3*p*p] bf16 rows in Conv2d weight order (c,py,px)
//   init_tokens     : x[b,t,:] = pos[t] (+ cls at t=0); the patch-embed GEMM then accumulates in place
//   pos_cls_grad    : dpos[t] += sum_b dx[b,t];  dcls += sum_b dx[b,0]
//   text_embed_ln   : LN(word[ids] + pos[t] + type[tt])   (BERT-style front-end, eps 1e-12)
//   text_aggregate  : last-4-layer sum + word-piece segment-sum + sentence mean
//                     (reference text_encoder.py:97-117 and :32-90)
#include "common.h"

template <typename InT>
__global__ __launch_bounds__(256) void patchify_kernel(const InT* __restrict__ img, bf16_t* __restrict__ out, int B,
                                                       int C, int H, int W, int p, int ld) {
  const int gw = W / p, gh = H / p;
  const long long total = (long long)B * C * H * gw;   // one thread per p-pixel run
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int gx = i % gw;
    long long r = i / gw;
    const int y = r % H; r /= H;
    const int c = r % C;
    const int b = r / C;
    const int gy = y / p, py = y - gy * p;
    const InT* src = img + (((long long)b * C + c) * H + y) * W + gx * p;
    bf16_t* dst = out + ((long long)b * gh * gw + gy * gw + gx) * ld + c * p * p + py * p;
    for (int k = 0; k < p; ++k) {
      if constexpr (sizeof(InT) == 4) dst[k] = f2bf(src[k]); else dst[k] = src[k];
    }
  }
}

// ld = row pitch of `out` in elements (>= C*patch*patch): patch sizes whose row length is not a multiple of the GEMM k-step
// (14x14x3 = 588) are written into rows padded to the next multiple of 64; the caller keeps the padding columns zero.
extern "C" int medmoe_patchify_ld(const void* img, void* out, int B, int C, int H, int W, int patch, int in_f32, int ld,
                                  hipStream_t stream) {
  if (!img || !out) return MM_ERR_ARG;
  if (B <= 0 || C <= 0 || patch <= 0 || (H % patch) || (W % patch) || ld < C * patch * patch || (ld % 8)) return MM_ERR_SHAPE;
  const long long total = (long long)B * C * H * (W / patch);
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  if (in_f32)
    hipLaunchKernelGGL(patchify_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)img, (bf16_t*)out, B, C, H, W, patch, ld);
  else
    hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(grid), dim3(256), 0, stream, (const bf16_t*)img, (bf16_t*)out, B, C, H, W, patch, ld);
  return mm_check_launch();
}

extern "C" int medmoe_patchify(const void* img, void* out, int B, int C, int H, int W, int patch, int in_f32,
                               hipStream_t stream) {
  if ((C * patch * patch) % 8) return MM_ERR_SHAPE;
  return medmoe_patchify_ld(img, out, B, C, H, W, patch, in_f32, C * patch * patch, stream);
}

__global__ __launch_bounds__(256) void init_tokens_kernel(bf16_t* __restrict__ x, const float* __restrict__ cls,
                                                          const float* __restrict__ pos, int B, int Nt, int D) {
  const long long total = (long long)B * Nt * D / 4;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const int col = e % D;
    const int t = (e / D) % Nt;
    float4 v = *(const float4*)(pos + (long long)t * D + col);
    if (t == 0) {
      const float4 c = *(const float4*)(cls + col);
      v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
    }
    uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w);
    *(uint2*)(x + e) = o;
  }
}

extern "C" int medmoe_init_tokens(void* x, const float* cls, const float* pos, int B, int Nt, int D,
                                  hipStream_t stream) {
  if (!x || !cls || !pos) return MM_ERR_ARG;
  if (B <= 0 || Nt <= 0 || D <= 0 || (D % 4)) return MM_ERR_SHAPE;
  const long long total = (long long)B * Nt * D / 4;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(init_tokens_kernel, dim3(grid), dim3(256), 0, stream, (bf16_t*)x, cls, pos, B, Nt, D);
  return mm_check_launch();
}

// one block per (token t, 512-column slab), two columns per thread; deterministic b-ascending sum (eight loads in flight,
// added in order)
__global__ __launch_bounds__(256) void pos_cls_grad_kernel(const bf16_t* __restrict__ dx, float* __restrict__ dpos,
                                                           float* __restrict__ dcls, int B, int Nt, int D) {
  const int t = blockIdx.x, col = (blockIdx.y * 256 + threadIdx.x) * 2;
  if (col >= D) return;
  const bf16_t* src = dx + (long long)t * D + col;
  const long long bs = (long long)Nt * D;
  float s0 = 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const uint32_t*)(src + (b + u) * bs);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += __uint_as_float(v[u] << 16); s1 += __uint_as_float(v[u] & 0xffff0000u); }
  }
  for (; b < B; ++b) {
    const uint32_t v = *(const uint32_t*)(src + b * bs);
    s0 += __uint_as_float(v << 16); s1 += __uint_as_float(v & 0xffff0000u);
  }
  dpos[(long long)t * D + col] += s0; dpos[(long long)t * D + col + 1] += s1;
  if (t == 0) { dcls[col] += s0; dcls[col + 1] += s1; }
}

extern "C" int medmoe_pos_cls_grad(const void* dx, float* dpos, float* dcls, int B, int Nt, int D,
                                   hipStream_t stream) {
  if (!dx || !dpos || !dcls) return MM_ERR_ARG;
  if (B <= 0 || Nt <= 0 || D <= 0 || (D % 2)) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(pos_cls_grad_kernel, dim3(Nt, (D + 511) / 512), dim3(256), 0, stream, (const bf16_t*)dx, dpos,
                     dcls, B, Nt, D);
  return mm_check_launch();
}

// one wave per token; D <= 2048
__global__ __launch_bounds__(256) void text_embed_ln_kernel(const int* __restrict__ ids, const int* __restrict__ tts,
                                                            const float* __restrict__ word, const float* __restrict__ pos,
                                                            const float* __restrict__ type, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                            int rows, int T, int D, int vocab, float eps, const int* __restrict__ src_of_row,
                                                            const int* __restrict__ rows_dev) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n4 = D >> 2;
  if (rows_dev) rows = min(rows, *rows_dev);              // packed rows of a variable-length batch (text_pack)
  for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
    const int src = src_of_row ? src_of_row[row] : row;   // token (b, t) = src this packed row holds
    const int t = src % T;
    int id = ids[src]; id = min(max(id, 0), vocab - 1);
    const int tt = tts ? min(max(tts[src], 0), 1) : 0;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < n4) {
        const float4 a = *(const float4*)(word + (long long)id * D + c * 4);
        const float4 p = *(const float4*)(pos + (long long)t * D + c * 4);
        const float4 y = *(const float4*)(type + (long long)tt * D + c * 4);
        v[i] = make_float4(a.x + p.x + y.x, a.y + p.y + y.y, a.z + p.z + y.z, a.w + p.w + y.w);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (lane + i * 64 < n4) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        sq += a * a + b * b + c * c + d * d;
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < n4) {
        const float4 g = *(const float4*)(gamma + c * 4), b = *(const float4*)(beta + c * 4);
        uint2 o;
        o.x = pack2bf((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y);
        o.y = pack2bf((v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
        *(uint2*)(out + (long long)row * D + c * 4) = o;
      }
    }
  }
}

extern "C" int medmoe_text_embed_ln(const int* ids, const int* type_ids, const float* word, const float* pos,
                                    const float* type, const float* gamma, const float* beta, void* out,
                                    int B, int T, int D, int vocab, float eps, hipStream_t stream) {
  if (!ids || !word || !pos || !type || !gamma || !beta || !out) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || D <= 0 || (D % 4) || D > 2048 || vocab <= 0) return MM_ERR_SHAPE;
  const int rows = B * T;
  const int grid = min((rows + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(text_embed_ln_kernel, dim3(grid), dim3(256), 0, stream, ids, type_ids, word, pos, type, gamma,
                     beta, (bf16_t*)out, rows, T, D, vocab, eps, (const int*)nullptr, (const int*)nullptr);
  return mm_check_launch();
}

// Variable-length text batches: the tower runs on the tokens with attention mask 1 only, packed in (caption, position) order.
//   tok_row[b*T + t] = packed row of token (b, t) or -1; src_of_row[r] = b*T + t of packed row r; seq_off[b] = first packed row of
//   caption b (seq_off[B] = count[0] = number of packed rows).  One workgroup: B <= 1024 captions.
__global__ __launch_bounds__(1024) void text_pack_kernel(const unsigned char* __restrict__ mask, int* __restrict__ tok_row,
                                                         int* __restrict__ src_of_row, int* __restrict__ seq_off, int* __restrict__ count,
                                                         int B, int T) {
  __shared__ int sh[1024];
  const int b = threadIdx.x;
  int n = 0;
  if (b < B)
    for (int t = 0; t < T; ++t) n += mask[b * T + t] ? 1 : 0;
  sh[b] = n;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                    // inclusive scan
    const int v = (b >= o) ? sh[b - o] : 0;
    __syncthreads();
    sh[b] += v;
    __syncthreads();
  }
  if (b < B) {
    int r = sh[b] - n;
    seq_off[b] = r;
    for (int t = 0; t < T; ++t) {
      if (mask[b * T + t]) { tok_row[b * T + t] = r; src_of_row[r] = b * T + t; ++r; }
      else tok_row[b * T + t] = -1;
    }
    if (b == B - 1) { seq_off[B] = sh[b]; count[0] = sh[b]; }
  }
}

extern "C" int medmoe_text_pack(const unsigned char* mask, int* tok_row, int* src_of_row, int* seq_off, int* count, int B, int T,
                                hipStream_t stream) {
  if (!mask || !tok_row || !src_of_row || !seq_off || !count) return MM_ERR_ARG;
  if (B <= 0 || B > 1024 || T <= 0) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(text_pack_kernel, dim3(1), dim3(1024), 0, stream, mask, tok_row, src_of_row, seq_off, count, B, T);
  return mm_check_launch();
}

// Word-piece segment map + caption lengths on the device (reference text_encoder.py:45-76 token loop, medmoe_module.py:221-223), one thread
// per caption: seg[b,t] = word index of token t (-1 = dropped: behind [SEP], or the unflushed last word of a caption without [SEP]);
// cap[b] = (#words whose first piece does not start with '[') + 1.  A token opens a new word unless it is a '##' continuation piece;
// token 0 always opens word 0.  ids are int64 (ids64 != 0) or int32.
__global__ __launch_bounds__(256) void segment_map_kernel(const void* __restrict__ ids_, int ids64, const unsigned char* __restrict__ is_cont,
                                                          const unsigned char* __restrict__ starts_bracket, int* __restrict__ seg,
                                                          int* __restrict__ cap, int B, int T, int vocab, int sep_id) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  auto id_at = [&](int t) -> int {
    const long long v = ids64 ? ((const long long*)ids_)[(long long)b * T + t] : (long long)((const int*)ids_)[(long long)b * T + t];
    return (int)min(max(v, 0ll), (long long)vocab - 1);
  };
  int sep_pos = T;
  for (int t = 0; t < T; ++t)
    if (id_at(t) == sep_id) { sep_pos = t; break; }
  int c = -1, at_sep = 0;
  for (int t = 0; t < T; ++t) {
    const int id = id_at(t);
    const bool st = t == 0 || (t <= sep_pos && !is_cont[id]);
    c += st ? 1 : 0;
    seg[(long long)b * T + t] = c;
    if (t == sep_pos) at_sep = c;
  }
  const int n_words = sep_pos < T ? at_sep + 1 : c;                 // without a [SEP] the loop never flushes the last bank: that word is dropped
  int prev = -1, n_real = 0;
  for (int t = 0; t < T; ++t) {
    const int w = seg[(long long)b * T + t];
    const bool keep = t <= sep_pos && w < n_words;
    if (keep && w != prev) {                                         // first piece of a kept word
      n_real += starts_bracket[id_at(t)] ? 0 : 1;
      prev = w;
    }
    if (!keep) seg[(long long)b * T + t] = -1;
  }
  cap[b] = n_real + 1;
}

// The same map with ONE WAVE per caption (T <= 128: lane l owns tokens 2l and 2l + 1): the id -> flag lookups of all tokens in parallel, the
// first [SEP] by a wave minimum, the word index as a wave prefix sum of the "opens a word" flags.  The thread-per-caption form above walks
// three dependent lookup chains of T tokens: 133 us at 128 captions of 77 tokens, whatever the batch.
__global__ __launch_bounds__(256) void segment_map_wave_kernel(const void* __restrict__ ids_, int ids64, const unsigned char* __restrict__ is_cont,
                                                               const unsigned char* __restrict__ starts_bracket, int* __restrict__ seg,
                                                               int* __restrict__ cap, int B, int T, int vocab, int sep_id) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;                                              // whole waves leave
  int id[2], cont[2], brk[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = 2 * lane + u;
    id[u] = -1; cont[u] = 0; brk[u] = 0;
    if (t < T) {
      const long long v = ids64 ? ((const long long*)ids_)[(long long)b * T + t] : (long long)((const int*)ids_)[(long long)b * T + t];
      id[u] = (int)min(max(v, 0ll), (long long)vocab - 1);
      cont[u] = is_cont[id[u]]; brk[u] = starts_bracket[id[u]];
    }
  }
  int sep_pos = T;
#pragma unroll
  for (int u = 1; u >= 0; --u) if (2 * lane + u < T && id[u] == sep_id) sep_pos = 2 * lane + u;
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) sep_pos = min(sep_pos, __shfl_xor(sep_pos, m, 64));
  int st[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) { const int t = 2 * lane + u; st[u] = (t < T && (t == 0 || (t <= sep_pos && !cont[u]))) ? 1 : 0; }
  int incl = st[0] + st[1];                                        // inclusive scan over the lanes
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { const int v = __shfl_up(incl, m, 64); if (lane >= m) incl += v; }
  const int base = incl - st[0] - st[1];                           // opens before this lane's tokens
  int c[2] = {base + st[0] - 1, base + st[0] + st[1] - 1};         // word index of each token (c = -1 + inclusive count)
  // at_sep = c(sep_pos); without a [SEP] the last word is never flushed: n_words = c(T - 1)
  const int probe = sep_pos < T ? sep_pos : T - 1;
  int at = (probe >> 1) == lane ? c[probe & 1] : 0;
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) at += __shfl_xor(at, m, 64);
  const int n_words = sep_pos < T ? at + 1 : at;
  int n_real = 0;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = 2 * lane + u;
    if (t < T) {
      const bool keep = t <= sep_pos && c[u] < n_words;
      if (keep && st[u] && !brk[u]) n_real += 1;                  // first piece of a kept word that does not start with '['
      seg[(long long)b * T + t] = keep ? c[u] : -1;
    }
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) n_real += __shfl_xor(n_real, m, 64);
  if (lane == 0) cap[b] = n_real + 1;
}

extern "C" int medmoe_segment_map(const void* ids, int ids64, const unsigned char* is_cont, const unsigned char* starts_bracket, int* seg,
                                  int* cap, int B, int T, int vocab, int sep_id, hipStream_t stream) {
  if (!ids || !is_cont || !starts_bracket || !seg || !cap) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || vocab <= 0 || sep_id < 0 || sep_id >= vocab) return MM_ERR_SHAPE;
  if (T <= 128) {
    hipLaunchKernelGGL(segment_map_wave_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, ids, ids64, is_cont, starts_bracket, seg, cap, B, T,
                       vocab, sep_id);
    return mm_check_launch();
  }
  hipLaunchKernelGGL(segment_map_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, ids, ids64, is_cont, starts_bracket, seg, cap, B, T, vocab,
                     sep_id);
  return mm_check_launch();
}

// text_embed_ln on the packed rows: out[r] = LN(word[ids[src]] + pos[t] + type[tt]) for r < *count
extern "C" int medmoe_text_embed_ln_packed(const int* ids, const int* type_ids, const float* word, const float* pos, const float* type,
                                           const float* gamma, const float* beta, void* out, int B, int T, int D, int vocab, float eps,
                                           const int* src_of_row, const int* count, hipStream_t stream) {
  if (!ids || !word || !pos || !type || !gamma || !beta || !out || !src_of_row || !count) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || D <= 0 || (D % 4) || D > 2048 || vocab <= 0) return MM_ERR_SHAPE;
  const int rows = B * T;
  const int grid = min((rows + 3) / 4, 256 * 8);
  hipLaunchKernelGGL(text_embed_ln_kernel, dim3(grid), dim3(256), 0, stream, ids, type_ids, word, pos, type, gamma,
                     beta, (bf16_t*)out, rows, T, D, vocab, eps, src_of_row, count);
  return mm_check_launch();
}

// block = (b, 512-column slab); each thread owns 2 columns and walks t ascending.
// seg[b,t] = word index of token t (non-decreasing along t, -1 = dropped token).
__global__ __launch_bounds__(256) void text_aggregate_kernel(const bf16_t* __restrict__ h0, const bf16_t* __restrict__ h1,
                                                             const bf16_t* __restrict__ h2, const bf16_t* __restrict__ h3,
                                                             int n_layers, const int* __restrict__ seg,
                                                             bf16_t* __restrict__ word16, float* __restrict__ word32,
                                                             float* __restrict__ sent, int T, int D, const int* __restrict__ tok_row) {
  // the caption's segment map and token rows once into LDS, then the hidden-state loads of FOUR tokens in flight at a time: the first form
  // read seg -> row -> four layers one token after the other, a chain of ~77 memory round trips per caption (237 us whatever the batch)
  __shared__ int s_seg[256], s_row[256];
  const int b = blockIdx.x, col = blockIdx.y * 512 + threadIdx.x * 2;
  for (int t = threadIdx.x; t < min(T, 256); t += 256) {
    s_seg[t] = seg[b * T + t];
    s_row[t] = tok_row ? tok_row[b * T + t] : b * T + t;
  }
  __syncthreads();
  if (col >= D) return;
  const bf16_t* hs[4] = {h0, h1, h2, h3};
  float a0 = 0.f, a1 = 0.f, s0 = 0.f, s1 = 0.f;
  int cur = -1, written = 0;
  auto flush = [&](int w) {
    const long long o = ((long long)b * T + w) * D + col;
    if (word16) *(uint32_t*)(word16 + o) = pack2bf(a0, a1);
    if (word32) { word32[o] = a0; word32[o + 1] = a1; }
    s0 += a0; s1 += a1;
    a0 = 0.f; a1 = 0.f;
  };
  for (int t0 = 0; t0 < T; t0 += 4) {
    int wv[4];
    float x0[4], x1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u;
      wv[u] = t < T ? (t < 256 ? s_seg[t] : seg[b * T + t]) : -1;
      x0[u] = 0.f; x1[u] = 0.f;
      if (wv[u] >= 0) {
        const long long o = (long long)(t < 256 ? s_row[t] : (tok_row ? tok_row[b * T + t] : b * T + t)) * D + col;   // packed hidden states: the token's row
#pragma unroll
        for (int l = 0; l < 4; ++l) {
          if (l < n_layers) {
            const uint32_t v = *(const uint32_t*)(hs[l] + o);
            x0[u] += __uint_as_float(v << 16); x1[u] += __uint_as_float(v & 0xffff0000u);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int w = wv[u];
      if (w < 0) continue;
      if (w != cur) {
        if (cur >= 0) { flush(cur); written = cur + 1; }
        cur = w;
      }
      a0 += x0[u]; a1 += x1[u];
    }
  }
  if (cur >= 0) { flush(cur); written = cur + 1; }
  for (int w = written; w < T; ++w) {          // zero padding (text_encoder.py:78-81)
    const long long o = ((long long)b * T + w) * D + col;
    if (word16) *(uint32_t*)(word16 + o) = 0u;
    if (word32) { word32[o] = 0.f; word32[o + 1] = 0.f; }
  }
  sent[(long long)b * D + col] = s0 / (float)T;       // mean over ALL T positions, then summed over layers
  sent[(long long)b * D + col + 1] = s1 / (float)T;
}

extern "C" int medmoe_text_aggregate(const void* h0, const void* h1, const void* h2, const void* h3, int n_layers,
                                     const int* seg, void* word_bf16, float* word_f32, float* sent, int B, int T,
                                     int D, hipStream_t stream) {
  if (!h0 || !seg || !sent || n_layers < 1 || n_layers > 4) return MM_ERR_ARG;
  if ((n_layers > 1 && !h1) || (n_layers > 2 && !h2) || (n_layers > 3 && !h3)) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || D <= 0 || (D % 2)) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(text_aggregate_kernel, dim3(B, (D + 511) / 512), dim3(256), 0, stream, (const bf16_t*)h0,
                     (const bf16_t*)h1, (const bf16_t*)h2, (const bf16_t*)h3, n_layers, seg, (bf16_t*)word_bf16,
                     word_f32, sent, T, D, (const int*)nullptr);
  return mm_check_launch();
}

// the same on PACKED hidden states (text_pack): token (b, t) lives in row tok_row[b*T + t]; tokens with seg >= 0 must have a row
extern "C" int medmoe_text_aggregate_packed(const void* h0, const void* h1, const void* h2, const void* h3, int n_layers,
                                            const int* seg, const int* tok_row, void* word_bf16, float* word_f32, float* sent, int B, int T,
                                            int D, hipStream_t stream) {
  if (!h0 || !seg || !sent || !tok_row || n_layers < 1 || n_layers > 4) return MM_ERR_ARG;
  if ((n_layers > 1 && !h1) || (n_layers > 2 && !h2) || (n_layers > 3 && !h3)) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || D <= 0 || (D % 2)) return MM_ERR_SHAPE;
  hipLaunchKernelGGL(text_aggregate_kernel, dim3(B, (D + 511) / 512), dim3(256), 0, stream, (const bf16_t*)h0,
                     (const bf16_t*)h1, (const bf16_t*)h2, (const bf16_t*)h3, n_layers, seg, (bf16_t*)word_bf16,
                     word_f32, sent, T, D, tok_row);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Trainable text tower (reference freeze_bert: false): backward of the word-piece aggregation and of the embedding front-end.
// ---------------------------------------------------------------------------------------------
// Aggregation backward.  Forward (text_aggregate_kernel): word[b, w] = sum over the selected layers and over the tokens t with seg[b, t] = w of
// the hidden states; sent[b] = (1 / T) sum_w word[b, w].  Every selected layer's hidden state therefore receives the SAME gradient:
// dH[b, t] = d_word[b, seg[b, t]] + d_sent[b] / T for kept tokens, 0 for dropped ones.
__global__ __launch_bounds__(256) void text_aggregate_bwd_kernel(const float* __restrict__ d_word, const float* __restrict__ d_sent,
                                                                 const int* __restrict__ seg, bf16_t* __restrict__ dH, int rows, int T, int D) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n4 = D >> 2;
  const float invT = 1.f / (float)T;
  for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
    const int b = row / T;
    const int w = seg[row];
    for (int c = lane; c < n4; c += 64) {
      uint2 o = make_uint2(0u, 0u);
      if (w >= 0) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d_word) v = *(const float4*)(d_word + ((long long)b * T + w) * D + c * 4);
        if (d_sent) {
          const float4 s = *(const float4*)(d_sent + (long long)b * D + c * 4);
          v.x += s.x * invT; v.y += s.y * invT; v.z += s.z * invT; v.w += s.w * invT;
        }
        o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w);
      }
      *(uint2*)(dH + (long long)row * D + c * 4) = o;
    }
  }
}

extern "C" int medmoe_text_aggregate_bwd(const float* d_word, const float* d_sent, const int* seg, void* dH, int B, int T, int D,
                                         hipStream_t stream) {
  if (!seg || !dH || (!d_word && !d_sent)) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || D <= 0 || (D % 4)) return MM_ERR_SHAPE;
  const int rows = B * T;
  hipLaunchKernelGGL(text_aggregate_bwd_kernel, dim3(min((rows + 3) / 4, 256 * 8)), dim3(256), 0, stream, d_word, d_sent, seg, (bf16_t*)dH, rows, T, D);
  return mm_check_launch();
}

// Embedding front-end backward: y = LN(word[id] + pos[t] + type[tt]) (text_embed_ln_kernel) - one wave per token recomputes the sum and its
// statistics, forms dx = rstd (dy g - mean(dy g) - xhat mean(dy g xhat)), writes it (fp32: the caller reduces it over the batch for the
// position / token-type tables) and adds it to the word table's row (fp32 atomics; a row is hit once per occurrence of its id);
// dgamma / dbeta partials meet in LDS, one atomic per column and workgroup.
__global__ __launch_bounds__(256) void text_embed_ln_bwd_kernel(const int* __restrict__ ids, const int* __restrict__ tts, const float* __restrict__ word,
                                                                const float* __restrict__ pos, const float* __restrict__ type,
                                                                const float* __restrict__ gamma, const bf16_t* __restrict__ dy,
                                                                float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ g_word, int rows, int T, int D, int vocab, float eps) {
  __shared__ float red[2][4][2048];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int n4 = D >> 2;
  float ag[8][4], ab[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }
  for (int row = blockIdx.x * 4 + wid; row < rows; row += gridDim.x * 4) {
    const int t = row % T;
    int id = ids[row]; id = min(max(id, 0), vocab - 1);
    const int tt = tts ? min(max(tts[row], 0), 1) : 0;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < n4) {
        const float4 a = *(const float4*)(word + (long long)id * D + c * 4);
        const float4 p = *(const float4*)(pos + (long long)t * D + c * 4);
        const float4 y = *(const float4*)(type + (long long)tt * D + c * 4);
        v[i] = make_float4(a.x + p.x + y.x, a.y + p.y + y.y, a.z + p.z + y.z, a.w + p.w + y.w);
        s += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (lane + i * 64 < n4) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        sq += a * a + b * b + c * c + d * d;
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
    float dgv[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < n4) {
        const uint2 r = *(const uint2*)(dy + (long long)row * D + c * 4);
        const float d4[4] = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
        const float4 g = *(const float4*)(gamma + c * 4);
        const float g4[4] = {g.x, g.y, g.z, g.w};
        float* xv = (float*)&v[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mean) * rstd;
          xv[e] = xh;                                       // v <- xhat
          ab[i][e] += d4[e]; ag[i][e] += d4[e] * xh;
          dgv[i][e] = d4[e] * g4[e];
          s1 += dgv[i][e]; s2 += dgv[i][e] * xh;
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + i * 64;
      if (c < n4) {
        const float* xv = (const float*)&v[i];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (dgv[i][e] - m1 - xv[e] * m2);
        *(float4*)(dx + (long long)row * D + c * 4) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(g_word + (long long)id * D + c * 4 + e, o[e]);
      }
    }
  }
  // dgamma / dbeta: the four waves' partials through LDS, one atomic per column and workgroup
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i * 64 >= n4) break;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][wid][i * 256 + lane * 4 + e] = ag[i][e]; red[1][wid][i * 256 + lane * 4 + e] = ab[i][e]; }
  }
  __syncthreads();
  for (int col = threadIdx.x; col < D; col += 256) {
    atomicAdd(dgamma + col, red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col]);
    atomicAdd(dbeta + col, red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col]);
  }
}

extern "C" int medmoe_text_embed_ln_bwd(const int* ids, const int* type_ids, const float* word, const float* pos, const float* type,
                                        const float* gamma, const void* dy, float* dx, float* dgamma, float* dbeta, float* g_word, int B, int T,
                                        int D, int vocab, float eps, hipStream_t stream) {
  if (!ids || !word || !pos || !type || !gamma || !dy || !dx || !dgamma || !dbeta || !g_word) return MM_ERR_ARG;
  if (B <= 0 || T <= 0 || D <= 0 || (D % 4) || D > 2048 || vocab <= 0) return MM_ERR_SHAPE;
  const int rows = B * T;
  hipLaunchKernelGGL(text_embed_ln_bwd_kernel, dim3(min((rows + 3) / 4, 256 * 4)), dim3(256), 0, stream, ids, type_ids, word, pos, type, gamma,
                     (const bf16_t*)dy, dx, dgamma, dbeta, g_word, rows, T, D, vocab, eps);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Image preprocessing on the device (the reference runs HF AutoImageProcessor on PIL lists on the
// CPU every step, swin.py:131): uint8 HWC images of any size -> resize to Ho x Wo -> x rescale -> (x - mean) / std ->
// bf16 [B,3,Ho,Wo], the layout medmoe_patchify reads.  Resize = bilinear, half-pixel centres, no antialias
// (torch.nn.functional.interpolate(mode="bilinear", align_corners=False)).  The HF processor's PIL bicubic filter is a
// third-party dependency that is absent offline: parity with it is unpinned; the test pins this kernel to the torch op.
// One thread per output pixel (3 channels): HBM-bound, 3 B read ~4x + 6 B written per pixel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* const* __restrict__ src, const int* __restrict__ hw,
                                                         bf16_t* __restrict__ dst, int B, int Ho, int Wo, float rescale,
                                                         float m0, float m1, float m2, float is0, float is1, float is2) {
  const long long total = (long long)B * Ho * Wo;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = i % Wo, y = (i / Wo) % Ho, b = i / ((long long)Wo * Ho);
    const int Hs = hw[2 * b], Ws = hw[2 * b + 1];
    const unsigned char* s = src[b];
    const float fy = fmaxf(((float)y + 0.5f) * ((float)Hs / (float)Ho) - 0.5f, 0.f);
    const float fx = fmaxf(((float)x + 0.5f) * ((float)Ws / (float)Wo) - 0.5f, 0.f);
    const int y0 = min((int)fy, Hs - 1), x0 = min((int)fx, Ws - 1);
    const int y1 = min(y0 + 1, Hs - 1), x1 = min(x0 + 1, Ws - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const unsigned char* p00 = s + ((long long)y0 * Ws + x0) * 3;
    const unsigned char* p01 = s + ((long long)y0 * Ws + x1) * 3;
    const unsigned char* p10 = s + ((long long)y1 * Ws + x0) * 3;
    const unsigned char* p11 = s + ((long long)y1 * Ws + x1) * 3;
    const float mean[3] = {m0, m1, m2}, istd[3] = {is0, is1, is2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = (float)p00[c] + wx * ((float)p01[c] - (float)p00[c]);
      const float bot = (float)p10[c] + wx * ((float)p11[c] - (float)p10[c]);
      const float v = (top + wy * (bot - top)) * rescale;
      dst[(((long long)b * 3 + c) * Ho + y) * Wo + x] = f2bf((v - mean[c]) * istd[c]);
    }
  }
}

extern "C" int medmoe_preprocess(const void* const* src_ptrs, const int* src_hw, void* dst, int B, int Ho, int Wo, float rescale,
                                 const float* mean3, const float* std3, hipStream_t stream) {
  if (!src_ptrs || !src_hw || !dst || !mean3 || !std3) return MM_ERR_ARG;
  if (B <= 0 || Ho <= 0 || Wo <= 0) return MM_ERR_SHAPE;
  for (int c = 0; c < 3; ++c) if (!(std3[c] > 0.f)) return MM_ERR_ARG;
  const long long total = (long long)B * Ho * Wo;
  const int grid = (int)min((total + 255) / 256, (long long)256 * 16);
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid), dim3(256), 0, stream, (const unsigned char* const*)src_ptrs, src_hw, (bf16_t*)dst,
                     B, Ho, Wo, rescale, mean3[0], mean3[1], mean3[2], 1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  return mm_check_launch();
}

// ---------------------------------------------------------------------------------------------
// BICUBIC preprocessing = what the reference's image processor does (swin.py:131: HF AutoImageProcessor of
// microsoft/swin-tiny-patch4-window7-224 -> PIL.Image.resize(resample=BICUBIC) on uint8, x 1/255, ImageNet mean / std).
// Pillow's resampler (third-party, libImaging/Resample.c; importable here, so the test pins this kernel to it bit for bit on
// the uint8 stage): separable, antialiased (filter support 2 * max(scale, 1)), cubic a = -0.5, HORIZONTAL pass first then
// vertical, each pass in 32-bit fixed point with 22 fractional bits, rounded (+ 1 << 21) and clipped to uint8 in between.
// The coefficient tables (bounds + int32 weights per output coordinate) are built on the host (medmoe_amd/data.py mirrors
// precompute_coeffs / normalize_coeffs_8bpc) once per source size and cached.
//   meta[b] = {Hs, Ws, ksize_h, ksize_v, off_bounds_h, off_k_h, off_bounds_v, off_k_v, off_tmp}   (offsets into coef / tmp)
// ---------------------------------------------------------------------------------------------
#define PRE_PREC 22
__device__ __forceinline__ int clip8i(int v) { return min(max(v >> PRE_PREC, 0), 255); }

__global__ __launch_bounds__(256) void bicubic_h_kernel(const unsigned char* const* __restrict__ src, const long long* __restrict__ meta,
                                                        const int* __restrict__ coef, unsigned char* __restrict__ tmp, int B, int Wo,
                                                        long long total_rows) {
  // one thread per (image row, output column): 3 channels
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total_rows * Wo; i += (long long)gridDim.x * 256) {
    const int xx = i % Wo;
    const long long row = i / Wo;
    int b = 0;                                           // which image holds tmp row `row`: meta[b][8] = first tmp row of image b
    while (b + 1 < B && meta[(b + 1) * 9 + 8] <= row) ++b;
    const long long* m = meta + b * 9;
    const int Ws = (int)m[1], ks = (int)m[2];
    const int y = (int)(row - m[8]);
    const int* bounds = coef + m[4];
    const int* k = coef + m[5] + (long long)xx * ks;
    const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
    const unsigned char* p = src[b] + ((long long)y * Ws + xmin) * 3;
    int s0 = 1 << (PRE_PREC - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xmax; ++x) { const int w = k[x]; s0 += p[3 * x] * w; s1 += p[3 * x + 1] * w; s2 += p[3 * x + 2] * w; }
    unsigned char* o = tmp + (row * Wo + xx) * 3;
    o[0] = (unsigned char)clip8i(s0); o[1] = (unsigned char)clip8i(s1); o[2] = (unsigned char)clip8i(s2);
  }
}

__global__ __launch_bounds__(256) void bicubic_v_kernel(const unsigned char* __restrict__ tmp, const long long* __restrict__ meta,
                                                        const int* __restrict__ coef, bf16_t* __restrict__ dst, unsigned char* __restrict__ u8_out,
                                                        int B, int Ho, int Wo, float rescale, float m0, float m1, float m2, float sd0,
                                                        float sd1, float sd2) {
  const long long total = (long long)B * Ho * Wo;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int xx = i % Wo, yy = (i / Wo) % Ho, b = i / ((long long)Wo * Ho);
    const long long* m = meta + b * 9;
    const int ks = (int)m[3];
    const int* bounds = coef + m[6];
    const int* k = coef + m[7] + (long long)yy * ks;
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const unsigned char* p = tmp + ((m[8] + ymin) * Wo + xx) * 3;
    int s0 = 1 << (PRE_PREC - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < ymax; ++y) { const int w = k[y]; const unsigned char* q = p + (long long)y * Wo * 3; s0 += q[0] * w; s1 += q[1] * w; s2 += q[2] * w; }
    const int v[3] = {clip8i(s0), clip8i(s1), clip8i(s2)};
    const float mean[3] = {m0, m1, m2}, sd[3] = {sd0, sd1, sd2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (u8_out) u8_out[(((long long)b * Ho + yy) * Wo + xx) * 3 + c] = (unsigned char)v[c];
      dst[(((long long)b * 3 + c) * Ho + yy) * Wo + xx] = f2bf(((float)v[c] * rescale - mean[c]) / sd[c]);      // HF: rescale, then (x - mean) / std
    }
  }
}

extern "C" int medmoe_preprocess_bicubic(const void* const* src_ptrs, const long long* meta, const int* coef, void* tmp, void* dst,
                                         void* u8_out, int B, int Ho, int Wo, long long total_src_rows, float rescale,
                                         const float* mean3, const float* std3, hipStream_t stream) {
  if (!src_ptrs || !meta || !coef || !tmp || !dst || !mean3 || !std3) return MM_ERR_ARG;
  if (B <= 0 || Ho <= 0 || Wo <= 0 || total_src_rows <= 0) return MM_ERR_SHAPE;
  for (int c = 0; c < 3; ++c) if (!(std3[c] > 0.f)) return MM_ERR_ARG;
  const long long n1 = total_src_rows * Wo, n2 = (long long)B * Ho * Wo;
  hipLaunchKernelGGL(bicubic_h_kernel, dim3((int)min((n1 + 255) / 256, (long long)256 * 16)), dim3(256), 0, stream,
                     (const unsigned char* const*)src_ptrs, meta, coef, (unsigned char*)tmp, B, Wo, total_src_rows);
  hipLaunchKernelGGL(bicubic_v_kernel, dim3((int)min((n2 + 255) / 256, (long long)256 * 16)), dim3(256), 0, stream,
                     (const unsigned char*)tmp, meta, coef, (bf16_t*)dst, (unsigned char*)u8_out, B, Ho, Wo, rescale, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2]);
  return mm_check_launch();
}
