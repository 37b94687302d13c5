// Microbenchmark: how fast can a CU drain C-tile shaped stores?  (round-1 GEMM epilogue analysis)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void tile_store(uint4* C, int ldc_bytes, int n_tiles_n, int tiles_per_block, int mode) {
  // block writes tiles of 256 rows x 256 B (128 bf16 columns); mode 0: 16 B per lane, 16 lanes per row (full 256-B row
  // segments per 16 lanes), mode 1: GEMM-epilogue shape (each store instruction = 16 rows x 64 B)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint4 v = make_uint4(tid, 1, 2, 3);
  for (int t = 0; t < tiles_per_block; ++t) {
    const long long id = (long long)t * gridDim.x + blockIdx.x;
    const long long tile_m = id / n_tiles_n, tile_n = id % n_tiles_n;
    char* base = (char*)C + tile_m * 256 * (long long)ldc_bytes + tile_n * 256;
    if (mode == 0) {
      // wave w covers rows w*32..w*32+31; instruction i: 4 rows x 256 B
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wid * 32 + i * 4 + (lane >> 4), c = lane & 15;
        *(uint4*)(base + (long long)row * ldc_bytes + c * 16) = v;
      }
    } else if (mode == 2) {
      // wave (wm = w>>1, wn = w&1): 64 rows x 128 B; instruction i: 8 rows x 128 B (one full line per 8 lanes)
      const int wm = wid >> 1, wn = wid & 1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wm * 64 + i * 8 + (lane >> 3), c = lane & 7;
        *(uint4*)(base + (long long)row * ldc_bytes + wn * 128 + c * 16) = v;
      }
    } else {
      // wave (wm = w>>1, wn = w&1): 64 rows x 128 B; instruction (tm, j): 16 rows x 64 B
      const int wm = wid >> 1, wn = wid & 1;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = wm * 64 + tm * 16 + (lane & 15), c = (lane >> 4);
          *(uint4*)(base + (long long)row * ldc_bytes + wn * 128 + j * 64 + c * 16) = v;
        }
    }
  }
}
int main() {
  const long long M = 50432, N = 3072;
  uint4* C; hipMalloc(&C, M * N * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n_tiles_n = N / 128, total = (M / 256) * n_tiles_n;
  for (int blocks : {16, 64, 256, 512})
    for (int mode = 0; mode < 3; ++mode) {
      const int tpb = blocks >= 256 ? total / blocks : total / 256;      // few blocks: the same work per block as at 256 (per-CU rate)
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(tile_store, dim3(blocks), dim3(512), 0, 0, C, (int)(N * 2), n_tiles_n, tpb, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("blocks %d mode %d: %.3f ms per pass, %.2f TB/s, %.1f GB/s per CU\n", blocks, mode, ms / 10,
                        (double)tpb * blocks * 65536 / (ms / 10 * 1e-3) / 1e12, (double)tpb * blocks * 65536 / (ms / 10 * 1e-3) / 1e9 / 256);
      }
    }
  return 0;
}
