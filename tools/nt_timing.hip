// Phase timing of gemm_nt256_kernel: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNT_TIMING -Iinclude -Imedmoe_amd/csrc
//   tools/nt_timing.hip -o tools/nt_timing.bin ; ./tools/nt_timing.bin [M N K]
// Prints, per wave and averaged over blocks, the shader clocks spent in the counted vmcnt waits, at the segment
// barriers, in the LOAD segments (LDS fragment reads + DMA issue), the MFMA segments and the epilogues.
#include "../medmoe_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
  int M = argc > 3 ? atoi(argv[1]) : 50432, N = argc > 3 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  void *A, *B, *C, *X = nullptr; float* bias = nullptr;
  const int epi = getenv("NT_EPI") ? atoi(getenv("NT_EPI")) : 0;     // 1: bias + GELU + aux store, 3: x GELU'(aux), 5: bias + residual
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
  hipMemset(A, 0, (size_t)M * K * 2); hipMemset(B, 0, (size_t)N * K * 2);
  if (getenv("NT_RAND")) {      // random bf16 in (-2, 2): zeros run at a higher clock (less switching power) than real data
    auto fill = [](void* d, size_t n) {
      unsigned short* h = (unsigned short*)malloc(n * 2);
      unsigned x = 12345u;
      for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00 | ((x >> 9) & 0x7f)); }
      hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice); free(h);
    };
    fill(A, (size_t)M * K); fill(B, (size_t)N * K);
  }
  if (epi) { hipMalloc(&X, (size_t)M * N * 2); hipMemset(X, 0, (size_t)M * N * 2); hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4); }
  auto run = [&]() {
    if (epi == 1) return medmoe_gemm_nt(A, K, B, K, C, N, M, N, K, bias, nullptr, 0, X, N, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 1.f, 1, 0, 0, 0);
    if (epi == 3) return medmoe_gemm_nt(A, K, B, K, C, N, M, N, K, nullptr, nullptr, 0, X, N, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 1.f, 3, 0, 0, 0);
    if (epi == 5) return medmoe_gemm_nt(A, K, B, K, C, N, M, N, K, bias, X, N, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 1.f, 0, 0, 0, 0);
    return medmoe_gemm_nt(A, K, B, K, C, N, M, N, K, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 1.f, 0, 0, 0, 0);
  };
#ifdef NT_EXPERIMENT
  if (getenv("NT_SKIP")) { int np = atoi(getenv("NT_SKIP")); hipMemcpyToSymbol(HIP_SYMBOL(g_nt_dbg_skip), &np, sizeof np); }
#endif
  if (getenv("NT_GRID")) medmoe_set_option(5, atoi(getenv("NT_GRID")));
  if (getenv("NT_4W")) medmoe_set_option(7, atoi(getenv("NT_4W")));
  for (int i = 0; i < 3; ++i) if (run()) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  unsigned long long z[64] = {0};
#ifdef NT_TIMING
  hipMemcpyToSymbol(HIP_SYMBOL(g_nt_timing), z, sizeof z);
#endif
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 10;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) run();
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
#ifdef NT_TIMING
  hipMemcpyFromSymbol(z, HIP_SYMBOL(g_nt_timing), sizeof z);
#endif
  const double nb = (double)z[6];
  printf("%dx%dx%d: %.3f ms %.0f TF/s; blocks*reps %.0f\n", M, N, K, ms, 2.0 * M * N * K / ms / 1e9, nb);
#ifdef NT_TIMING
  const char* nm[6] = {"total", "vmcnt wait", "barrier wait", "load segment", "mfma segment", "epilogue"};
  for (int i = 0; i < 6; ++i) {
    printf("  %-20s", nm[i]);
    for (int w = 0; w < 8; ++w) printf(" %8.0f", z[w * 8 + i] / nb);
    printf("  clk/block per wave 0..7\n");
  }
#endif
  return 0;
}
