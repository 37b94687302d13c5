// Phase timing of local_pair2_kernel: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPAIR_TIMING -Iinclude -Imedmoe_amd/csrc
//   tools/pair_timing.hip -o tools/pair_timing.bin ; ./tools/pair_timing.bin [ntt]
#include "../medmoe_amd/csrc/loss.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
// gemm.hip is not linked into this tool: the score GEMM is not called here
bool mm_launch_scores512(const void*, const void*, const int*, void*, float*, int, int, int, int, int, const int*, int, int, long long, long long, hipStream_t) { return false; }
int main(int argc, char** argv) {
  const int ntt = argc > 1 ? atoi(argv[1]) : 3;
  const int B = 256, HW = 196, HWP = 208, T = 77, GW = 224, TP = 16 * ntt;
  const long long ldp = (long long)B * TP;
  void *a1, *dS, *U, *gmp; float *lse, *wn, *sim; int *caps, *list;
  hipMalloc(&a1, B * HWP * ldp * 2); hipMalloc(&dS, B * HWP * ldp * 2); hipMalloc(&U, B * HWP * ldp * 2);
  hipMalloc(&gmp, (size_t)B * HWP * GW * 2); hipMalloc(&lse, (size_t)B * HWP * B * 4); hipMalloc(&wn, B * T * 4); hipMalloc(&sim, B * B * 4);
  hipMalloc(&caps, B * 4); hipMalloc(&list, B * 4);
  std::vector<unsigned short> h((size_t)B * HWP * ldp, 0x3c00);       // bf16 0.0078: a plausible softmax value
  hipMemcpy(a1, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemset(gmp, 0, (size_t)B * HWP * GW * 2); hipMemset(lse, 0, (size_t)B * HWP * B * 4);
  std::vector<float> w(B * T, 1.f); hipMemcpy(wn, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  std::vector<int> c(B, TP - 3), l(B); for (int i = 0; i < B; ++i) l[i] = i;
  hipMemcpy(caps, c.data(), B * 4, hipMemcpyHostToDevice); hipMemcpy(list, l.data(), B * 4, hipMemcpyHostToDevice);
  auto run = [&]() { return medmoe_local_pair2_ragged(a1, lse, gmp, wn, caps, nullptr, sim, dS, U, B, B, HW, T, 4.f, 5.f, 1e-8f, list, B, ntt, 0, ldp, 0); };
  if (run()) { printf("launch failed\n"); return 1; }
  hipMemcpy(a1, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  unsigned long long z[16] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_pair_timing), z, sizeof z);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0); run(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpyFromSymbol(z, HIP_SYMBOL(g_pair_timing), sizeof z);
  printf("ntt %d: %.3f ms for %d pairs = %.1f ns per pair\n", ntt, ms, B * B, ms * 1e6 / (B * B));
  const char* nm[8] = {"phase 0 (tile -> LDS, zero image)", "phase 1 (exp column sums)", "phase 2 (A, image, num)", "GEMM2 pass 1 (n2)",
                       "reduce + scalar (cos, sim, d*)", "phase 3 (GEMM2 pass 2 + dS)", "copy-out dS, A, U", ""};
  double tot = 0; for (int i = 0; i < 7; ++i) tot += z[i];
  for (int i = 0; i < 7; ++i) printf("  %-36s %8.0f clk per pair  %5.1f %%\n", nm[i], (double)z[i] / z[15], 100.0 * z[i] / tot);
  return 0;
}
