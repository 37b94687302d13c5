// Phase clocks of attn_fwd_kernel (wave 0 of every workgroup): hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_TIMING
//   -Iinclude -Imedmoe_amd/csrc tools/attn_timing.hip -o tools/attn_timing.bin ; ./tools/attn_timing.bin
#include "../medmoe_amd/csrc/attention.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main() {
  const int B = 1024, N = 197, H = 12, D = H * 64;
  void *qkv, *out; float* lse;
  hipMalloc(&qkv, (size_t)B * N * 3 * D * 2); hipMalloc(&out, (size_t)B * N * D * 2); hipMalloc(&lse, (size_t)B * H * N * 4);
  std::vector<unsigned short> h((size_t)B * N * 3 * D);
  unsigned x = 1u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x80ff) | 0x3e00 | ((x >> 9) & 0x7f)); }
  hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int i = 0; i < 3; ++i) medmoe_attn_fwd(qkv, out, lse, nullptr, B, N, H, 64, 0);
  hipDeviceSynchronize();
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_attn_timing), z, sizeof z);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0); medmoe_attn_fwd(qkv, out, lse, nullptr, B, N, H, 64, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpyFromSymbol(z, HIP_SYMBOL(g_attn_timing), sizeof z);
  const double n = (double)z[5];
  printf("attn_fwd B %d N %d H %d: %.3f ms; per workgroup (wave 0, clock64 ticks): fill %.0f  scores %.0f  softmax %.0f  PV+store %.0f  total %.0f\n",
         B, N, H, ms, z[0] / n, z[1] / n, z[2] / n, z[3] / n, z[4] / n);
  return 0;
}
