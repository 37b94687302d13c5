// Cost of one workgroup s_barrier on gfx950 with nothing between barriers:
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_cost.hip -o tools/barrier_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out, int iters) {
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main() {
  long long* d; hipMalloc(&d, 8);
  for (int threads : {64, 128, 256, 512, 1024}) {
    const int iters = 10000;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%4d threads/block, 256 blocks: %.1f clk per s_barrier\n", threads, (double)h / iters);
  }
  return 0;
}
