// Hardware probe: cost of one candidate-set update (SeSet::admit) in isolation.
#include "../../abismal_amd/csrc/abm_kernels.hip"
#include <cstdio>
using namespace abm;
__global__ __launch_bounds__(64) void k(int n_updates, long long *cycles, int *sink) {
  SeSet S;
  S.begin_read(100);
  S.cutoff = 40;
  unsigned x = 12345u + blockIdx.x;
  const long long t0 = clock64();
  for (int i = 0; i < n_updates; ++i) {
    x = x * 1664525u + 1013904223u;
    const int d = 1 + (x >> 8) % 12;            // uniform across the wave
    if (d <= S.cutoff) S.admit(true, d, 0, 1000u + i);
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) { cycles[blockIdx.x] = t1 - t0; sink[blockIdx.x] = S.hk + S.sz + S.cutoff; }
}
int main() {
  long long *dc; int *ds; hipMalloc(&dc, 8 * 65536); hipMalloc(&ds, 4 * 65536);
  for (int blocks : {256, 1024, 4096, 8192}) {
    const int n = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, n, dc, ds);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, n, dc, ds);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c0; hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost);
    printf("blocks=%5d  %8.3f ms  wave0: %6.1f cycles/update  chip: %7.1f M updates/s\n", blocks, ms, double(c0) / n, double(blocks) * n / ms / 1e3);
  }
  return 0;
}
