// Hardware probe: random-gather ceilings for the access shapes the mapper uses.
//  (a) one 8-byte load per lane from a large table (counter / index probes)
//  (b) nine consecutive 8-byte words per lane at a random 8-byte-aligned place
//      (a candidate's genome window for the Hamming filter)
// Reports G loads/s and GB/s of useful bytes at 1.5 GB footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
template <int WORDS>
__global__ __launch_bounds__(64) void gather(const uint64_t *__restrict__ tab, uint64_t nwords, int iters, uint64_t *out) {
  uint64_t acc = 0, s = mix(blockIdx.x * 64ull + threadIdx.x + 1);
  for (int it = 0; it < iters; ++it) {
    s = mix(s + it);
    const uint64_t at = s % (nwords - WORDS);
#pragma unroll
    for (int w = 0; w < WORDS; ++w) acc += tab[at + w];
  }
  if (acc == 0x1234567) out[0] = acc;
}
// same window as 16-byte loads (8-byte aligned only): 5 x dwordx4 instead of 9 x dwordx2
__global__ __launch_bounds__(64) void gather_x4(const uint64_t *__restrict__ tab, uint64_t nwords, int iters, uint64_t *out) {
  uint64_t acc = 0, s = mix(blockIdx.x * 64ull + threadIdx.x + 1);
  for (int it = 0; it < iters; ++it) {
    s = mix(s + it);
    const uint64_t at = s % (nwords - 10);
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(tab + at);
#pragma unroll
    for (int w = 0; w < 5; ++w) { const ulonglong2 v = p[w]; acc += v.x + v.y; }
  }
  if (acc == 0x1234567) out[0] = acc;
}
template <int WORDS> void run(const uint64_t *d, uint64_t nwords, int blocks, int iters, uint64_t *dout, const char *label) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(gather<WORDS>, dim3(blocks), dim3(64), 0, 0, d, nwords, iters / 4, dout);
  hipEventRecord(a);
  hipLaunchKernelGGL(gather<WORDS>, dim3(blocks), dim3(64), 0, 0, d, nwords, iters, dout);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = double(blocks) * 64 * iters;
  printf("%-28s blocks=%6d  %7.2f ms  %6.2f G gathers/s  %7.1f GB/s useful\n", label, blocks, ms, n / ms / 1e6, n * WORDS * 8 / ms / 1e6);
}
int main() {
  const uint64_t nwords = 1536ull << 17;  // 1.5 GB
  uint64_t *d, *dout; hipMalloc(&d, nwords * 8); hipMalloc(&dout, 8); hipMemset(d, 1, nwords * 8);
  for (int blocks : {256 * 8, 256 * 16, 256 * 32}) {
    run<1>(d, nwords, blocks, 2000, dout, "8B random (1.5GB)");
    run<9>(d, nwords, blocks, 400, dout, "9x8B window random (1.5GB)");
  }
  {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 8192, iters = 400;
    hipLaunchKernelGGL(gather_x4, dim3(blocks), dim3(64), 0, 0, d, nwords, iters / 4, dout);
    hipEventRecord(a);
    hipLaunchKernelGGL(gather_x4, dim3(blocks), dim3(64), 0, 0, d, nwords, iters, dout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("5x16B window random (1.5GB)  blocks=%6d  %7.2f ms  %6.2f G windows/s\n", blocks, ms, double(blocks) * 64 * iters / ms / 1e6);
  }
  run<1>(d, (512ull << 20) / 8, 256 * 32, 2000, dout, "8B random (512MB)");
  run<1>(d, (128ull << 20) / 8, 256 * 32, 2000, dout, "8B random (128MB)");
  return 0;
}
