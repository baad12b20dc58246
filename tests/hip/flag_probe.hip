// Probe: when does a host thread see a flag that a still-running kernel stored to pinned host memory?
// (the single-end host entry point releases its kernel turn on such a flag)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void k(unsigned *flag, long long spin) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}
int main() {
  for (unsigned flags : {hipHostMallocMapped, hipHostMallocMapped | hipHostMallocCoherent, hipHostMallocDefault}) {
    unsigned *f = nullptr;
    if (hipHostMalloc(reinterpret_cast<void **>(&f), 4, flags) != hipSuccess) { printf("alloc failed\n"); continue; }
    *f = 0;
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, st, f, 400000000ll);
    double seen = -1;
    while (hipStreamQuery(st) == hipErrorNotReady) {
      if (seen < 0 && __atomic_load_n(f, __ATOMIC_RELAXED)) seen = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    const double done = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("flags %u: flag seen at %.2f ms, kernel done at %.2f ms\n", flags, seen, done);
    hipStreamDestroy(st);
    hipHostFree(f);
  }
  return 0;
}
