// Hardware probe: what a random 32-byte gather costs the memory system by load flavour and allocation type.
// Each lane reads 32 bytes (two 16-byte loads) at a random 32-byte-aligned place of a 4 GB table, four gathers in
// flight per lane.  Variants: plain loads, nt, sc1, sc0 sc1 (inline asm), and plain loads from an uncached and from a
// fine-grained allocation.  Prints G gathers/s; run under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`
// to see how many bytes a gather moves (requests x 64 B, or 32 B for the _32B kind).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// four 32-byte gathers of a lane in flight, then one wait -- all inside one asm block, so that no register the loads
// are still to write can be handed to anything else
#define GATHER4(FL)                                                                                                 \
  asm volatile("global_load_dwordx4 %0, %8, off " FL "\n\tglobal_load_dwordx4 %1, %8, off offset:16 " FL "\n\t"      \
               "global_load_dwordx4 %2, %9, off " FL "\n\tglobal_load_dwordx4 %3, %9, off offset:16 " FL "\n\t"      \
               "global_load_dwordx4 %4, %10, off " FL "\n\tglobal_load_dwordx4 %5, %10, off offset:16 " FL "\n\t"    \
               "global_load_dwordx4 %6, %11, off " FL "\n\tglobal_load_dwordx4 %7, %11, off offset:16 " FL "\n\t"    \
               "s_waitcnt vmcnt(0)"                                                                                 \
               : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])  \
               : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory")

template <int F>
__global__ __launch_bounds__(64) void gather32(const char *__restrict__ tab, uint64_t n32, int iters, unsigned *out) {
  unsigned acc = 0;
  uint64_t s = mix(blockIdx.x * 64ull + threadIdx.x + 1);
  for (int it = 0; it < iters; ++it) {
    v4u x[8];
    const char *p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s = mix(s + it * 4 + k);
      p[k] = tab + (s % n32) * 32;
    }
    if (F == 0) GATHER4("");
    else if (F == 1) GATHER4("nt");
    else if (F == 2) GATHER4("sc1");
    else GATHER4("sc0 sc1");
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += x[k].x ^ x[k].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int F> void run(const char *d, uint64_t bytes, const char *label, unsigned *dout) {
  const int blocks = 256 * 20, iters = 300;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(gather32<F>, dim3(blocks), dim3(64), 0, 0, d, bytes / 32, iters / 4, dout);
  hipEventRecord(a);
  hipLaunchKernelGGL(gather32<F>, dim3(blocks), dim3(64), 0, 0, d, bytes / 32, iters, dout);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = double(blocks) * 64 * iters * 4;
  const hipError_t e = hipGetLastError();
  printf("%-44s %8.2f ms  %6.2f G gathers/s  %7.1f GB/s useful (32 B each)%s\n", label, ms, n / ms / 1e6, n * 32 / ms / 1e6,
         e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main() {
  const uint64_t bytes = 4ull << 30;
  char *d = nullptr, *du = nullptr, *df = nullptr;
  unsigned *dout; hipMalloc(&dout, 64);
  if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 1, bytes);
  run<0>(d, bytes, "hipMalloc, plain loads", dout);
  run<1>(d, bytes, "hipMalloc, nt loads", dout);
  run<2>(d, bytes, "hipMalloc, sc1 loads", dout);
  run<3>(d, bytes, "hipMalloc, sc0 sc1 loads", dout);
  if (hipExtMallocWithFlags(reinterpret_cast<void **>(&du), bytes, hipDeviceMallocUncached) == hipSuccess) {
    hipMemset(du, 1, bytes);
    run<0>(du, bytes, "uncached allocation, plain loads", dout);
    run<1>(du, bytes, "uncached allocation, nt loads", dout);
    hipFree(du);
  } else printf("uncached allocation not available\n");
  if (hipExtMallocWithFlags(reinterpret_cast<void **>(&df), bytes, hipDeviceMallocFinegrained) == hipSuccess) {
    hipMemset(df, 1, bytes);
    run<0>(df, bytes, "fine-grained allocation, plain loads", dout);
    hipFree(df);
  } else printf("fine-grained allocation not available\n");
  // the same gathers from a table the Infinity Cache holds (192 MB)
  run<0>(d, 192ull << 20, "hipMalloc, plain loads, 192 MB table", dout);
  run<1>(d, 192ull << 20, "hipMalloc, nt loads, 192 MB table", dout);
  // ... and from tables the L2s hold (4 MB per XCD, every XCD caching its own copy of whatever its CUs ask for): what a
  // region-binned filter pass would gather its windows from (round 5)
  for (uint64_t mb : {1ull, 2ull, 4ull, 8ull, 16ull, 64ull}) {
    char label[96];
    snprintf(label, sizeof(label), "hipMalloc, plain loads, %llu MB table", static_cast<unsigned long long>(mb));
    run<0>(d, mb << 20, label, dout);
  }
  return 0;
}
