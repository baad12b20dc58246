// Hardware probe: cost of the wavefront DP (three 21-wide bands per wave) in isolation.
#include "../../abismal_amd/csrc/abm_kernels.hip"
#include <cstdio>
using namespace abm;
__global__ __launch_bounds__(64) void k(int reps, long long *cycles, int *sink) {
  extern __shared__ __align__(16) unsigned char smem[];
  WaveLds lds;
  lds.W = 7; lds.WB = 3; lds.GW = 10;
  lds.qpk = reinterpret_cast<u64 *>(smem);
  lds.gwin = lds.qpk + 28;
  lds.tb = reinterpret_cast<u8 *>(lds.gwin + 21 * 10);
  const int lane = threadIdx.x;
  for (int i = lane; i < 28 + 210; i += 64) lds.qpk[i] = 0x1248124812481248ull * (i + 1);
  __syncthreads();
  AlnJob job = {0, 0, 0, 0, 0};
  if (lane < 63) { job.bw = 21; job.jl = lane % 21; job.g = lane / 21; job.t0nib = 3; }
  int acc = 0;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    int bv, br;
    wavefront<false>(lds, job, 100, 21, 21, bv, br);
    acc += bv + br;
  }
  const long long t1 = clock64();
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + lane] = acc;
}
int main() {
  long long *dc; int *ds; hipMalloc(&dc, 8 * 65536); hipMalloc(&ds, 4 * 64 * 65536);
  for (int blocks : {256, 1024, 4096, 8192}) {
    const int reps = 50;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 8192, 0, reps, dc, ds);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 8192, 0, reps, dc, ds);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c0; hipMemcpy(&c0, dc, 8, hipMemcpyDeviceToHost);
    const double steps = 2.0 * (100 - 1 + 21) - 20 + 1;
    printf("blocks=%5d %8.3f ms  wave0: %8.0f cycles/round = %5.1f cycles/step  chip: %6.2f M rounds/s (= %6.2f M alignments/s)\n",
           blocks, ms, double(c0) / reps, double(c0) / reps / steps, double(blocks) * reps / ms / 1e3, 3.0 * blocks * reps / ms / 1e3);
  }
  return 0;
}
