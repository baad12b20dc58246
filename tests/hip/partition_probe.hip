// Hardware probe (round 5): at what rate can waves PARTITION a stream of small records into region
// bins -- what a batch-wide, region-binned candidate filter would have to do with its 28 G (read, offset, position)
// records per 10 M reads before it could gather their windows from L2-resident plane lines.
// Every workgroup (256 lanes) stages records in LDS, one 256-byte buffer per bin (32 records of 8 bytes), and flushes a
// bin as soon as it holds a full 128-byte line: 8 lanes write 16 bytes each, a whole line per request, to a place taken
// from the bin's global cursor.  Bins: 256 (one LDS buffer set = 64 KB + counters).  Records: random 8-byte values whose
// low byte names the bin.  Prints records/s and GB/s of records written; the total written is checked against the
// total produced.  Variants: 8-byte and (two words of a 16-byte pair: 12 useful) 16-byte records.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

constexpr int kBins = 256, kLine = 16;  // records per flushed line

// one round: every lane offers R records; afterwards the lanes flush the bins that hold a line (two flush passes a round)
template <int R, int kSlots>
__global__ __launch_bounds__(256) void partition8(uint64_t *__restrict__ out, unsigned long long *__restrict__ cursor,
                                                  uint64_t cap_per_bin, int rounds, unsigned long long *__restrict__ total) {
  extern __shared__ uint64_t smem[];
  uint64_t (*buf)[kSlots] = reinterpret_cast<uint64_t (*)[kSlots]>(smem);
  unsigned *cnt = reinterpret_cast<unsigned *>(smem + kBins * kSlots);
  unsigned *full_list = cnt + kBins;
  unsigned *n_full = full_list + kBins;
  const int t = threadIdx.x;
  cnt[t] = 0;
  if (t == 0) *n_full = 0;
  __syncthreads();
  uint64_t s = mix(blockIdx.x * 256ull + t + 1);
  unsigned long long written = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      s = mix(s + r * R + k);
      const unsigned bin = static_cast<unsigned>(s) & (kBins - 1);
      const unsigned at = atomicAdd(&cnt[bin], 1u);
      if (at < kSlots) buf[bin][at] = s;  // (overflowing a buffer loses the record: counted by the check at the end)
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
      const unsigned c = min(cnt[t], static_cast<unsigned>(kSlots));
      if (c >= kLine) full_list[atomicAdd(n_full, 1u)] = t;
      __syncthreads();
      const unsigned nf = *n_full;
      // 8 lanes per full bin, 32 bins per step: lane j of the group writes records 2 j, 2 j + 1 (16 bytes)
      for (unsigned base = 0; base < nf; base += 32) {
        const unsigned k = base + (t >> 3), j = t & 7;
        unsigned long long g = 0;
        unsigned b = 0;
        if (k < nf) {
          b = full_list[k];
          if (j == 0) g = atomicAdd(&cursor[b], static_cast<unsigned long long>(kLine));
        }
        g = __shfl(g, (t & 63) & ~7);
        if (k < nf && g + kLine <= cap_per_bin) {
          typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
          const u64x2 v = {buf[b][2 * j], buf[b][2 * j + 1]};
          *reinterpret_cast<u64x2 *>(out + b * cap_per_bin + g + 2 * j) = v;
          if (j == 0) written += kLine;
        }
      }
      __syncthreads();
      if (c >= kLine) {  // what a flushed bin still holds moves to the front
        const unsigned rest = c - kLine;
        for (unsigned i = 0; i < rest; ++i) buf[t][i] = buf[t][kLine + i];
        cnt[t] = rest;
      }
      if (t == 0) *n_full = 0;
      __syncthreads();
    }
  }
  // (the last partial lines stay in LDS: a real pass would flush them; they are a rounding error of the rate)
  if (written) atomicAdd(total, written);
}

template <int R, int kSlots> void run(int blocks, int rounds) {
  const uint64_t per_bin = (static_cast<uint64_t>(blocks) * 256 * rounds * R / kBins) * 5 / 4 + 4096;
  uint64_t *out = nullptr;
  unsigned long long *cursor = nullptr, *total = nullptr;
  if (hipMalloc(&out, per_bin * kBins * 8) != hipSuccess) { printf("alloc failed\n"); return; }
  (void)hipMalloc(&cursor, kBins * 8); (void)hipMalloc(&total, 8);
  const size_t lds = static_cast<size_t>(kBins) * kSlots * 8 + kBins * 4 * 2 + 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(partition8<R, kSlots>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(cursor, 0, kBins * 8); (void)hipMemset(total, 0, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((partition8<R, kSlots>), dim3(blocks), dim3(256), lds, 0, out, cursor, per_bin, rounds, total);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    unsigned long long w = 0;
    (void)hipMemcpy(&w, total, 8, hipMemcpyDeviceToHost);
    const double produced = double(blocks) * 256 * rounds * R;
    const hipError_t e = hipGetLastError();
    printf("partition into %d bins, 8-byte records, %d per lane and round, %zu KB LDS per workgroup, 128-byte flushes: %8.2f ms  %6.2f G records/s  %7.1f GB/s written  (%.4f of the records produced)%s\n",
           kBins, R, lds >> 10, ms, w / ms / 1e6, w * 8.0 / ms / 1e6, w / produced, e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
  }
  (void)hipFree(out); (void)hipFree(cursor); (void)hipFree(total);
}

int main() {
  run<1, 32>(256 * 8, 4000);
  run<2, 40>(256 * 8, 2000);
  run<4, 56>(256 * 4, 2000);
  return 0;
}
