// GPU test program (built and run by tests/test_gpu_se_set.py): the paired-end candidate set in memory
// (PeSet: append / heapify / push / pop_max / tie_run) against libstdc++'s heap calls driven as
// pe_candidates::update drives them (src/abismal.cpp:824-842), through growth to 32768 entries and eviction beyond.
// Prints "OK ..." or the first mismatch.
#include "../../abismal_amd/csrc/abm_pe_set.hpp"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace abm;

struct El { short d; unsigned pos; };
static bool by_d(const El &a, const El &b) { return a.d < b.d; }

__device__ __host__ inline int draw(unsigned seq, unsigned idx, int shape) {
  unsigned x = (seq * 2654435761u) ^ (idx * 40503u + 977u);
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
  const unsigned r = x >> 4;
  switch (shape) {
    case 0: return 1 + r % 12;                          // a third beyond good_cutoff = 10
    case 1: return (r % 16) ? 3 : 1 + r % 3;            // mostly one value
    case 2: return (r % 300) ? 2 : 1;                   // long runs of ties, a few better hits
    default: return 1 + r % 4;
  }
}
constexpr int kShapes = 4;

__global__ __launch_bounds__(64) void run(int n_chunks, u32 *heaps, u32 *poss, i16 *lds_, int *meta, long long *cycles) {
  const long long t_begin = clock64();
  int n_admits = 0;
  PeSet S;
  S.heap = heaps + static_cast<size_t>(blockIdx.x) * kPeCapLarge;
  S.lpos = poss + static_cast<size_t>(blockIdx.x) * kPeCapLarge;
  S.ld = lds_ + static_cast<size_t>(blockIdx.x) * kPeCapLarge;
  S.cap_avail = kPeCapLarge;
  S.spill_pos = nullptr; S.spill_d = nullptr; S.spill_cap = 0; S.spilled = false;
  S.begin_read(100);
  S.cutoff = S.good_cutoff;  // set_specific
  const int shape = blockIdx.x % kShapes, lane = threadIdx.x;
  int n_runs = 0;
  for (int i = 0; i < n_chunks; ++i) {
    const unsigned idx = i * 64 + lane;
    const int h = draw(blockIdx.x, idx, shape);
    const unsigned p = 1000u + idx;
    u64 todo = __ballot(h <= S.cutoff);
    if (!S.heaped && todo) todo = S.append(todo, h, p);
    while (todo && !S.sure_ambig) {
      if (S.tie_run(todo, __ballot(h == S.cutoff), p, 0u)) { ++n_runs; continue; }
      const int l = __builtin_ctzll(todo);
      const int before = S.cutoff;
      S.admit(true, rdlane(h, l), 0u, rdlane(p, l));
      ++n_admits;
      todo &= ~(((1ull << l) << 1) - 1);
      if (S.cutoff < before) todo &= __ballot(h <= S.cutoff);
    }
  }
  wave_sync();
  if (!S.heaped) S.heapify();
  if (threadIdx.x == 0) { cycles[2 * blockIdx.x] = clock64() - t_begin; cycles[2 * blockIdx.x + 1] = n_admits; }
  if (threadIdx.x == 0) { meta[4 * blockIdx.x] = S.sz; meta[4 * blockIdx.x + 1] = S.cutoff; meta[4 * blockIdx.x + 2] = n_runs; meta[4 * blockIdx.x + 3] = S.capacity; }
}

int main() {
  const int blocks = 12, chunks = 1100, n = chunks * 64;
  u32 *dh, *dp; i16 *dl; int *dm; long long *dc;
  (void)hipMalloc(&dc, blocks * 16);
  (void)hipMalloc(&dh, sizeof(u32) * blocks * kPeCapLarge); (void)hipMalloc(&dp, sizeof(u32) * blocks * kPeCapLarge);
  (void)hipMalloc(&dl, sizeof(i16) * blocks * kPeCapLarge); (void)hipMalloc(&dm, blocks * 16);
  hipLaunchKernelGGL(run, dim3(blocks), dim3(64), 0, 0, chunks, dh, dp, dl, dm, dc);
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAIL launch\n"); return 1; }
  std::vector<u32> hh(static_cast<size_t>(blocks) * kPeCapLarge), hp(hh.size()); std::vector<int> hm(blocks * 4);
  (void)hipMemcpy(hh.data(), dh, hh.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hp.data(), dp, hp.size() * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost);
  long long runs = 0;
  for (int b = 0; b < blocks; ++b) {
    std::vector<El> v(kPeCapLarge);
    v[0] = El{static_cast<short>(0.4 * 100), 0u};
    int sz = 1, capacity = kPeCapSmall, cutoff = 10;
    const int good = 10;
    for (int i = 0; i < n; ++i) {
      const int d = draw(b, i, b % kShapes);
      if (d > cutoff) continue;
      if (sz == capacity) {
        if (capacity != static_cast<int>(kPeCapLarge) && d <= good) ++capacity;
        else { std::pop_heap(v.begin(), v.begin() + sz, by_d); --sz; }
      }
      v[sz++] = El{static_cast<short>(d), 1000u + i};
      std::push_heap(v.begin(), v.begin() + sz, by_d);
      cutoff = std::min<int>(cutoff, v[0].d);
    }
    runs += hm[4 * b + 2];
    if (hm[4 * b] != sz || hm[4 * b + 1] != cutoff || hm[4 * b + 3] != capacity) {
      printf("FAIL seq %d: size/cutoff/capacity %d/%d/%d vs %d/%d/%d\n", b, hm[4 * b], hm[4 * b + 1], hm[4 * b + 3], sz, cutoff, capacity);
      return 1;
    }
    for (int k = 0; k < sz; ++k) {
      const u32 key = hh[static_cast<size_t>(b) * kPeCapLarge + k], handle = key & 0x7FFFu;
      if (static_cast<int>(key >> 16) != v[k].d || hp[static_cast<size_t>(b) * kPeCapLarge + handle] != v[k].pos) {
        printf("FAIL seq %d (shape %d) heap[%d]: gpu d %u pos %u, libstdc++ d %d pos %u\n", b, b % kShapes, k, key >> 16,
               hp[static_cast<size_t>(b) * kPeCapLarge + handle], v[k].d, v[k].pos);
        return 1;
      }
    }
  }
  printf("OK %d sequences of %d candidates (%lld runs of ties applied at once)\n", blocks, n, runs);
  std::vector<long long> hc(blocks * 2);
  (void)hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < blocks; ++b)  // (each sequence on a wave of its own: what a lone wave pays per update)
    printf("   sequence %d (shape %d): %lld cycles, %lld single updates, %d runs\n", b, b % kShapes, hc[2 * b], hc[2 * b + 1], hm[4 * b + 2]);
  return 0;
}
