// GPU test program (built and run by tests/test_gpu_se_set.py): the lane-resident single-end candidate set
// (SeSet::admit -> replace_top / sift_up) against libstdc++'s pop_heap / push_heap on the host -- the very calls
// se_candidates::update makes (src/abismal.cpp:394-404) -- over random update sequences of several shapes.
// Prints "OK <n sequences>" or the first mismatch, and the cycles per update of wave 0.
#include "../../abismal_amd/csrc/abm_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace abm;

struct El { short d; unsigned pos; };
static bool by_d(const El &a, const El &b) { return a.d < b.d; }

__device__ __host__ inline int draw(unsigned seq, unsigned idx, int shape) {
  unsigned x = (seq * 2654435761u) ^ (idx * 40503u + 12345u);
  x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
  const unsigned r = x >> 4;
  switch (shape) {
    case 0: return 1 + r % 12;                          // uniform
    case 1: return (r % 16) ? 3 : 1 + r % 3;            // mostly one value
    case 2: return 1 + (r % 40 == 0 ? 0 : 1 + r % 2);   // 2-3 with rare 1s
    case 3: return 30 - min(static_cast<int>(idx / 512), 28) + r % 2;  // drifting down
    case 4: return (r % 200) ? 2 : 1;                   // long runs of ties, a few better hits among them
    default: return 1 + r % 3;
  }
}
constexpr int kShapes = 6;

// candidates arrive 64 at a time, one per lane, and are replayed in order exactly as seed_pass's replay() does it
__global__ __launch_bounds__(64) void run(int n_chunks, int *keys, unsigned *pos, int *meta, long long *cycles) {
  SeSet S;
  S.begin_read(100);
  const int shape = blockIdx.x % kShapes, lane = threadIdx.x;
  int n_runs = 0;
  const long long t0 = clock64();
  for (int i = 0; i < n_chunks; ++i) {
    const unsigned idx = i * 64 + lane;
    const int h = draw(blockIdx.x, idx, shape);
    const unsigned p = 1000u + idx;
    u64 todo = __ballot(h <= S.cutoff);
    while (todo) {
      if (S.tie_run(todo, __ballot(h == S.cutoff), p, 0u)) { ++n_runs; continue; }
      const int l = __builtin_ctzll(todo);
      const int before = S.cutoff;
      S.admit(false, rdlane(h, l), 0u, rdlane(p, l));
      todo &= ~(((1ull << l) << 1) - 1);
      if (S.cutoff < before) todo &= __ballot(h <= S.cutoff);
    }
  }
  const long long t1 = clock64();
  keys[blockIdx.x * 64 + threadIdx.x] = S.hk;
  pos[blockIdx.x * 64 + threadIdx.x] = S.pp;
  if (threadIdx.x == 0) { meta[3 * blockIdx.x] = S.sz; meta[3 * blockIdx.x + 1] = S.cutoff; meta[3 * blockIdx.x + 2] = n_runs; cycles[blockIdx.x] = t1 - t0; }
}

int main() {
  const int blocks = 240, chunks = 60, n = chunks * 64;
  int *dk, *dm; unsigned *dp; long long *dc;
  hipMalloc(&dk, blocks * 64 * 4); hipMalloc(&dp, blocks * 64 * 4); hipMalloc(&dm, blocks * 12); hipMalloc(&dc, blocks * 8);
  hipLaunchKernelGGL(run, dim3(blocks), dim3(64), 0, 0, chunks, dk, dp, dm, dc);
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAIL launch\n"); return 1; }
  std::vector<int> hk(blocks * 64), hm(blocks * 3); std::vector<unsigned> hp(blocks * 64); std::vector<long long> hc(blocks);
  hipMemcpy(hk.data(), dk, hk.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hp.data(), dp, hp.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hm.data(), dm, hm.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost);
  long long runs = 0;
  for (int b = 0; b < blocks; ++b) {
    // se_candidates: v[0] = {0.4 * readlen, pos 0}, sz 1, cutoff = v[0].diffs; update() as in the oracle's SeSet::admit
    std::vector<El> v(50);
    v[0] = El{static_cast<short>(0.4 * 100), 0u};
    int sz = 1, cutoff = v[0].d;
    for (int i = 0; i < n; ++i) {
      const int d = draw(b, i, b % kShapes);
      if (d > cutoff) continue;
      if (sz == 50) { std::pop_heap(v.begin(), v.begin() + sz, by_d); v[sz - 1] = El{static_cast<short>(d), 1000u + i}; }
      else v[sz++] = El{static_cast<short>(d), 1000u + i};
      std::push_heap(v.begin(), v.begin() + sz, by_d);
      cutoff = v[0].d;
    }
    runs += hm[3 * b + 2];
    if (hm[3 * b] != sz || hm[3 * b + 1] != cutoff) { printf("FAIL seq %d: size/cutoff %d/%d vs %d/%d\n", b, hm[3 * b], hm[3 * b + 1], sz, cutoff); return 1; }
    for (int k = 0; k < sz; ++k) {
      const int key = hk[b * 64 + k], slot = key & 255;
      if ((key >> 8) != v[k].d || hp[b * 64 + slot] != v[k].pos) {
        printf("FAIL seq %d (shape %d) heap[%d]: gpu d %d pos %u, libstdc++ d %d pos %u\n", b, b % kShapes, k, key >> 8, hp[b * 64 + slot], v[k].d, v[k].pos);
        return 1;
      }
    }
  }
  printf("OK %d sequences of %d candidates (%lld runs of ties applied at once); wave 0: %.1f cycles per candidate\n", blocks, n, runs, double(hc[0]) / n);
  return 0;
}
