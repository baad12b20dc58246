// abismal_amd: seed-extension tables (DevIndex::ext2 / ext3t / ext3a), derived on the device at index upload.
//
// find_candidates / find_candidates_three (src/abismal.cpp:1163-1259) narrow a seed's bucket letter by letter,
// each letter a std::lower_bound bisection over the bucket: log2(bucket) DEPENDENT pairs of loads (index entry,
// then the genome letter behind it) per letter, and at hg38 scale a third of the seed offsets needs at least one.
// Buckets are sorted by exactly those letters (AbismalIndex::sort_buckets, src/AbismalIndex.cpp:857-978), so the
// sub-bucket a bisection finds is the run of entries whose next letter equals the read's -- a function of the
// read's next letters alone.  With 288 GB of HBM that function is simply tabulated: for every key of the hashed
// letters PLUS the next e (25 + e 2-letter bits; 16 + e base-3 digits) one 8-byte entry holds where the narrowing
// loop stands after those e letters -- finished (range <= max_candidates, or emptied and stepped back as the
// reference does), or still open at 25 + e / 16 + e letters with the range reached.  A seed offset then costs ONE
// independent 8-byte load per table instead of two counter loads and the bisections of its first e letters;
// only ranges still open continue with the bisection loop (narrow_both), from where the table left off.
//
// Exactness does not rest on the sort: while the boundary array is built every pair of neighbouring index
// entries is checked for key order, and a base bucket with entries out of order gets "fallback" entries, which
// send the kernel down the bisection path from the bucket's counters.
#include "abm_kernels.hpp"

#include <algorithm>

namespace abm {

namespace {

constexpr u32 kGapInline = 64;  // a thread fills gaps of up to this many keys itself; longer ones go to the gap list

struct GapList { u64 *rec; u32 *count; u32 cap; };  // rec[3k..]: first key, last key, value

// key of table `mode` (0: 2-letter, 1: 3-letter C->T alphabet, 2: 3-letter G->A alphabet) of `depth` letters
// at genome position pos, exactly as the index hashes and sorts them (get_1bit_hash / get_base_3_hash,
// src/AbismalIndex.hpp:285-305; the sort's letter functions are the same: oracle/abo_index.cpp sort_table)
__device__ __forceinline__ u64 ext_key(const u64 *__restrict__ genome, u64 pos, u32 depth, int mode) {
  u64 key = 0;
  u64 w = genome[pos >> 4];
  u32 at = static_cast<u32>(pos & 15u);
  u64 wi = pos >> 4;
  for (u32 j = 0; j < depth; ++j) {
    const u32 nib = static_cast<u32>(w >> (at << 2)) & 15u;
    if (mode == 0) key = (key << 1) | bit2(nib);
    else {
      // sort symbol classes of find_candidates_three: below mid, below top, the rest
      const u32 s = sortsym3(nib, mode == 2);
      const u32 mid = mode == 2 ? 2u : 1u, top = mode == 2 ? 8u : 4u;
      key = key * 3u + (s < mid ? 0u : (s < top ? 1u : 2u));
    }
    if (++at == 16) { at = 0; w = genome[++wi]; }
  }
  return key;
}

// bound[K] = number of index entries whose key is below K, for K in [0, n_keys]: thread k looks at entries k - 1 and
// k and fills the keys between them.
__global__ __launch_bounds__(256) void ext_bounds_kernel(const u64 *__restrict__ genome, const u32 *__restrict__ index, u64 n_idx,
                                                         const u32 *__restrict__ counter, u32 depth, int mode, u64 n_keys,
                                                         u32 base_shift_or_div, u32 *__restrict__ bound,
                                                         u32 *__restrict__ bad /*bitmap over base buckets*/, u32 *__restrict__ fail,
                                                         GapList gaps) {
  const u64 k = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k > n_idx) return;
  // thread n_idx closes the array: keys above the last entry's
  const u64 key = k < n_idx ? ext_key(genome, index[k], depth, mode) : n_keys;
  if (k < n_idx) {  // the hashed letters must name the bucket the entry lies in; if not, no table at all
    const u64 b = mode == 0 ? (key >> base_shift_or_div) : (key / base_shift_or_div);
    if (!(counter[b] <= k && k < counter[b + 1])) { atomicOr(fail, 1u); return; }
  }
  u64 first;  // first key this thread fills
  if (k == 0) first = 0;
  else {
    const u64 prev = ext_key(genome, index[k - 1], depth, mode);
    if (k < n_idx && key < prev) {  // out of order: both base buckets fall back to bisection
      const u64 b0 = mode == 0 ? (key >> base_shift_or_div) : (key / base_shift_or_div);
      const u64 b1 = mode == 0 ? (prev >> base_shift_or_div) : (prev / base_shift_or_div);
      atomicOr(&bad[b0 >> 5], 1u << (b0 & 31u));
      atomicOr(&bad[b1 >> 5], 1u << (b1 & 31u));
      return;
    }
    first = prev + 1;
  }
  if (first > key) return;  // same key as the entry before
  const u64 n = key - first + 1;
  if (n <= kGapInline) { for (u64 x = first; x <= key; ++x) bound[x] = static_cast<u32>(k); return; }
  const u32 slot = atomicAdd(gaps.count, 1u);
  if (slot < gaps.cap) { gaps.rec[3ull * slot] = first; gaps.rec[3ull * slot + 1] = key; gaps.rec[3ull * slot + 2] = k; }
  else for (u64 x = first; x <= key; ++x) bound[x] = static_cast<u32>(k);  // (list full: slow but complete)
}

__global__ __launch_bounds__(256) void ext_gaps_kernel(GapList gaps, u32 *__restrict__ bound) {
  const u32 n = min(*gaps.count, gaps.cap);
  for (u32 g = blockIdx.x; g < n; g += gridDim.x) {
    const u64 first = gaps.rec[3ull * g], last = gaps.rec[3ull * g + 1];
    const u32 v = static_cast<u32>(gaps.rec[3ull * g + 2]);
    for (u64 x = first + threadIdx.x; x <= last; x += blockDim.x) bound[x] = v;
  }
}

// entry of key K: the state of the reference's narrowing loop after the table's letters (see the file comment)
__global__ __launch_bounds__(256) void ext_entries_kernel(const u32 *__restrict__ bound, const u32 *__restrict__ counter,
                                                          const u32 *__restrict__ bad, u64 n_keys, u32 extra, int mode, u32 maxc,
                                                          uint2 *__restrict__ out) {
  // span[s] = keys under one prefix that is s letters short of the full depth
  u64 span[8];
  span[0] = 1;
  for (u32 s = 1; s <= extra; ++s) span[s] = span[s - 1] * (mode == 0 ? 2u : 3u);
  // (grid-stride: a launch cannot have 2^32 threads, which the deepest 2-letter table has keys)
  for (u64 K = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x; K < n_keys; K += static_cast<u64>(gridDim.x) * blockDim.x) {
  const u64 base = K / span[extra];
  if ((bad[base >> 5] >> (base & 31u)) & 1u) { out[K] = make_uint2(0u, 2u << 30); continue; }
  // boundaries that coincide with a base bucket's come from the index's own counter array
  auto bnd = [&](u64 x) -> u32 { return x % span[extra] == 0 ? counter[x / span[extra]] : bound[x]; };
  u32 p = 0;  // letters beyond the hashed ones
  u32 lo = counter[base], hi = counter[base + 1], plo = lo, phi = hi;
  while (hi - lo > maxc && p < extra) {
    plo = lo; phi = hi;
    ++p;
    const u64 first = K - K % span[extra - p];
    lo = bnd(first);
    hi = bnd(first + span[extra - p]);
  }
  u32 state = 0, len = p;
  if (lo == hi) {
    if (p > 0) { len = p - 1; lo = plo; hi = phi; }  // emptied: the reference steps back one letter
    else len = 0;
  }
  else if (hi - lo > maxc) state = 1;  // (p == extra) still open
  const u32 size = hi - lo;
  if (size >= (1u << 27)) { out[K] = make_uint2(0u, 2u << 30); continue; }
  out[K] = make_uint2(lo, size | (len << 27) | (state << 30));
  }
}

}  // namespace

u64 ext_keys(int mode, u32 extra) {
  u64 n = mode == 0 ? (1ull << kKeyWeight) : static_cast<u64>(kHashMod3);
  for (u32 s = 0; s < extra; ++s) n *= mode == 0 ? 2u : 3u;
  return n;
}

// Builds the table of `mode` with `extra` letters into out[ext_keys(mode, extra)].  scratch: (n_keys + 1) u32 for
// the boundary array, then ceil(base buckets / 32) u32 of bitmap, then the gap list (1 + 3 * 2 * kExtGapCap u32).
size_t ext_scratch_bytes(int mode, u32 extra) {
  const u64 n_keys = ext_keys(mode, extra);
  const u64 n_base = ext_keys(mode, 0);
  return static_cast<size_t>((n_keys + 1) * 4 + ((n_base + 31) / 32 + 1) * 4 + 64 + 8ull * 3 * kExtGapCap + 64);
}

hipError_t build_ext_table(const DevIndex &ix, int mode, u32 extra, u32 maxc, u64 n_idx, uint2 *out, void *scratch, u32 *d_fail,
                           hipStream_t st) {
  const u64 n_keys = ext_keys(mode, extra), n_base = ext_keys(mode, 0);
  const u32 *index = mode == 0 ? ix.index : (mode == 1 ? ix.index_t : ix.index_a);
  const u32 *counter = mode == 0 ? ix.counter : (mode == 1 ? ix.counter_t : ix.counter_a);
  char *p = static_cast<char *>(scratch);
  u32 *bound = reinterpret_cast<u32 *>(p); p += (n_keys + 1) * 4;
  p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 63) & ~static_cast<uintptr_t>(63));
  u32 *bad = reinterpret_cast<u32 *>(p); p += ((n_base + 31) / 32 + 1) * 4;
  p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(p) + 63) & ~static_cast<uintptr_t>(63));
  GapList gaps;
  gaps.count = reinterpret_cast<u32 *>(p); p += 64;
  gaps.rec = reinterpret_cast<u64 *>(p);
  gaps.cap = kExtGapCap;
  hipError_t e = hipMemsetAsync(bad, 0, ((n_base + 31) / 32 + 1) * 4, st);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(gaps.count, 0, 64, st);
  if (e != hipSuccess) return e;
  const u32 depth = (mode == 0 ? kKeyWeight : kKeyWeight3) + extra;
  u32 per_base = 1;
  for (u32 s = 0; s < extra; ++s) per_base *= 3u;
  const u32 shift_or_div = mode == 0 ? extra : per_base;
  const u64 threads = n_idx + 1;
  hipLaunchKernelGGL(ext_bounds_kernel, dim3(static_cast<u32>((threads + 255) / 256)), dim3(256), 0, st, ix.genome, index, n_idx,
                     counter, depth, mode, n_keys, shift_or_div, bound, bad, d_fail, gaps);
  hipLaunchKernelGGL(ext_gaps_kernel, dim3(4096), dim3(256), 0, st, gaps, bound);
  hipLaunchKernelGGL(ext_entries_kernel, dim3(static_cast<u32>(std::min<u64>((n_keys + 255) / 256, 1u << 22))), dim3(256), 0, st, bound, counter, bad, n_keys,
                     extra, mode, maxc, out);
  return hipGetLastError();
}

// ---- window records (DevIndex::wrec) -------------------------------------------------------------------------
// One thread per (index entry, block): block b of entry e's record is the 64 bases from position index[e] - back + 64 b
// of the genome's bit planes.  Bases before the genome's first (never reached: the index leaves the padding out) read
// as code 0, like the planes' own guard blocks past the end.
__global__ __launch_bounds__(256) void window_records_kernel(const u64 *__restrict__ planes0, u64 n_plane_blocks,
                                                             const u64 *__restrict__ genome, u64 n_bases,
                                                             const u32 *__restrict__ index, u64 n_entries, u64 first_record,
                                                             u32 blocks, u32 back, u64 *__restrict__ out) {
  const u64 t = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  const u64 e = t / blocks;
  const u32 b = static_cast<u32>(t % blocks);
  if (e >= n_entries) return;
  const long long bit = static_cast<long long>(index[e]) - static_cast<long long>(back) + 64ll * b;
  auto plane = [&](long long k, u32 which) -> u64 {
    return k >= 0 && static_cast<u64>(k) < n_plane_blocks ? planes0[2 * static_cast<u64>(k) + which] : 0ull;
  };
  const long long k = bit >> 6;  // (arithmetic shift: floor)
  const u32 s = static_cast<u32>(bit & 63);
  u64 lo = plane(k, 0) >> s, hi = plane(k, 1) >> s;
  if (s) { lo |= plane(k + 1, 0) << (64 - s); hi |= plane(k + 1, 1) << (64 - s); }
  if (b == blocks - 1) {
    // the record's spare base (its last: 2 max_len - key weight = 64 blocks - 1 bases are ever looked at) says whether
    // the stretch holds a blank nibble, which two bits cannot express: the narrowing probes ask (record_nibble)
    lo &= ~(1ull << 63); hi &= ~(1ull << 63);
    const long long first = static_cast<long long>(index[e]) - static_cast<long long>(back), end = first + 64ll * blocks - 1;
    bool blank = false;
    for (long long q = first < 0 ? 0 : first; q < end && !blank; ) {
      if (static_cast<u64>(q) >= n_bases) { blank = true; break; }  // (past the genome: padding, blank as well)
      const u64 w = genome[q >> 4];
      const u32 from = static_cast<u32>(q & 15), upto = static_cast<u32>(end - q < 16 - from ? from + (end - q) : 16);
      const u64 ones = (w | (w >> 1) | (w >> 2) | (w >> 3)) & 0x1111111111111111ull;
      u64 part = 0x1111111111111111ull;
      if (upto < 16) part &= (1ull << (4 * upto)) - 1;
      part &= ~((1ull << (4 * from)) - 1);
      if ((~ones & part) != 0) blank = true;
      q += upto - from;
    }
    if (blank) lo |= 1ull << 63;
  }
  u64 *o = out + 2 * ((first_record + e) * blocks + b);
  o[0] = lo; o[1] = hi;
}

u32 window_record_max_len(u32 blocks) { return blocks ? (64u * blocks + kKeyWeight) / 2u : 0u; }
u32 window_record_blocks_for(u32 max_len) {
  u32 b = 1;
  while (window_record_max_len(b) < max_len) ++b;
  return b;
}
size_t window_record_bytes(u64 n_entries, u32 blocks) { return (n_entries * blocks + 2) * 16 + 256; }
hipError_t build_window_records(const DevIndex &ix, u64 n_plane_blocks, u64 n_bases, const u64 n_idx[3], u32 blocks, u64 *out, hipStream_t st) {
  const u32 back = window_record_max_len(blocks) - kKeyWeight;
  const u32 *arrays[3] = {ix.index, ix.index_t, ix.index_a};
  u64 first = 0;
  for (int m = 0; m < 3; ++m) {
    const u64 threads = n_idx[m] * blocks;
    if (threads)
      hipLaunchKernelGGL(window_records_kernel, dim3(static_cast<u32>((threads + 255) / 256)), dim3(256), 0, st, ix.planes[0], n_plane_blocks,
                         ix.genome, n_bases, arrays[m], n_idx[m], first, blocks, back, out);
    first += n_idx[m];
  }
  return hipGetLastError();
}

}  // namespace abm
