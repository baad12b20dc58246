// abismal_amd device-side primitives (gfx950, wave64).
//
// Everything here runs with ONE wavefront per workgroup (64 threads): a read is
// mapped by one wave, so workgroup barriers degenerate to wave-level ordering
// and per-read state that is uniform across the wave lives in SGPRs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace abm {

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i16 = int16_t;
using i32 = int32_t;

constexpr u32 kKeyWeight = 25;   // src/AbismalIndex.hpp:68
constexpr u32 kKeyWeight3 = 16;  // src/AbismalIndex.hpp:69
constexpr u32 kWindow = 20;      // src/AbismalIndex.hpp:76 (12 in a short-read index: DevIndex::window)
constexpr u32 kHashMod3 = 43046721u;
constexpr u32 kMinReadLen = 44;  // src/abismal.cpp:212-213
constexpr u32 kLdsReadLen = 1024;  // reads up to this keep their traceback table ((L + 61) x 61 bytes) and CIGAR scratch in LDS
constexpr u32 kMaxReadLen = 32766; // longest read mapped: the reference refuses reads of padding_size = 32767 bases or more
                                   // (src/abismal.cpp:179-185, src/AbismalIndex.hpp:93); longer single-end reads than
                                   // kLdsReadLen go through a launch of their own with that table in global memory
constexpr u32 kMaxBand = 61;     // src/AbismalAlign.hpp:108,133
constexpr u32 kSeCap = 50;       // src/abismal.cpp:448
constexpr u32 kPeCapSmall = 32, kPeCapLarge = 32u << 10;  // src/abismal.cpp:861-862

constexpr u32 kFlagRC = 0x10, kFlagAmbig = 0x100, kFlagARich = 0x1000;

struct DevIndex {
  const u64 *genome;
  const u32 *counter, *counter_t, *counter_a;
  const u32 *index, *index_t, *index_a;
  u32 max_candidates;
  u32 window;   // seed window of this index: 20, or 12 (--enable-short, src/AbismalIndex.hpp:73-77)
  u32 min_len;  // shortest read that is mapped: key_weight + window - 1 (src/abismal.cpp:212-213)
  // The genome once more for the Hamming filter, as two bit planes: block k = 64 bases = {u64 low bits, u64 high
  // bits} of the base codes (code = index of the base's bit in its one-hot nibble).  A 100-base window is 2-3
  // blocks (32-48 bytes) instead of 64 bytes of nibbles, and the two copies are laid out half a 128-byte line
  // apart -- block k of planes[c] is at byte 16 k + 64 c of a line-aligned array -- so that every window lies
  // inside ONE line of one of them.  Null when the genome has nibbles that are not one-hot.
  const u64 *planes[2];
  // Two bits cannot say "N" (a blank nibble, which matches nothing: the long N runs the index leaves out, the
  // padding at both ends); the planes hold code 0 there.  nmap has one bit per chunk of 4096 bases, set if a window
  // starting in the chunk could reach a blank; the filter redoes such candidates (rare: they lie within a read
  // length of an N run) on the nibble array.
  const u32 *nmap;
  // Seed-extension tables (abm_ext.hip), all three or none: for every key of the hashed letters plus the next e2
  // (2-letter table) / e3 (3-letter tables) one entry {x = lo, y = size | len << 27 | state << 30} saying where the
  // narrowing loop of find_candidates / find_candidates_three stands after those letters, for ext_maxc candidates:
  // state 0 = finished with [lo, lo + size) after key weight + len letters, 1 = still open there (len = e), 2 = not
  // tabulated (bisect from the bucket's counters).
  const uint2 *ext2, *ext3t, *ext3a;
  u32 e2, e3, ext_maxc;
  // Window records (abm_ext.hip, build_window_records): every index entry once more, as the stretch of the bit planes
  // a candidate window of that entry can lie in -- wrec_blocks blocks of 64 bases {u64 low bits, u64 high bits}
  // starting wrec_back bases before the entry's position.  Seed offset i of a read puts its window at bit
  // wrec_back - i of the record.  A checked bucket's windows then are CONSECUTIVE records (48 bytes each for reads up
  // to 108 bases: 2.7 windows per 128-byte line) instead of one random line each.  Record number of entry e: e for
  // `index`, wrec_t0 + e for index_t, wrec_a0 + e for index_a.  Serves reads of up to wrec_max_len bases; null = none
  // (the filter then gathers its windows from `planes`).
  const u64 *wrec;
  u32 wrec_t0, wrec_a0;
  u32 wrec_blocks, wrec_back, wrec_max_len;
  u32 direct_min;  // pair kernels: ranges of at least this many entries are narrowed directly (narrow_direct); 0 = never
  // the index's chromosome table as `abismal-amd map` sees it (names and n + 1 starts, the two padding entries included):
  // what the single-end kernel needs to write a read's SAM text itself (SeArgs::sam_tail)
  const u32 *chrom_starts;    // [n_chroms + 1]
  const u32 *chrom_name_off;  // [n_chroms + 1] into chrom_names
  const char *chrom_names;
  u32 n_chroms;
};
constexpr u32 kSortDepth = 256;  // letters a bucket is sorted by (src/AbismalIndex.hpp: seed::n_sorting_positions)
// direct narrowing of big ranges (narrow_direct, abm_kernels_core.hpp): the pair kernels, ranges of at least kDirectMin entries
#ifndef ABM_PE_DIRECT_NARROWING
#define ABM_PE_DIRECT_NARROWING true
#endif
// (the single-end kernel: ABM_SE_DIRECT_NARROWING, abm_kernels_core.hpp)
#ifndef ABM_PE_DIRECT_MIN
#define ABM_PE_DIRECT_MIN 64  // (2x150 at hg38 scale, round 5's kernels: 6.11-6.38 M reads/s at 64, 6.04-6.26 at 128, profiles/r05_exp_final_knobs.log; round 3's kernels had their best at 128: profiles/r03_exp_pe_direct_threshold.log)
#endif
constexpr u32 kDirectMin = ABM_PE_DIRECT_MIN;
#ifndef ABM_SE_DIRECT_MIN
#define ABM_SE_DIRECT_MIN 64  // (every range beyond max_candidates = 100: 10 M x 100 bp 433 ms at 64, 440 at 128, 455 at 256, 468 at 512; profiles/r05_exp_se_direct_threshold.log)
#endif
constexpr u32 kDirectMinSe = ABM_SE_DIRECT_MIN;
constexpr u32 kPlaneBlock = 64;       // bases per bit-plane block
constexpr u32 kPlaneLineBlocks = 8;   // blocks per 128-byte line
constexpr u32 kPlaneChunkBits = 12;   // nmap: log2 of the bases per chunk
constexpr u32 kPlaneReach = 512;      // >= the bases a window of the cooperative filter spans (reads up to 448 + slack)

struct Hit {  // == abm_hit
  i16 diffs;
  u16 flags;
  u32 pos;
};

// ---- lane helpers -----------------------------------------------------------
__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x & 63u); }

// Between phases a wave needs its own LDS / global writes ordered before its later reads -- what __syncthreads()
// meant while every workgroup was one wave.  Waves of a multi-wave workgroup here never run in lock-step (each maps
// its own reads), so a workgroup barrier is exactly what must NOT be used.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int rdlane(int v, int l) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l));
}
__device__ __forceinline__ u32 rdlane(u32 v, int l) {
  return static_cast<u32>(rdlane(static_cast<int>(v), l));
}
__device__ __forceinline__ u64 rdlane(u64 v, int l) {
  const u32 lo = rdlane(static_cast<u32>(v), l), hi = rdlane(static_cast<u32>(v >> 32), l);
  return (static_cast<u64>(hi) << 32) | lo;
}
// v_writelane: one lane of a wave-resident array takes a uniform value
__device__ __forceinline__ void wrlane(int &v, int l, int x) { v = (lane_id() == l) ? x : v; }
__device__ __forceinline__ void wrlane(u32 &v, int l, u32 x) { v = (lane_id() == l) ? x : v; }

// Wave-wide scans in six DPP steps (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of rows 0 and 2 to
// rows 1 and 3, then lane 31 to rows 2 and 3): one v_add / v_max with a DPP operand each, against a ds_bpermute, a
// compare and a select per step of the shuffle form.  Lanes a step does not reach keep `old`, the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_from(u32 identity, u32 v) {
  return static_cast<u32>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ u32 wave_incl_sum(u32 x) {
  x += dpp_from<0x111, 0xf>(0u, x);
  x += dpp_from<0x112, 0xf>(0u, x);
  x += dpp_from<0x114, 0xf>(0u, x);
  x += dpp_from<0x118, 0xf>(0u, x);
  x += dpp_from<0x142, 0xa>(0u, x);  // row_bcast:15
  x += dpp_from<0x143, 0xc>(0u, x);  // row_bcast:31
  return x;
}
__device__ __forceinline__ u32 wave_excl_sum(u32 x, u32 &total) {
  const u32 inc = wave_incl_sum(x);
  total = rdlane(inc, 63);
  return inc - x;
}
// inclusive running maximum of unsigned values (identity 0)
__device__ __forceinline__ u32 wave_incl_max(u32 x) {
  x = max(x, dpp_from<0x111, 0xf>(0u, x));
  x = max(x, dpp_from<0x112, 0xf>(0u, x));
  x = max(x, dpp_from<0x114, 0xf>(0u, x));
  x = max(x, dpp_from<0x118, 0xf>(0u, x));
  x = max(x, dpp_from<0x142, 0xa>(0u, x));
  x = max(x, dpp_from<0x143, 0xc>(0u, x));
  return x;
}
__device__ __forceinline__ int wave_incl_max(int x) { return static_cast<int>(wave_incl_max(static_cast<u32>(x))); }

__device__ __forceinline__ u64 wave_max_u64(u64 x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const u64 y = __shfl_xor(x, d);
    x = y > x ? y : x;
  }
  return x;
}

// Results (hits, op counts, CIGAR slots and arena entries) are written THROUGH to memory at system scope (sc0 sc1):
// a host that takes them while the kernel is still running (abm_map_se_batch_sliced) must never find them waiting in
// an L2 for the end of the kernel -- a plain store to pinned host memory does (measured: a run whose slices were
// announced after an s_waitcnt alone wrote stale results).  A handful of stores per read; device buffers take them too.
__device__ __forceinline__ void store_out(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_out(Hit *p, const Hit &h) {
  static_assert(sizeof(Hit) == 8, "a hit is one 8-byte store");
  u64 bits;
  __builtin_memcpy(&bits, &h, 8);
  __hip_atomic_store(reinterpret_cast<u64 *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- sequence primitives ------------------------------------------------------
// read nibble: src/dna_four_bit_bisulfite.hpp:26-57
__device__ __forceinline__ u32 read_nibble(u32 c, bool a_alphabet) {
  c &= 0xDFu;  // fold case
  return c == 'A' ? (a_alphabet ? 5u : 1u)
       : c == 'C' ? 2u
       : c == 'G' ? 4u
       : c == 'T' ? (a_alphabet ? 8u : 10u)
                  : 0u;
}
// complement as the mapper does it (src/common.hpp:28-44): non-ACGT -> N
__device__ __forceinline__ u32 comp_base(u32 c) {
  return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
}
__device__ __forceinline__ u32 gnib(const u64 *__restrict__ g, u64 k) {
  return static_cast<u32>(g[k >> 4] >> ((k & 15u) << 2)) & 15u;
}
// src/AbismalIndex.hpp:255-269, src/abismal.cpp:1196-1203
__device__ __forceinline__ u32 bit2(u32 nt) { return (nt & 5u) == 0u; }
__device__ __forceinline__ u32 trit(u32 nt, bool g_to_a) {
  return g_to_a ? ((((nt & 8u) != 0u) << 1) | ((nt & 2u) != 0u))
                : ((((nt & 4u) != 0u) << 1) | ((nt & 1u) != 0u));
}
__device__ __forceinline__ u32 sortsym3(u32 nt, bool g_to_a) { return g_to_a ? (nt & 10u) : (nt & 5u); }

}  // namespace abm
