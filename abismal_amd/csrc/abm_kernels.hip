// abismal_amd HIP kernels for gfx950: read packing and the single-end mapping
// kernel (seed probe -> bucket narrowing -> Hamming filter -> ordered replay
// into the candidate set -> banded alignment -> CIGAR).  One wavefront maps one
// read; see DESIGN.md for the data layout and the reasoning.
#include <type_traits>
#include "abm_kernels_core.hpp"

#include <hipcub/hipcub.hpp>

#ifndef ABM_SE_WAVES_PER_SIMD
#define ABM_SE_WAVES_PER_SIMD 5  // 96 VGPRs per lane, 20 waves per CU
#endif


namespace abm {

// =============================================================================
// Kernel 1: ASCII reads -> 4-bit bisulfite encodings, one coalesced pass.
// For every read four packed streams are produced (forward/revcomp x T-/A-rich
// alphabet), 16 bases per u64, base j at bits 4(j%16), tail nibbles 0xF
// (prep_read + revcomp + pack_read: src/abismal.cpp:1377-1426, src/common.hpp:28-44).
// Layout: packed[read][enc][word], enc = rc*2 + alphabet, stride W words.
// =============================================================================
__global__ __launch_bounds__(256) void pack_reads_kernel(const char *__restrict__ blob,
                                                         const u64 *__restrict__ off, u64 n_reads,
                                                         u32 W, u64 *__restrict__ packed,
                                                         u32 *__restrict__ lens) {
  const u64 gid = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  const u64 r = gid / W;
  const u32 w = static_cast<u32>(gid % W);
  if (r >= n_reads)
    return;
  const u64 b = off[r];
  const u32 L = static_cast<u32>(off[r + 1] - b);
  if (w == 0)
    lens[r] = L;
  u64 ft = 0, fa = 0, rt = 0, ra = 0;
  for (u32 j = 0; j < 16; ++j) {
    const u32 k = w * 16 + j;
    u32 nft = 15, nfa = 15, nrt = 15, nra = 15;
    if (k < L) {
      const u32 cf = static_cast<u8>(blob[b + k]);
      const u32 cr = comp_base(static_cast<u8>(blob[b + (L - 1 - k)]));
      nft = read_nibble(cf, false);
      nfa = read_nibble(cf, true);
      nrt = read_nibble(cr, false);
      nra = read_nibble(cr, true);
    }
    const u32 sh = j << 2;
    ft |= static_cast<u64>(nft) << sh;
    fa |= static_cast<u64>(nfa) << sh;
    rt |= static_cast<u64>(nrt) << sh;
    ra |= static_cast<u64>(nra) << sh;
  }
  if (w * 16 >= L)  // words past the read are never consumed; keep them defined
    ft = fa = rt = ra = ~0ull;
  u64 *dst = packed + (r * 4) * W + w;
  dst[0] = ft;
  dst[W] = fa;
  dst[2 * W] = rt;
  dst[3 * W] = ra;
}

// =============================================================================
// Index upload: the genome's bit planes (DevIndex::planes).  Thread per block of 64 bases.
// =============================================================================
__global__ __launch_bounds__(256) void make_planes_kernel(const u64 *__restrict__ genome, u64 n_words, u64 n_bases,
                                                          u64 n_blocks, u64 *__restrict__ p0, u64 *__restrict__ p1,
                                                          u32 *__restrict__ nmap, u32 *__restrict__ bad) {
  const u64 b = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  u64 lo = 0, hi = 0;
  bool multi = false, blank = false;
  for (u32 k = 0; k < 4; ++k) {
    const u64 w = 4 * b + k;
    const u64 x = w < n_words ? genome[w] : 0ull;
    // code of a one-hot nibble 1, 2, 4, 8 = 0, 1, 2, 3: low bit from bits 1|3, high bit from bits 2|3
    lo |= static_cast<u64>(every_fourth_bit((x >> 1) | (x >> 3))) << (16 * k);
    hi |= static_cast<u64>(every_fourth_bit((x >> 2) | (x >> 3))) << (16 * k);
    const u64 ones = (x & 0x1111111111111111ull) + ((x >> 1) & 0x1111111111111111ull) +
                     ((x >> 2) & 0x1111111111111111ull) + ((x >> 3) & 0x1111111111111111ull);
    u64 part = 0x1111111111111111ull;  // the nibbles of this word that belong to the genome
    const u64 first = 16 * w;
    if (first >= n_bases) part = 0;
    else if (n_bases - first < 16) part &= (1ull << (4 * (n_bases - first))) - 1;
    if (ones & (part * 14ull)) multi = true;                         // two or more bits: an IUPAC letter
    if ((~(ones | (ones >> 1) | (ones >> 2)) & part) != 0) blank = true;  // no bit: N (src/dna_four_bit_bisulfite.hpp:156-165)
  }
  p0[2 * b] = lo; p0[2 * b + 1] = hi;
  p1[2 * b] = lo; p1[2 * b + 1] = hi;
  if (multi) atomicOr(bad, 1u);
  if (blank) {
    // a window that starts up to kPlaneReach bases before this block can see the blank
    const u64 at = b * kPlaneBlock, from = at >= kPlaneReach ? at - kPlaneReach : 0;
    for (u64 c = from >> kPlaneChunkBits; c <= (at + kPlaneBlock - 1) >> kPlaneChunkBits; ++c)
      atomicOr(&nmap[c >> 5], 1u << (c & 31u));
  }
}

// =============================================================================
// Kernel 2: single-end mapping, one wave per read (persistent, strided).
// =============================================================================
// LONG: the launch for reads beyond kLdsReadLen bases (up to kMaxReadLen).  Such a read's packed encodings, bit
// strings and two alignment windows still fit LDS (117 KB at 32766 bases: one wave per CU), its traceback table
// ((L + 61) x 61 bytes, 2 MB) and CIGAR scratch lie in a per-wave piece of global memory, its bands are 61 lanes
// wide (one alignment per set and round), its filter runs on the nibble array (COOP is false), its scores wrap at 16
// bits like the reference's, and the reads are the ones listed in `order`, their encodings packed by list position.
// ---- a read's SAM text written by the wave that mapped it (SeArgs::sam_tail) -------------------------------------
// format_se (src/abismal.cpp:481-545) + the host's put_record, minus QNAME: the wave has everything else -- hit,
// CIGAR (its first ops are still in LDS: CigarSink::fin), conversion type -- and the read's text is on the device.
// Scalar fields are written byte by byte by lane 0 into an LDS line buffer (a few hundred scalar instructions: 1 % of a
// read's work), SEQ by all lanes, and the line leaves as 4-byte words, written through like the other results.
// Returns the line's length, 0 for "no record", 0xFFFFFFFF for "the host formats this one".
struct SamWriter {
  u8 *buf;
  u32 w, cap;
  __device__ __forceinline__ void put(u32 c) { if (w < cap && lane_id() == 0) buf[w] = static_cast<u8>(c); ++w; }
  __device__ __forceinline__ void put_uint(u32 v) {
    u32 digits = 1;
    for (u32 t = v; t >= 10u; t /= 10u) ++digits;
    u32 at = w + digits;
    w = at;
    do { --at; if (at < cap && lane_id() == 0) buf[at] = static_cast<u8>('0' + v % 10u); v /= 10u; } while (v);
  }
  __device__ __forceinline__ void put_str(const char *s, u32 n) { for (u32 i = 0; i < n; ++i) put(static_cast<u8>(s[i])); }
};
__device__ __forceinline__ u32 format_sam_tail(const SeArgs &a, u8 *line /*LDS, 4-byte aligned, a.sam_stride bytes*/, const u32 *fin, u64 r,
                                               u32 L, const Hit &best, u32 n_ops) {
  const int lane = lane_id();
  const bool ambig = (best.flags & kFlagAmbig) != 0;
  if (best.pos == 0 || L == 0 || (ambig && !a.sam_allow_ambig)) return 0;
  if (n_ops == 0 || n_ops > a.cig_stride || n_ops > kSeCap) return 0xFFFFFFFFu;
  wave_sync();  // (fin was written by whichever lanes stored the CIGAR)
  u32 reflen = 0;
  for (u32 k = 0; k < n_ops; ++k) {
    const u32 v = static_cast<u32>(uni(static_cast<int>(fin[k]))), op = v & 15u;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += v >> 4;
  }
  // Chroms::locate: the last start <= pos; the record exists if the alignment ends inside that chromosome
  const DevIndex &ix = a.ix;
  u32 lo = 0, n = ix.n_chroms + 1;  // upper_bound over starts[0 .. n_chroms]
  while (n > 0) {
    const u32 half = n >> 1;
    const u32 sv = static_cast<u32>(uni(static_cast<int>(ix.chrom_starts[lo + half])));
    if (!(best.pos < sv)) { lo += half + 1; n -= half + 1; } else n = half;
  }
  if (lo == 0 || lo > ix.n_chroms) return 0;
  const u32 chrom = lo - 1;
  const u32 c0 = static_cast<u32>(uni(static_cast<int>(ix.chrom_starts[chrom]))), c1 = static_cast<u32>(uni(static_cast<int>(ix.chrom_starts[chrom + 1])));
  if (static_cast<u64>(best.pos) + reflen > c1) return 0;
  const u32 n0 = static_cast<u32>(uni(static_cast<int>(ix.chrom_name_off[chrom]))), n1 = static_cast<u32>(uni(static_cast<int>(ix.chrom_name_off[chrom + 1])));
  const bool rc = (best.flags & kFlagRC) != 0;
  SamWriter o{line, 0, a.sam_stride};
  o.put('\t');
  o.put_uint((rc ? 0x10u : 0u) | ((a.sam_allow_ambig && ambig) ? 0x100u : 0u));
  o.put('\t');
  for (u32 i = n0; i < n1; ++i) o.put(static_cast<u8>(uni(static_cast<int>(ix.chrom_names[i]))));
  o.put('\t');
  o.put_uint(best.pos - c0 + 1u);
  o.put_str("\t255\t", 5);
  for (u32 k = 0; k < n_ops; ++k) {
    const u32 v = static_cast<u32>(uni(static_cast<int>(fin[k])));
    o.put_uint(v >> 4);
    o.put(static_cast<u8>("MIDNSHP=XB"[min(v & 15u, 9u)]));
  }
  o.put_str("\t*\t0\t0\t", 7);
  // SEQ as htslib prints it after its 4-bit round trip: IUPAC letters upper-cased, everything else N; a reverse-strand
  // hit shows the reverse complement as the mapper makes it (src/common.hpp:28-44: A<->T, C<->G, everything else N)
  const char *seq = a.blob + a.off[r];
  for (u32 i = lane; i < L; i += 64) {
    u32 c = static_cast<u8>(seq[rc ? L - 1 - i : i]);
    if (rc) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    else {
      const u32 u = (c >= 'a' && c <= 'z') ? c - 32u : c;
      // "=ACMGRSVTWYHKDBN": the letters A B C D G H K M N R S T V W Y
      const bool ok = u == '=' || (u >= 'A' && u <= 'Z' && ((0x016E34CFu >> (u - 'A')) & 1u));
      c = ok ? u : 'N';
    }
    if (o.w + i < o.cap) line[o.w + i] = static_cast<u8>(c);
  }
  o.w += L;
  o.put_str("\t*\tNM:i:", 8);
  const int nm = best.diffs;
  if (nm < 0) { o.put('-'); o.put_uint(static_cast<u32>(-nm)); } else o.put_uint(static_cast<u32>(nm));
  o.put_str("\tCV:A:", 6);
  o.put((best.flags & kFlagARich) ? 'A' : 'T');
  o.put('\n');
  if (o.w > o.cap) return 0xFFFFFFFFu;
  wave_sync();
  u32 *dst = reinterpret_cast<u32 *>(a.sam_tail + r * a.sam_stride);
  const u32 *src = reinterpret_cast<const u32 *>(line);
  for (u32 k = lane; k < (o.w + 3) / 4; k += 64) store_out(dst + k, src[k]);
  return o.w;
}

template <bool TIMED, bool COOP, bool LONG, bool REC = false>
__device__ __forceinline__ void map_se_body(const SeArgs &a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  WaveLds lds;
  lds.W = a.W;
  lds.WB = a.WB;
  lds.qpk = reinterpret_cast<u64 *>(smem);
  lds.GW = a.GW;
  lds.qbits = lds.qpk + 4 * a.W;
  lds.MB = LONG ? 0u : (a.max_len + kPlaneBlock - 1) / kPlaneBlock;
  lds.qmask = lds.qbits + 4 * a.WB;
  lds.max_jobs = LONG ? 2u : kMaxJobs;
  {
    unsigned char *p = reinterpret_cast<unsigned char *>(lds.qmask + 4 * lds.MB * 4);
    const u32 cap2 = (a.ctmp_cap + 1) & ~1u;
    if (LONG) lds.ctmp = a.long_ctmp + static_cast<u64>(blockIdx.x) * cap2;
    else { lds.ctmp = reinterpret_cast<u32 *>(p); p += cap2 * 4; }
    lds.jpos = reinterpret_cast<u32 *>(p); p += kSeCap * 4;
    lds.jdf = reinterpret_cast<u32 *>(p); p += kSeCap * 4;
    lds.gwin = reinterpret_cast<u64 *>(p); p += lds.max_jobs * a.GW * 8;
    lds.pcache = reinterpret_cast<u64 *>(p); p += (8u << kPosCacheBits) + (LONG ? 0u : a.tb_extra);
    // the traceback table overlays window slots 1.. and the window cache (a traceback uses slot 0 only)
    lds.tb = LONG ? a.long_tb + static_cast<u64>(blockIdx.x) * a.long_tb_bytes : reinterpret_cast<u8 *>(lds.gwin + a.GW);
    lds.lbest = reinterpret_cast<int *>(p); p += 64 * 4;
    lds.smark = reinterpret_cast<u32 *>(p); p += 128 * 4;
    lds.sdelta = reinterpret_cast<u32 *>(p); p += 128 * 4;
    lds.mark = reinterpret_cast<u16 *>(p);
  }
  lds.hres = reinterpret_cast<u16 *>(lds.lbest);  // 128 x u16 = the 64 ints of lbest, idle during the seed passes
  lds.G = a.G;
  lds.smark[lane] = 0; lds.smark[64 + lane] = 0;
  u32 seg_epoch = 0;

  // (rc, a_rich) calls per mode, in the reference's order
  // T-rich :1556-1572 | A-rich (-A/-P) | random PBAT :1649-1676
  const u32 n_calls = a.mode == 2 ? 4u : 2u;
  const u32 call_rc = a.mode == 2 ? 0xCu /*0,0,1,1*/ : 0x2u /*0,1*/;
  const u32 call_ar = a.mode == 2 ? 0x6u /*0,1,1,0*/ : (a.mode == 1 ? 0x3u : 0x0u);

  WorkTally wt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const CigarSink sink = {a.cig_stride, a.ctmp_cap, a.cig_arena, a.cig_arena_count, a.cig_arena_cap, (!LONG && a.sam_tail) ? lds.jpos : nullptr};
  const u64 n_items = a.n_reads;
  long long t_begin = 0, t_a = 0, t_b = 0;
  ABM_STAMP(t_begin);
  u32 n_aln = 0, n_single = 0;
  bool overflow = false, too_long = false;

  // Reads are handed out by one device-wide counter (zeroed before every launch): work per read spans four orders
  // of magnitude, so a static split would leave the launch waiting on whichever wave drew the heaviest reads.  A
  // wave's FIRST read is dealt statically (its own number in the heaviest-first order), the rest come from the
  // counter: the wave that draws n_items or more first signals `drained` -- from here on only reads already in
  // flight are left, and the host may let the next batch's kernel in.
  const u64 dealt = gridDim.x;
  auto next_read = [&]() -> u64 {
    unsigned long long v = 0;
    if (lane == 0) {
      v = atomicAdd(a.next_read, 1ull) + dealt;
      if (v >= n_items && a.drained) __hip_atomic_store(a.drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return (static_cast<u64>(static_cast<u32>(uni(static_cast<int>(v >> 32)))) << 32) |
           static_cast<u32>(uni(static_cast<int>(v)));
  };
  u64 r_next = blockIdx.x;
  while (r_next < n_items) {
    const u64 slot = r_next;
    const u64 r = a.order ? static_cast<u64>(a.order[r_next]) : r_next;
    r_next = next_read();  // fetched early; its latency hides under this read's work
    __builtin_amdgcn_s_setprio(0);  // (a heavy read raises its wave's priority: seed_pass)
    long long t_read = 0;
    if (TIMED) t_read = clock64();
    bool rd_overflow = false;  // THIS read's CIGAR found no room (the launch's flag `overflow` is sticky)
    const u32 L = a.lens[r];
    Hit best;
    best.diffs = 0x7fff; best.flags = 0; best.pos = 0;
    u32 n_ops = 0;
    u32 *cig_out = a.cig + r * a.cig_stride;
    if (L > kMaxReadLen) too_long = true;
    // (a read beyond this launch's length is the long-read launch's: it overwrites what is stored for it here)
    if (L >= a.ix.min_len && L <= a.max_len) {
      // stage the four encodings and derive their 2-letter bit strings
      const u64 *src = a.packed + (LONG ? slot : r) * 4 * a.W;
      for (u32 k = lane; k < 4 * a.W; k += 64) lds.qpk[k] = src[k];
      wave_sync();
      for (u32 e = 0; e < 4; ++e)
        for (u32 wb = 0; wb < a.WB; ++wb) {
          const u32 j = wb * 64 + lane;
          const bool b = j < L ? bit2(q_nibble(lds.qpk + e * a.W, j)) : true;
          const u64 word = __ballot(b);
          if (lane == 0) lds.qbits[e * a.WB + wb] = word;
        }
      if constexpr (COOP) build_qmasks(lds, L);
      wave_sync();
      if (L < max(a.ix.window, L >> 1) + kKeyWeight - 1)  // 44-46 bases: seeds reach past the end of the read
        ghost_bits(a.packed, a.lens, r, L, a.max_len, a.ix.min_len, a.W, a.WB, lds.qbits);

      SeSet S;
      S.begin_read(L);
      for (u32 cidx = 0; cidx < n_calls; ++cidx) {
        const bool rc = (call_rc >> cidx) & 1u, ar = (call_ar >> cidx) & 1u;
        const bool g_to_a = rc != ar;  // get_conv_type, src/abismal.cpp:1261-1267
        const u32 enc = (rc ? 2u : 0u) + (g_to_a ? 1u : 0u);
        const u32 flags = (rc ? kFlagRC : 0u) | (ar ? kFlagARich : 0u);
        S.cutoff = S.good_cutoff;  // set_specific
        seed_pass<true, TIMED, COOP, REC>(a.ix, lds, enc, g_to_a, flags, L, S, wt, seg_epoch);
        // should_do_sensitive, :367-370
        if (S.sz != static_cast<int>(kSeCap) || S.cutoff > S.good_cutoff) {
          S.cutoff = S.top_d();  // set_sensitive
          seed_pass<false, TIMED, COOP, REC>(a.ix, lds, enc, g_to_a, flags, L, S, wt, seg_epoch);
        }
      }
      ABM_STAMP(t_a);
      choose_se<LONG>(a.ix, lds, L, a.valid_frac, S, best, cig_out, sink, n_ops, rd_overflow, n_aln, n_single);  // (n_aln, n_single: dead unless TIMED)
      overflow |= rd_overflow;
      ABM_STAMP(t_b);
      if (TIMED) wt.t_align += t_b - t_a;
    }
    if constexpr (!LONG) {
      if (a.sam_tail != nullptr) {
        // (the line buffer: the traceback table's place -- window slots 1.., the window cache and the table's extra bytes --
        // idle once the CIGAR is out; launch_map_se asks for at least sam_stride bytes there)
        const u32 len = (L > a.max_len) ? 0xFFFFFFFFu : format_sam_tail(a, lds.tb, lds.jpos, r, (L >= a.ix.min_len) ? L : 0u, best, best.pos != 0 ? n_ops : 0u);
        if (lane == 0) store_out(a.sam_len + r, len);
      }
    }
    if (lane == 0) {
      store_out(a.res + r, best);
      store_out(a.cig_n + r, best.pos != 0 ? n_ops : 0u);
      if (TIMED && a.read_cycles) a.read_cycles[r] = static_cast<u32>((clock64() - t_read) >> 10);
    }
    if (!LONG && a.slice_left != nullptr) {
      // results leave slice by slice: everything this read wrote (hit, count, slot, arena entries) was written through
      // to memory (store_out), and the wave waits for those stores to be acknowledged before its slice's count goes
      // down (a system-scope fence per read instead writes back and invalidates the L2 each time: the kernel took
      // 1.7 times as long).
      // The wave that takes a count to zero fences at system scope once, then tells the host.  A read whose CIGAR
      // found no room in the arena never counts: its slice stays open and the host maps the batch again.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every store of this wave acknowledged)
      if (lane == 0 && !rd_overflow) {
        const u32 sl = a.slice_id[r];
        if (sl != 0xFFFFu && atomicSub(&a.slice_left[sl], 1u) == 1u) {
          __threadfence_system();
          __hip_atomic_store(a.slice_done + sl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  if (TIMED && a.work) {  // exact work tallies for the roofline model: kept by the diagnostic build only (see seed_pass)
    auto wsum = [&](u32 v) { u32 t; (void)wave_excl_sum(v, t); return t; };
    const u32 s0 = wsum(wt.seed_iters), s1 = wsum(wt.probes), s2 = wsum(wt.cands), s3 = wsum(wt.words);
    const u32 wsum_hits = wsum(wt.cache_hits);
    if (lane == 0) {
      atomicAdd(&a.work[0], static_cast<unsigned long long>(s0));
      atomicAdd(&a.work[1], static_cast<unsigned long long>(s1));
      atomicAdd(&a.work[2], static_cast<unsigned long long>(s2));
      atomicAdd(&a.work[3], static_cast<unsigned long long>(s3));
      atomicAdd(&a.work[4], static_cast<unsigned long long>(wt.updates));
      atomicAdd(&a.work[5], static_cast<unsigned long long>(n_aln));
      atomicAdd(&a.work[11], static_cast<unsigned long long>(wsum_hits));
      atomicAdd(&a.work[12], static_cast<unsigned long long>(n_single));  // reads whose set held one alignable entry
      if (TIMED) {
        atomicAdd(&a.work[6], static_cast<unsigned long long>(wt.t_probe));
        atomicAdd(&a.work[7], static_cast<unsigned long long>(wt.t_stream));
        atomicAdd(&a.work[8], static_cast<unsigned long long>(wt.t_replay));
        atomicAdd(&a.work[9], static_cast<unsigned long long>(wt.t_align));
        atomicAdd(&a.work[10], static_cast<unsigned long long>(phase_stamp() - t_begin));
        atomicAdd(&a.work[13], static_cast<unsigned long long>(wt.light_steps));  // filter steps of at most 64 candidates
        atomicAdd(&a.work[14], static_cast<unsigned long long>(wt.fifo_updates));
        atomicAdd(&a.work[15], static_cast<unsigned long long>(wt.steps));
      }
    }
  }
  if (lane == 0 && (overflow || too_long))
    atomicOr(a.status, (overflow ? 1u : 0u) | (too_long ? 2u : 0u));
  if (a.host_tail != nullptr && lane == 0) {  // the last wave to get here publishes the launch's two summary words
    __threadfence();
    const u32 before = atomicAdd(a.finished, 1u);
    if (before + 1u == gridDim.x) {
      __threadfence();
      a.host_tail[0] = __hip_atomic_load(a.cig_arena_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.host_tail[1] = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// (the launch bound's second argument is waves per SIMD: 5 x 4 SIMDs = 20 one-wave workgroups per CU)
// REC: the filter reads the window records (DevIndex::wrec) -- a launch none of whose reads is longer than they serve
template <bool TIMED, bool COOP, bool REC = false>
__global__ __launch_bounds__(64, ABM_SE_WAVES_PER_SIMD) void map_se_kernel(SeArgs a) { map_se_body<TIMED, COOP, false, REC>(a); }
__global__ __launch_bounds__(64, 1) void map_se_long_kernel(SeArgs a) { map_se_body<false, false, true>(a); }

// reads of this batch that the long-read launch takes: kLdsReadLen < length <= kMaxReadLen
__global__ __launch_bounds__(256) void collect_long_kernel(const u32 *__restrict__ lens, u64 n, u32 *__restrict__ list,
                                                           u32 *__restrict__ count) {
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n && lens[r] > kLdsReadLen && lens[r] <= kMaxReadLen) list[atomicAdd(count, 1u)] = static_cast<u32>(r);
}
// pack_reads_kernel for listed reads: packed[slot][enc][word], slot = position in the list
__global__ __launch_bounds__(256) void pack_listed_kernel(const char *__restrict__ blob, const u64 *__restrict__ off,
                                                          const u32 *__restrict__ list, u64 m, u32 W, u64 *__restrict__ packed) {
  const u64 gid = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  const u64 slot = gid / W;
  const u32 w = static_cast<u32>(gid % W);
  if (slot >= m) return;
  const u64 r = list[slot];
  const u64 b = off[r];
  const u32 L = static_cast<u32>(off[r + 1] - b);
  u64 ft = 0, fa = 0, rt = 0, ra = 0;
  for (u32 j = 0; j < 16; ++j) {
    const u32 k = w * 16 + j;
    u32 nft = 15, nfa = 15, nrt = 15, nra = 15;
    if (k < L) {
      const u32 cf = static_cast<u8>(blob[b + k]);
      const u32 cr = comp_base(static_cast<u8>(blob[b + (L - 1 - k)]));
      nft = read_nibble(cf, false); nfa = read_nibble(cf, true);
      nrt = read_nibble(cr, false); nra = read_nibble(cr, true);
    }
    const u32 sh = j << 2;
    ft |= static_cast<u64>(nft) << sh; fa |= static_cast<u64>(nfa) << sh;
    rt |= static_cast<u64>(nrt) << sh; ra |= static_cast<u64>(nra) << sh;
  }
  if (w * 16 >= L) ft = fa = rt = ra = ~0ull;
  u64 *dst = packed + (slot * 4) * W + w;
  dst[0] = ft; dst[W] = fa; dst[2 * W] = rt; dst[3 * W] = ra;
}

// =============================================================================
// Heaviest-first ordering.  Work per read spans four orders of magnitude and is
// predicted well by how full the read's first seed buckets are, so reads are
// binned by log2 of that occupancy and handed out from the heaviest bin down
// (longest-processing-time-first): a wave that draws a monster draws it early.
// Results are indexed by read, so the processing order never shows in the output.
// =============================================================================
__device__ __forceinline__ u32 weight_class(const DevIndex &ix, const u64 *pk, u32 L, bool g_to_a) {
  // same keys as seed_pass at offset 0 (first 25 / 16 bases)
  u32 k2 = 0, k3 = 0;
  const u64 w0 = pk[0], w1 = L > 16 ? pk[1] : 0ull;
  for (u32 j = 0; j < kKeyWeight; ++j) {
    const u32 nb = static_cast<u32>((j < 16 ? w0 >> (j << 2) : w1 >> ((j - 16) << 2))) & 15u;
    k2 = (k2 << 1) | bit2(nb);
  }
  for (u32 j = 0; j < kKeyWeight3; ++j)
    k3 = k3 * 3u + trit(static_cast<u32>(w0 >> (j << 2)) & 15u, g_to_a);
  const u32 *cnt3 = g_to_a ? ix.counter_a : ix.counter_t;
  const u32 occ = (ix.counter[k2 + 1] - ix.counter[k2]) + (cnt3[k3 + 1] - cnt3[k3]);
  return 32u - static_cast<u32>(__clz(static_cast<int>(occ | 1u)));  // 1..32
}

__global__ __launch_bounds__(256) void weigh_reads_kernel(DevIndex ix, const u64 *__restrict__ packed,
                                                          const u32 *__restrict__ lens, u64 n, u32 W,
                                                          int mode, u8 *__restrict__ cls,
                                                          u32 *__restrict__ class_count) {
  __shared__ u32 hist[33];
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n) {
    const u32 L = lens[r];
    u32 c = 0;
    if (L >= ix.min_len) {
      const u64 *pk = packed + r * 4 * W;
      // forward call of the mode and its reverse-strand partner
      const bool ar = mode == 1;
      c = max(weight_class(ix, pk + (ar ? 1 : 0) * W, L, ar), weight_class(ix, pk + (2 + (ar ? 0 : 1)) * W, L, !ar));
    }
    cls[r] = static_cast<u8>(c);
    atomicAdd(&hist[c], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 33 && hist[threadIdx.x]) atomicAdd(&class_count[threadIdx.x], hist[threadIdx.x]);
}

__global__ void order_bases_kernel(u32 *class_count /*[33] in: counts, out: start of each class (heaviest first)*/) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    u32 at = 0;
    for (int c = 32; c >= 0; --c) { const u32 k = class_count[c]; class_count[c] = at; at += k; }
  }
}

__global__ __launch_bounds__(256) void order_scatter_kernel(const u8 *__restrict__ cls, u64 n,
                                                            u32 *__restrict__ class_cursor,
                                                            u32 *__restrict__ order) {
  // one global reservation per (block, class); ranks inside the block come from LDS
  __shared__ u32 hist[33], base[33];
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  u32 c = 0, rank = 0;
  if (r < n) {
    c = cls[r];
    rank = atomicAdd(&hist[c], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 33 && hist[threadIdx.x]) base[threadIdx.x] = atomicAdd(&class_cursor[threadIdx.x], hist[threadIdx.x]);
  __syncthreads();
  if (r < n) order[base[c] + rank] = static_cast<u32>(r);
}

hipError_t launch_order_reads(const DevIndex &ix, const u64 *d_packed, const u32 *d_lens, u64 n, u32 W,
                              int mode, u8 *d_cls, u32 *d_class33, u32 *d_order, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_class33, 0, 33 * sizeof(u32), st);
  if (e != hipSuccess) return e;
  const u32 blocks = static_cast<u32>((n + 255) / 256);
  hipLaunchKernelGGL(weigh_reads_kernel, dim3(blocks), dim3(256), 0, st, ix, d_packed, d_lens, n, W, mode, d_cls, d_class33);
  hipLaunchKernelGGL(order_bases_kernel, dim3(1), dim3(64), 0, st, d_class33);
  hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, st, d_cls, n, d_class33, d_order);
  return hipGetLastError();
}

// ---- the order of a batch whose results leave slice by slice -----------------------------
// Heaviest-first over the whole batch would finish every slice at the end of the launch.  Here only the few heaviest
// reads go first -- the classes from the top that together hold at most 1/32 of the batch: they are what a launch's
// tail is made of -- then the reads before the first slice (a batch's lead-in), then slice after slice in input
// order, so that slices complete one after the other while the kernel runs.
// hist: rows [n_slices + 1][33] (row n_slices = the lead-in), cursors [33] + [n_slices + 1], then the threshold class.
__device__ __forceinline__ u32 slice_row(const u32 *__restrict__ first, u32 n_slices, u32 r) {
  if (r < first[0]) return n_slices;
  u32 lo = 0, hi = n_slices;  // first[lo] <= r < first[hi]
  while (hi - lo > 1) {
    const u32 mid = (lo + hi) >> 1;
    if (first[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void weigh_sliced_kernel(DevIndex ix, const u64 *__restrict__ packed, const u32 *__restrict__ lens,
                                                           u64 n, u32 W, int mode, u8 *__restrict__ cls,
                                                           const u32 *__restrict__ slice_first, u32 n_slices,
                                                           u16 *__restrict__ slice_id, u32 *__restrict__ hist) {
  __shared__ u32 local[4][33];
  __shared__ u32 row0;
  for (u32 k = threadIdx.x; k < 4 * 33; k += blockDim.x) (&local[0][0])[k] = 0;
  const u64 r_first = static_cast<u64>(blockIdx.x) * blockDim.x;
  if (threadIdx.x == 0) row0 = slice_row(slice_first, n_slices, static_cast<u32>(r_first));
  __syncthreads();
  const u64 r = r_first + threadIdx.x;
  if (r < n) {
    const u32 L = lens[r];
    u32 c = 0;
    if (L >= ix.min_len) {
      const u64 *pk = packed + r * 4 * W;
      const bool ar = mode == 1;
      c = max(weight_class(ix, pk + (ar ? 1 : 0) * W, L, ar), weight_class(ix, pk + (2 + (ar ? 0 : 1)) * W, L, !ar));
    }
    cls[r] = static_cast<u8>(c);
    const u32 row = slice_row(slice_first, n_slices, static_cast<u32>(r));
    slice_id[r] = row < n_slices ? static_cast<u16>(row) : static_cast<u16>(0xFFFFu);
    const u32 rel = row - row0;
    if (rel < 4u) atomicAdd(&local[rel][c], 1u);
    else atomicAdd(&hist[static_cast<size_t>(row) * 33 + c], 1u);
  }
  __syncthreads();
  for (u32 k = threadIdx.x; k < 4 * 33; k += blockDim.x) {
    const u32 v = (&local[0][0])[k], row = row0 + k / 33;
    if (v && row <= n_slices) atomicAdd(&hist[static_cast<size_t>(row) * 33 + k % 33], v);
  }
}

__global__ __launch_bounds__(256) void order_bases_sliced_kernel(u32 *__restrict__ hist, u32 n_slices, u64 n,
                                                                 u32 *__restrict__ slice_left) {
  __shared__ u32 tot[33];
  __shared__ u32 threshold, heavy_total;
  u32 *cur_heavy = hist + static_cast<size_t>(n_slices + 1) * 33, *cur_slice = cur_heavy + 33, *t_out = cur_slice + n_slices + 1;
  if (threadIdx.x < 33) {
    u32 sum = 0;
    for (u32 row = 0; row <= n_slices; ++row) sum += hist[static_cast<size_t>(row) * 33 + threadIdx.x];
    tot[threadIdx.x] = sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 at = 0, T = 33;
    for (int c = 32; c >= 1; --c) {
      if (at + tot[c] > n / 32) break;
      at += tot[c];
      T = static_cast<u32>(c);
    }
    u32 run = 0;
    for (int c = 32; c >= static_cast<int>(T); --c) { cur_heavy[c] = run; run += tot[c]; }
    heavy_total = run;
    threshold = T;
    *t_out = T;
  }
  __syncthreads();
  const u32 T = threshold;
  for (u32 row = threadIdx.x; row <= n_slices; row += blockDim.x) {
    u32 light = 0, all = 0;
    for (u32 c = 0; c < 33; ++c) {
      const u32 v = hist[static_cast<size_t>(row) * 33 + c];
      all += v;
      if (c < T) light += v;
    }
    cur_slice[row] = light;
    if (row < n_slices) slice_left[row] = all;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 at = heavy_total;
    u32 v = cur_slice[n_slices];
    cur_slice[n_slices] = at;
    at += v;
    for (u32 row = 0; row < n_slices; ++row) { v = cur_slice[row]; cur_slice[row] = at; at += v; }
  }
}

__global__ __launch_bounds__(256) void order_scatter_sliced_kernel(const u8 *__restrict__ cls, const u16 *__restrict__ slice_id, u64 n,
                                                                   u32 n_slices, u32 *__restrict__ hist, u32 *__restrict__ order) {
  // one global reservation per (block, bin): bins = the 33 classes (heavy reads) and the block's first four rows
  __shared__ u32 cnt[37], base[37];
  __shared__ u32 row0;
  u32 *cur_heavy = hist + static_cast<size_t>(n_slices + 1) * 33, *cur_slice = cur_heavy + 33;
  const u32 T = cur_slice[n_slices + 1];
  if (threadIdx.x < 37) cnt[threadIdx.x] = 0;
  const u64 r_first = static_cast<u64>(blockIdx.x) * blockDim.x;
  if (threadIdx.x == 0) { const u16 s0 = slice_id[r_first]; row0 = s0 == 0xFFFFu ? n_slices : s0; }
  __syncthreads();
  const u64 r = r_first + threadIdx.x;
  u32 bin = 0xFFFFFFFFu, rank = 0, row = 0;
  if (r < n) {
    const u32 c = cls[r];
    const u16 s = slice_id[r];
    row = s == 0xFFFFu ? n_slices : s;
    if (c >= T) bin = c;
    else if (row - row0 < 4u) bin = 33 + (row - row0);
    if (bin != 0xFFFFFFFFu) rank = atomicAdd(&cnt[bin], 1u);
    else order[atomicAdd(&cur_slice[row], 1u)] = static_cast<u32>(r);
  }
  __syncthreads();
  if (threadIdx.x < 37 && cnt[threadIdx.x]) {
    const u32 b = threadIdx.x;
    base[b] = b < 33 ? atomicAdd(&cur_heavy[b], cnt[b]) : (row0 + (b - 33) <= n_slices ? atomicAdd(&cur_slice[row0 + (b - 33)], cnt[b]) : 0u);
  }
  __syncthreads();
  if (r < n && bin != 0xFFFFFFFFu) order[base[bin] + rank] = static_cast<u32>(r);
}

hipError_t launch_order_reads_sliced(const DevIndex &ix, const u64 *d_packed, const u32 *d_lens, u64 n, u32 W, int mode,
                                     u8 *d_cls, const u32 *d_slice_first, u32 n_slices, u16 *d_slice_id, u32 *d_hist,
                                     u32 *d_slice_left, u32 *d_order, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_hist, 0, order_sliced_hist_words(n_slices) * sizeof(u32), st);
  if (e != hipSuccess) return e;
  const u32 blocks = static_cast<u32>((n + 255) / 256);
  hipLaunchKernelGGL(weigh_sliced_kernel, dim3(blocks), dim3(256), 0, st, ix, d_packed, d_lens, n, W, mode, d_cls, d_slice_first,
                     n_slices, d_slice_id, d_hist);
  hipLaunchKernelGGL(order_bases_sliced_kernel, dim3(1), dim3(256), 0, st, d_hist, n_slices, n, d_slice_left);
  hipLaunchKernelGGL(order_scatter_sliced_kernel, dim3(blocks), dim3(256), 0, st, d_cls, d_slice_id, n, n_slices, d_hist, d_order);
  return hipGetLastError();
}

// CIGAR compaction for the host entry points: fixed slots -> one blob + n+1 offsets
__global__ __launch_bounds__(256) void cigar_counts_kernel(const Hit *__restrict__ res, const u32 *__restrict__ cig_n,
                                                           u64 n, u32 stride, unsigned long long *__restrict__ cnt) {
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  // the true op count even where the slot held fewer: the caller patches those reads in afterwards
  if (r < n) cnt[r] = (res == nullptr || res[r].pos != 0) ? cig_n[r] : 0u;
  if (r == n) cnt[r] = 0;
}
__global__ __launch_bounds__(256) void cigar_gather_kernel(const u32 *__restrict__ cig, u32 stride,
                                                           const unsigned long long *__restrict__ off, u64 n,
                                                           u32 *__restrict__ blob) {
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const unsigned long long a = off[r], b = min(off[r + 1], off[r] + stride);
  for (unsigned long long k = a; k < b; ++k) blob[k] = cig[r * stride + (k - a)];
}

// counts -> exclusive offsets (n+1 entries, in place) -> blob.  `tmp` is scan scratch of *tmp_bytes.
hipError_t launch_compact_cigars(const Hit *d_res, const u32 *d_cig, const u32 *d_cig_n, u64 n, u32 stride,
                                 unsigned long long *d_off /*[n+1]*/, u32 *d_blob, void *tmp, size_t *tmp_bytes,
                                 hipStream_t st) {
  if (tmp == nullptr) {
    return hipcub::DeviceScan::ExclusiveSum(nullptr, *tmp_bytes, d_off, d_off, static_cast<int>(n + 1), st);
  }
  const u32 blocks = static_cast<u32>((n + 1 + 255) / 256);
  hipLaunchKernelGGL(cigar_counts_kernel, dim3(blocks), dim3(256), 0, st, d_res, d_cig_n, n, stride, d_off);
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, *tmp_bytes, d_off, d_off, static_cast<int>(n + 1), st);
  if (e != hipSuccess) return e;
  if (d_blob) hipLaunchKernelGGL(cigar_gather_kernel, dim3(blocks), dim3(256), 0, st, d_cig, stride, d_off, n, d_blob);
  return hipGetLastError();
}

hipError_t launch_gather_cigars(const u32 *d_cig, u32 stride, const unsigned long long *d_off, u64 n, u32 *d_blob,
                                hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(cigar_gather_kernel, dim3(static_cast<u32>((n + 255) / 256)), dim3(256), 0, st, d_cig, stride, d_off, n, d_blob);
  return hipGetLastError();
}

// ---- launchers ----------------------------------------------------------------
u32 se_window_words(u32 max_len, double valid_frac) {
  const int md = static_cast<i16>(valid_frac * max_len);
  int bw = 2 * md + 1;
  if (bw > static_cast<int>(kMaxBand) || bw < 1) bw = kMaxBand;
  return ((max_len + bw + 15 + 15) >> 4) + 1;
}

u32 tb_extra_bytes(u32 GW, u32 max_len, double valid_frac) {
  const int md = static_cast<i16>(valid_frac * max_len);
  int bw = 2 * md + 1;
  if (bw > static_cast<int>(kMaxBand) || bw < 0) bw = kMaxBand;
  if (bw < 1) bw = 1;
  const size_t need = static_cast<size_t>(max_len + bw) * bw;
  const size_t have = static_cast<size_t>(kMaxJobs - 1) * GW * 8 + (static_cast<size_t>(8) << kPosCacheBits);
  return need > have ? static_cast<u32>((need - have + 7) & ~static_cast<size_t>(7)) : 0u;
}

size_t se_lds_bytes(u32 W, u32 WB, u32 cig_stride, u32 max_len, double valid_frac) {
  const u32 GW = se_window_words(max_len, valid_frac);
  const u32 MB = (max_len + kPlaneBlock - 1) / kPlaneBlock;
  size_t b = static_cast<size_t>(4) * W * 8 + static_cast<size_t>(4) * WB * 8 + static_cast<size_t>(4) * MB * 4 * 8 +
             (static_cast<size_t>(8) << kPosCacheBits) +
             static_cast<size_t>(kMaxJobs) * GW * 8 + static_cast<size_t>((cig_stride + 1) & ~1u) * 4 +
             2 * kSeCap * 4 + 64 * 4 + 2 * 128 * 4 + 64 * 2;
  b += tb_extra_bytes(GW, max_len, valid_frac);
  return (b + 15) & ~static_cast<size_t>(15);
}

int se_resident_waves(u32 W, u32 WB, u32 cig_stride, u32 max_len, double valid_frac) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  const size_t lds = se_lds_bytes(W, WB, cig_stride, max_len, valid_frac);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_se_kernel<false, true>, 64, lds) != hipSuccess) return 0;
  // Measured on MI355X at hg38 scale (round 2: scripts/grid_sweep.sh and se_variant.sh, in the history of this repository): what matters is waves without
  // register spills.  10 M reads: one lane per window, 20 waves/CU 1405 ms (28: 1577, 32: 1681); cooperative
  // window loads with 8 rounds in flight, 16 waves at 128 registers 1029 ms (20 waves at 96 with spills:
  // 1386); with 2 rounds in flight 20 waves fit almost without spills: 923 ms (24 waves: 999).
  constexpr int kSeWavesPerCu = 4 * ABM_SE_WAVES_PER_SIMD;
  return min(per_cu, kSeWavesPerCu) * prop.multiProcessorCount;
}

hipError_t launch_make_planes(const u64 *d_genome, u64 n_words, u64 n_bases, u64 n_blocks, u64 *d_planes0,
                              u64 *d_planes1, u32 *d_nmap, u32 *d_bad, hipStream_t st) {
  if (n_blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(make_planes_kernel, dim3(static_cast<u32>((n_blocks + 255) / 256)), dim3(256), 0, st, d_genome,
                     n_words, n_bases, n_blocks, d_planes0, d_planes1, d_nmap, d_bad);
  return hipGetLastError();
}

hipError_t launch_pack_reads(const char *d_blob, const u64 *d_off, u64 n, u32 W, u64 *d_packed,
                             u32 *d_lens, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const u64 threads = n * W;
  const u32 blocks = static_cast<u32>((threads + 255) / 256);
  hipLaunchKernelGGL(pack_reads_kernel, dim3(blocks), dim3(256), 0, st, d_blob, d_off, n, W, d_packed, d_lens);
  return hipGetLastError();
}

size_t se_long_lds_bytes(u32 W, u32 WB, u32 GW) {
  const size_t b = static_cast<size_t>(4) * W * 8 + static_cast<size_t>(4) * WB * 8 + 2 * kSeCap * 4 + static_cast<size_t>(2) * GW * 8 +
                   (static_cast<size_t>(8) << kPosCacheBits) + 64 * 4 + 2 * 128 * 4 + 64 * 2;
  return (b + 15) & ~static_cast<size_t>(15);
}
size_t se_long_tb_bytes(u32 max_len) { return (static_cast<size_t>(max_len + kMaxBand) * kMaxBand + 255) & ~static_cast<size_t>(255); }
int se_long_resident_waves(u32 W, u32 WB, u32 GW) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_se_long_kernel, 64, se_long_lds_bytes(W, WB, GW)) != hipSuccess) return 0;
  return min(per_cu, 4) * prop.multiProcessorCount;
}
hipError_t launch_collect_long(const u32 *d_lens, u64 n, u32 *d_list, u32 *d_count, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(collect_long_kernel, dim3(static_cast<u32>((n + 255) / 256)), dim3(256), 0, st, d_lens, n, d_list, d_count);
  return hipGetLastError();
}
hipError_t launch_pack_listed(const char *d_blob, const u64 *d_off, const u32 *d_list, u64 m, u32 W, u64 *d_packed, hipStream_t st) {
  if (m == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_listed_kernel, dim3(static_cast<u32>((m * W + 255) / 256)), dim3(256), 0, st, d_blob, d_off, d_list, m, W, d_packed);
  return hipGetLastError();
}
hipError_t launch_map_se_long(SeArgs a, u32 n_waves, hipStream_t st) {
  if (a.n_reads == 0) return hipSuccess;
  const size_t lds = se_long_lds_bytes(a.W, a.WB, a.GW);
  const u32 blocks = static_cast<u32>(a.n_reads < n_waves ? a.n_reads : n_waves);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(map_se_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipLaunchKernelGGL(map_se_long_kernel, dim3(blocks), dim3(64), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_map_se(SeArgs a, u32 max_len, u32 n_waves, bool timed, hipStream_t st) {
  if (a.n_reads == 0) return hipSuccess;
  const size_t lds = se_lds_bytes(a.W, a.WB, a.ctmp_cap, max_len, a.size_frac);
  const u32 blocks = static_cast<u32>(a.n_reads < n_waves ? a.n_reads : n_waves);
  // COOP: lanes share a candidate's window on the bit planes (a.G != 0); otherwise one lane per window
  if (a.G != 0 && a.ix.wrec != nullptr && max_len <= a.ix.wrec_max_len && (a.G == 2 || a.G == 4)) {
    if (timed) hipLaunchKernelGGL((map_se_kernel<true, true, true>), dim3(blocks), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_se_kernel<false, true, true>), dim3(blocks), dim3(64), lds, st, a);
  }
  else if (a.G != 0) {
    if (timed) hipLaunchKernelGGL((map_se_kernel<true, true>), dim3(blocks), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_se_kernel<false, true>), dim3(blocks), dim3(64), lds, st, a);
  }
  else {
    if (timed) hipLaunchKernelGGL((map_se_kernel<true, false>), dim3(blocks), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_se_kernel<false, false>), dim3(blocks), dim3(64), lds, st, a);
  }
  return hipGetLastError();
}

}  // namespace abm
