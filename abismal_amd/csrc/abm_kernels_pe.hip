// abismal_amd HIP kernels for gfx950: paired-end mapping, one wavefront per pair.
// Restates map_paired_ended / map_paired_ended_rand's per-pair body
// (src/abismal.cpp:1950-1999, :2094-2155): two or four orientation calls
// (map_fragments :1849-1885), each seeding both ends into growable candidate
// sets (pe_candidates :775-863), mating them (best_pair :1722-1831), feeding the
// single-end sets (best_single :1715-1720), then valid_pair and the single-end
// fallback.  The per-pair body is split by phase where its register pressure
// is (round 5): a SEED kernel (both seed passes of every orientation call's two
// ends; shaped like the single-end kernel: no alignment state, small LDS) hands
// the finished candidate lists over in global memory, and MATE kernels take
// them from there (sort, scoring, mating, tracebacks, best_single, fallback) --
// lists of up to kPeTier1Cap entries in LDS, longer ones in global memory.
// Pairs whose sets outgrow the seed kernel (its staging area, or the heap of a
// sensitive pass) go through the WHOLE-pair kernel: the same code unsplit, one
// wave per pair with 32768-entry sets in global memory (tier 2).
#include <climits>
#include "abm_kernels_core.hpp"
#include "abm_pe_set.hpp"

// Waves per SIMD the pair kernels are compiled for: 4 (128 registers per lane), and for the production kernels on the
// bit planes also 3 (170 registers): with 2x150-base pairs the LDS of a wave allows 13 waves per CU anyway, and the
// build with more registers is the faster one there; with 2x100 the LDS allows 16 and four waves per SIMD win
// (profiles/r03_exp_pe_loop_and_registers.log, r03_exp_pe_2x100_variants.log).  pe_waves_per_simd() picks per launch.
#ifndef ABM_PE_WAVES_PER_SIMD
#define ABM_PE_WAVES_PER_SIMD 4
#endif

namespace abm {

// pe_element, src/abismal.cpp:547-619 (wave-uniform)
struct PairBest {
  int aln_score, max_aln_score;
  int d1, d2;
  u32 f1, f2, p1, p2;
  __device__ __forceinline__ void clear() { aln_score = 0; p1 = 0; d1 = 0x7fff; p2 = 0; d2 = 0x7fff; }
  __device__ __forceinline__ void clear(u32 l1, u32 l2) {
    aln_score = 0;
    p1 = 0; d1 = static_cast<i16>(0.4 * l1);
    p2 = 0; d2 = static_cast<i16>(0.4 * l2);
    max_aln_score = static_cast<i16>(static_cast<i16>(2 * l1) + static_cast<i16>(2 * l2));
  }
  __device__ __forceinline__ bool ambig() const { return f1 & kFlagAmbig; }
  __device__ __forceinline__ bool sure_ambig() const { return ambig() && aln_score == max_aln_score; }
  __device__ __forceinline__ bool should_report(bool allow_ambig) const {
    return p1 != 0 && (allow_ambig || !ambig());
  }
  // pe_element::update, :570-587 (s1 = read 1's hit, s2 = read 2's)
  __device__ __forceinline__ bool offer(int scr, int sd1, u32 sf1, u32 sp1, int sd2, u32 sf2, u32 sp2) {
    const int have = d1 + d2, got = sd1 + sd2;
    if (scr > aln_score || (scr == aln_score && got < have)) {
      d1 = sd1; f1 = sf1; p1 = sp1; d2 = sd2; f2 = sf2; p2 = sp2; aln_score = scr;
      return true;
    }
    if (scr == aln_score && got == have) f1 |= kFlagAmbig;
    return false;
  }
};

// COOP: the seed passes filter on the genome's bit planes with cooperative window loads (hamming_planes), as the
// single-end kernel does; otherwise one lane per window on the nibble array (genomes with IUPAC letters, reads > 448)
// LONG: the launch for pairs with an end beyond kLdsReadLen bases (up to kMaxReadLen), as map_se_long_kernel is for
// single-end reads: both ends' packed encodings and bit strings lie in a per-wave piece of global memory (two ends of
// 32766 bases are 164 KB: more than a CU's LDS), so do every traceback table ((L + 61) x 61 bytes) and the CIGAR scratch;
// bands are up to 61 lanes wide (two window slots), the filter runs on the nibble array, scores wrap at 16 bits like the
// reference's score_t, and the lists are tier 2's (global memory).
// PHASE: kWhole = seeding and mating in one kernel (tier 2's whole-pair launch, the long-end launch; BIG = false is the
// unsplit tier 1 of rounds 1-4, kept for same-box comparisons), kSeed / kMate = the two halves of the split
enum : int { kWhole = 0, kSeed = 1, kMate = 2 };
#ifndef ABM_PE_RADIX_SORT_MIN
#define ABM_PE_RADIX_SORT_MIN 512
#endif
constexpr int kRadixSortMin = ABM_PE_RADIX_SORT_MIN;  // tier 2: lists longer than this are sorted by radix passes, shorter ones by the bitonic network
template <bool BIG, bool COOP, bool LONG = false, int PHASE = kWhole, bool REC = false> struct PeWave {
  const PeArgs &a;
  WaveLds lds;      // qpk/qbits point at end 0; end 1 follows at +4W / +4WB
  PeLds pl;
  WorkTally wt;
  u32 n_aln;
  u32 seg_epoch;    // tags the seed passes' segment marks (locate128)
  bool overflow, need_big;
  u32 L[2];
  SeSet se[2];
  PeSet P;
  // finished-set bookkeeping for the orientation call in flight (index 0 = endA)
  int lsz[2];
  bool worth[2], heap_order[2];
  u32 lflags[2];
  // CIGAR bookkeeping per read end
  u32 n_ops[2], ref_len[2];
  // diagnostic build only (TIMED): shader cycles per phase
  long long t_sort, t_score, t_mate, t_single;
  u32 *log_base;  // tier 2: this wave's best_single log
  int max_set;
  u32 *stage_pos;  // seed kernel: this wave's staging area for a list that outgrows its LDS slot
  i16 *stage_d;

  // (ends are run-time values -- the orientation calls are ONE piece of code looped over -- so the two-element register
  // arrays are read and written through selects, never indexed)
  __device__ __forceinline__ u32 len_of(int end) const { return end ? L[1] : L[0]; }
  __device__ __forceinline__ WaveLds lds_of(int end) const {
    WaveLds w = lds;
    w.qpk = lds.qpk + end * 4 * lds.W;
    w.qbits = lds.qbits + end * 4 * lds.WB;
    w.qmask = lds.qmask + end * 4 * lds.MB * 4;
    return w;
  }
  __device__ __forceinline__ u32 *cig_of(int end, u64 r) const { return (end ? a.cig2 : a.cig1) + r * a.cig_stride; }
  __device__ __forceinline__ CigarSink sink() const { return CigarSink{a.cig_stride, a.ctmp_cap, a.cig_arena, a.cig_arena_count, a.cig_arena_cap}; }

  // ---- one end of one orientation call: both seed passes, then freeze the set ----
  template <bool TIMED> __device__ __forceinline__ void seed_end(int which, int end, bool rc, bool ar) {
    const WaveLds w = lds_of(end);
    const bool g_to_a = rc != ar;
    const u32 enc = (rc ? 2u : 0u) + (g_to_a ? 1u : 0u);
    const u32 flags = (rc ? kFlagRC : 0u) | (ar ? kFlagARich : 0u);
    lflags[which] = flags;
    P.lpos = pl.lpos[which];
    P.ld = pl.ld[which];
    if constexpr (PHASE == kSeed) {  // (a list may outgrow its LDS slot: PeSet::append)
      P.cap_avail = a.cap;
      P.spill_pos = stage_pos; P.spill_d = stage_d; P.spill_cap = a.scap; P.spilled = false;
    }
    P.begin_read(len_of(end));
    if (len_of(end) >= a.ix.min_len) {
      P.cutoff = P.good_cutoff;  // set_specific
      seed_pass<true, TIMED, COOP, REC>(a.ix, w, enc, g_to_a, flags, len_of(end), P, wt, seg_epoch);
      if (!P.overflow && P.wants_sensitive()) {
        P.set_sensitive();
        seed_pass<false, TIMED, COOP, REC>(a.ix, w, enc, g_to_a, flags, len_of(end), P, wt, seg_epoch);
      }
    }
    need_big |= P.overflow;
    const int n = P.sz;
    wave_sync();
    if (P.heaped) freeze_heap_order(which, n);
    lsz[which] = n;
    max_set = max(max_set, n);
    heap_order[which] = P.heaped || n <= 1;
    worth[which] = n != static_cast<int>(kPeCapLarge) || P.cutoff != 0;  // should_align, :799-802
    wave_sync();
  }

  // the list in the heap's array order (what best_single replays if no mating happens, :1715-1720)
  __device__ __forceinline__ void freeze_heap_order(int which, int n) {
    const int lane = lane_id();
    u32 *lp = pl.lpos[which];
    i16 *ldv = pl.ld[which];
    if (n <= 64) {
      u32 e = 0, p = 0;
      if (lane < n) { e = P.heap[lane]; p = lp[e & 0x7FFFu]; }
      wave_sync();
      if (lane < n) { lp[lane] = p; ldv[lane] = static_cast<i16>(static_cast<int>(e) >> 16); }
    }
    else {  // through the per-wave scratch table
      for (int i = lane; i < n; i += 64) pl.tmp[i] = lp[P.heap[i] & 0x7FFFu];
      wave_sync();
      for (int i = lane; i < n; i += 64) {
        lp[i] = pl.tmp[i];
        ldv[i] = static_cast<i16>(static_cast<int>(P.heap[i]) >> 16);
      }
    }
    wave_sync();
  }

  // a list still in arrival order, put into the heap's array order after the fact (only needed
  // when the other end's set makes the reference skip the mating)
  __device__ __forceinline__ void heap_order_now(int which) {
    if (heap_order[which]) return;
    P.lpos = pl.lpos[which];
    P.ld = pl.ld[which];
    P.sz = lsz[which];
    P.heaped = false;
    P.heapify();
    freeze_heap_order(which, lsz[which]);
    heap_order[which] = true;
  }

  // ---- the hand-over between the seed kernel and the mate kernels (PeArgs::hand_*) ----
  __device__ __forceinline__ u32 *hand_slot(u64 r, int slot) const {
    const u32 n_slots = a.mode == 2 ? 8u : 4u;
    return a.hand_hdr + (r * n_slots + static_cast<u32>(slot)) * 2;
  }
  // seed kernel: the list seed_end(which) has just finished -- in P.lpos / P.ld: its LDS slot (frozen in heap order if
  // it was heaped) or the staging area -- goes to the hand-over area; no room there = the pair is the whole-pair kernel's
  __device__ __forceinline__ void emit_list(int which, u64 r, int slot) {
    if (need_big) return;
    const int lane = lane_id();
    const int n = lsz[which];
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(a.hand_count, static_cast<unsigned long long>(n));
    const u64 off = (static_cast<u64>(static_cast<u32>(uni(static_cast<int>(at >> 32)))) << 32) | static_cast<u32>(uni(static_cast<int>(at)));
    if (off + static_cast<u64>(n) > static_cast<u64>(a.hand_cap)) { need_big = true; return; }
    const u32 *lp = P.lpos;
    const i16 *ldv = P.ld;
    for (int i = lane; i < n; i += 64) { a.hand_pos[off + i] = lp[i]; a.hand_d[off + i] = ldv[i]; }
    if (lane == 0) {
      u32 *h = hand_slot(r, slot);
      h[0] = static_cast<u32>(off);
      h[1] = static_cast<u32>(n) | (worth[which] ? 1u << 16 : 0u) | (heap_order[which] ? 1u << 17 : 0u);
    }
    wave_sync();  // (the next seed_end writes the LDS slot / staging area this copy reads)
  }
  // mate kernels: the same list back into lists `which` of this wave (LDS, or tier 2's global memory)
  __device__ __forceinline__ void load_list(int which, u64 r, int slot, u32 flags) {
    const int lane = lane_id();
    const u32 *h = hand_slot(r, slot);
    const u32 off = static_cast<u32>(uni(static_cast<int>(h[0]))), meta = static_cast<u32>(uni(static_cast<int>(h[1])));
    const int n = static_cast<int>(meta & 0xFFFFu);
    lflags[which] = flags;
    lsz[which] = n;
    worth[which] = (meta >> 16) & 1u;
    heap_order[which] = (meta >> 17) & 1u;
    max_set = max(max_set, n);
    u32 *lp = pl.lpos[which];
    i16 *ldv = pl.ld[which];
    for (int i = lane; i < n; i += 64) { lp[i] = a.hand_pos[off + i]; ldv[i] = a.hand_d[off + i]; }
    wave_sync();
  }

  // prepare_for_mating (:844-852): sort by position, drop duplicates; diffs are
  // then recomputed (the Hamming distance is a function of the position alone)
  __device__ __forceinline__ void sort_unique(int which, int end) {
    const int lane = lane_id();
    const int n = lsz[which];
    u32 *buf = pl.heap;  // the live heap is dead by now
    bool sorted = false;
    if constexpr (BIG) {
      // Tier 2's long lists: a radix sort by position, four passes of eight bits between the two per-wave tables in
      // global memory (round 5).  The bitonic network below costs n log2(n)^2 / 128 compare-exchange steps of the wave --
      // 78 passes over a list of 4096 -- where this costs four: per pass a histogram of the digit in LDS (256 counters
      // in the place of the window slots, idle here), their prefix sums, and a scatter chunk by chunk in which the lanes
      // of a chunk that share a digit find their rank among themselves from eight ballots.  Equal positions are equal
      // entries (the duplicates unique() drops), so the order among them is of no consequence.
      if (n > kRadixSortMin) {
        u32 *src = pl.lpos[which], *dst = pl.tmp;
        u32 *hist = reinterpret_cast<u32 *>(lds.gwin);  // [256] running offsets of the digits
        for (u32 shift = 0; shift < 32; shift += 8) {
          for (int b = lane; b < 256; b += 64) hist[b] = 0;
          wave_sync();
          for (int i = lane; i < n; i += 64) atomicAdd(&hist[(src[i] >> shift) & 255u], 1u);
          wave_sync();
          {  // exclusive prefix sums of the 256 counters: four per lane, then across the lanes
            const u32 c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
            u32 total;
            const u32 base = wave_excl_sum(c0 + c1 + c2 + c3, total);
            wave_sync();
            hist[4 * lane] = base; hist[4 * lane + 1] = base + c0; hist[4 * lane + 2] = base + c0 + c1; hist[4 * lane + 3] = base + c0 + c1 + c2;
          }
          wave_sync();
          for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool have = i < n;
            const u32 v = have ? src[i] : 0u, d = (v >> shift) & 255u;
            u64 same = __ballot(have);
#pragma unroll
            for (u32 bit = 0; bit < 8; ++bit) {
              const u64 ones = __ballot((d >> bit) & 1u);
              same &= ((d >> bit) & 1u) ? ones : ~ones;
            }
            const u32 rank = static_cast<u32>(__popcll(same & ((1ull << lane) - 1)));
            const u32 at = have ? hist[d] : 0u;
            wave_sync();  // (every lane has read its digit's offset before the digit's first lane moves it on)
            if (have) {
              dst[at + rank] = v;
              if (rank == 0) hist[d] = at + static_cast<u32>(__popcll(same));
            }
            wave_sync();
          }
          u32 *t = src; src = dst; dst = t;
          // (the first pass reads the list itself and writes the scratch table; from then on the heap's table is the other side)
          if (shift == 0) dst = buf;
        }
        // four passes: list -> tmp -> buf -> tmp -> buf
        sorted = true;
      }
    }
    int m = 1;
    while (m < n) m <<= 1;
    if (!sorted) {
    for (int i = lane; i < m; i += 64) buf[i] = i < n ? ld_list<BIG>(pl.lpos[which] + i) : 0xFFFFFFFFu;
    wave_sync();
    // one compare-exchange pass of the bitonic network at distance j inside stage k, over
    // elements [0, cnt) of `v` whose global indices start at `base`
    auto pass = [&](u32 *v, int cnt, int base, int k, int j) {
      for (int i = lane; i < cnt; i += 64) {
        const int p = i ^ j;
        if (p > i) {
          const u32 x = v[i], y = v[p];
          const bool up = ((base + i) & k) == 0;
          if ((x > y) == up) { v[i] = y; v[p] = x; }
        }
      }
      wave_sync();
    };
    if (!BIG) {
      for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) pass(buf, m, 0, k, j);
    }
    else {
      // tier 2: the buffer is in global memory.  Every pass at distance j < C only pairs elements
      // of the same aligned block of C, so those passes run on a copy of the block in LDS (the
      // window slots and the window cache are idle here); only distances >= C touch global memory.
      u32 *blk = reinterpret_cast<u32 *>(lds.gwin);
      int C = 256;
      while (2 * C <= static_cast<int>(2 * (lds.max_jobs * lds.GW + (1u << kPosCacheBits))) && 2 * C <= m) C <<= 1;
      if (C > m) C = m;
      auto local = [&](int k_from, int k_to, int j_top) {  // stages k_from..k_to, distances j_top..1, block by block
        for (int b = 0; b < m; b += C) {
          for (int i = lane; i < C; i += 64) blk[i] = buf[b + i];
          wave_sync();
          for (int k = k_from; k <= k_to; k <<= 1)
            for (int j = min(k >> 1, j_top); j > 0; j >>= 1) pass(blk, C, b, k, j);
          for (int i = lane; i < C; i += 64) buf[b + i] = blk[i];
          wave_sync();
        }
      };
      local(2, C, C >> 1);  // stages that fit a block entirely
      for (int k = 2 * C; k <= m; k <<= 1) {
        for (int j = k >> 1; j >= C; j >>= 1) pass(buf, m, 0, k, j);
        local(k, k, C >> 1);
      }
    }
    }
    // unique + recompute diffs
    const WaveLds w = lds_of(end);
    const u64 *qpk = w.qpk + enc_of(lflags[which]) * w.W;
    const u32 nwords = (len_of(end) + 15) >> 4;
    int out = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const u32 v = i < n ? buf[i] : 0u;
      const bool keep = i < n && (i == 0 || buf[i - 1] != v);
      const u64 kept = __ballot(keep);
      if (keep) {
        const int dst = out + __popcll(kept & ((1ull << lane) - 1));
        pl.lpos[which][dst] = v;
        pl.ld[which][dst] = static_cast<i16>(v != 0 ? hamming(a.ix.genome, qpk, nwords, v)
                                                    : static_cast<int>(static_cast<i16>(0.4 * len_of(end))));
      }
      out += __popcll(kept);
    }
    lsz[which] = out;
    wave_sync();
  }

  // Tier 2's lists lie in global memory, and every binary search in them is a chain of dependent loads (thirteen for a
  // list of 4096 entries, per 64 entries that search): the marking pass of score_pairable and the windows of mate were a
  // sixth of a mid-sized pair's time.  So a list that is about to be searched leaves every 2^s-th position in LDS (at most
  // 512 of them, in the window cache's place -- idle outside the seed passes): a search runs through those first and ends
  // with at most s steps in memory -- none when the whole list fits.  Tier 1's lists are in LDS as they are.
  u32 *samp;      // [512] the sampled positions
  int samp_shift; // s
  int samp_n;     // samples held
  __device__ __forceinline__ void sample_list(int which) {
    if constexpr (BIG) {
      const int n = lsz[which];
      int s = 0;
      while ((n >> s) > 512) ++s;
      samp_shift = s;
      samp_n = (n + (1 << s) - 1) >> s;
      for (int j = lane_id(); j < samp_n; j += 64) samp[j] = pl.lpos[which][static_cast<u32>(j) << s];
      wave_sync();
    }
  }
  // first index of list `which` in [from, n) whose (position + add) is not below key, i.e. (pos + add) >= key -- or, with
  // STRICT, is above key: (pos + add) > key; the list is sorted and sample_list(which) has been called for it
  template <bool STRICT = false>
  __device__ __forceinline__ int search_list(int which, long long add, long long key, int from = 0) const {
    auto below = [&](u32 pos) { const long long v = static_cast<long long>(pos) + add; return STRICT ? v <= key : v < key; };
    int lo = from, n = lsz[which] - from;
    if constexpr (BIG) {
      // among the samples first: j = first sample at or after `from`'s block that is not below
      int jl = (from + (1 << samp_shift) - 1) >> samp_shift, jn = samp_n - jl;
      if (jn < 0) jn = 0;
      while (jn > 0) {
        const int half = jn >> 1;
        if (below(samp[jl + half])) { jl += half + 1; jn -= half + 1; } else jn = half;
      }
      // the answer lies in (sample jl - 1, sample jl]
      const int hi = min(lsz[which], jl << samp_shift);
      lo = max(from, jl > 0 ? ((jl - 1) << samp_shift) + 1 : 0);
      n = hi - lo;
      if (n < 0) n = 0;
    }
    while (n > 0) {
      const int half = n >> 1;
      if (below(ld_list<BIG>(pl.lpos[which] + lo + half))) { lo += half + 1; n -= half + 1; }
      else n = half;
    }
    return lo;
  }
  // first index whose position is >= key (lists are sorted here)
  __device__ __forceinline__ int lower_bound_pos(int which, long long key) const { return search_list<false>(which, 0, key); }

  // scores of every entry that has at least one concordant partner: the only
  // alignments best_pair can ever request (it computes them lazily, :1783-1790)
  __device__ __forceinline__ void score_pairable(int endA) {
    const int lane = lane_id();
    const int ends[2] = {endA, 1 - endA};
    const long long lenB = len_of(ends[1]);
    // pass 1: mark.  score 0 = cannot pair, 2L = exact hit (align() returns at once), -1 = needs the DP
    #pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int other = 1 - which, n = lsz[which];
      sample_list(other);
      for (int i = lane; i < n; i += 64) {
        const u32 pos = ld_list<BIG>(pl.lpos[which] + i);
        int mark = 0;
        if (pos != 0) {
          // A entry a pairs with B entry b iff a+min <= b+lenB <= a+max
          long long lo, hi;
          if (which == 0) { lo = static_cast<long long>(pos) + a.min_frag - lenB; hi = static_cast<long long>(pos) + a.max_frag - lenB; }
          else { lo = static_cast<long long>(pos) + lenB - a.max_frag; hi = static_cast<long long>(pos) + lenB - a.min_frag; }
          const int k = lower_bound_pos(other, lo);
          if (k < lsz[other] && static_cast<long long>(ld_list<BIG>(pl.lpos[other] + k)) <= hi)
            mark = ld_list<BIG>(pl.ld[which] + i) == 0 ? static_cast<int>(static_cast<i16>(2 * len_of(ends[which]))) : -1;
        }
        pl.lsc[which][i] = static_cast<i16>(mark);
      }
    }
    wave_sync();
    // pass 2: run the marked entries through the wavefront DP, up to kSeCap jobs at a time
    #pragma unroll
    for (int which = 0; which < 2; ++which) {
      const int end = ends[which], n = lsz[which];
      const int md = static_cast<i16>(a.valid_frac * len_of(end));
      int cursor = 0;
      for (;;) {
        int n_jobs = 0;
        while (cursor < n && n_jobs < static_cast<int>(kSeCap)) {
          const int i = cursor + lane;
          const bool need = i < n && ld_list<BIG>(pl.lsc[which] + i) == -1;
          const u64 m = __ballot(need);
          const int cnt = __popcll(m), take = min(cnt, static_cast<int>(kSeCap) - n_jobs);
          const int rank = __popcll(m & ((1ull << lane) - 1));
          if (need && rank < take) {
            lds.jpos[n_jobs + rank] = ld_list<BIG>(pl.lpos[which] + i);
            lds.jdf[n_jobs + rank] = (static_cast<u32>(ld_list<BIG>(pl.ld[which] + i)) << 16) | (lflags[which] & 0xFFFFu);
            pl.jidx[n_jobs + rank] = static_cast<u32>(i);
            pl.lsc[which][i] = 0;  // claimed
          }
          n_jobs += take;
          wave_sync();
          if (take < cnt) break;  // list full: the rest of this chunk is picked up next time
          cursor += 64;
        }
        if (n_jobs == 0) break;
        for (int s = 0; s < n_jobs;) {  // rounds of side-by-side bands
          const int first = s;
          s = score_jobs<LONG>(a.ix, lds, first, n_jobs, static_cast<int>(len_of(end)), md, static_cast<int>(end * 4 * lds.W));
          if (lane < s - first) pl.lsc[which][pl.jidx[first + lane]] = static_cast<i16>(lds.lbest[lane]);
          n_aln += static_cast<u32>(s - first);
          wave_sync();
        }
      }
    }
  }

  // traceback of one end's winning hit; updates (pos, diffs) and the end's CIGAR
  __device__ __forceinline__ void traceback_end(int end, u64 r, int scoring, int &d, u32 flags, u32 &pos,
                                                u32 &alen) {
    const int lane = lane_id();
    const int Ln = static_cast<int>(len_of(end));
    const int md = static_cast<i16>(a.valid_frac * len_of(end));
    u32 *cig_out = cig_of(end, r);
    u32 nops = 0;
    int n_ins = 0, n_del = 0;
    if (d == 0) {  // align<true> returns at once; build_cigar gives the default CIGAR (:404-409)
      if (lane == 0) cig_out[0] = static_cast<u32>(Ln) << 4;
      nops = 1;
      alen = static_cast<u32>(Ln);
    }
    else {
      const int bw = band_for(d, md);
      AlnJob job = {0, 0, 0, 0, 0};
      const u64 t_beg = static_cast<u64>(pos) - static_cast<u64>((bw - 1) / 2);
      if (lane < bw) {
        job.bw = bw; job.jl = lane;
        job.qoff = static_cast<int>((end * 4 + enc_of(flags)) * lds.W);
        job.t0nib = static_cast<int>(t_beg & 15u);
      }
      if (lane == 0) { lds.jpos[0] = pos; lds.jdf[0] = (static_cast<u32>(d) << 16) | (flags & 0xFFFFu); }
      wave_sync();
      stage_windows(a.ix, lds, 0, 1, md);
      wave_sync();
      int bv, brow;
      if constexpr (LONG || !kTracebackByRows) wavefront<true, LONG>(lds, job, Ln, bw, bw, bv, brow);
      else wavefront_rows<true>(lds, job, Ln, bw, bv, brow);
      const u64 k64 = (static_cast<u64>(static_cast<u32>(bv)) << 32) |
                      (static_cast<u64>(0xFFFFu - static_cast<u32>(brow)) << 8) |
                      static_cast<u64>(0xFFu - static_cast<u32>(lane));
      const u64 topk = wave_max_u64(lane < bw ? k64 : 0ull);
      const int br = static_cast<int>(0xFFFFu - static_cast<u32>((topk >> 8) & 0xFFFFu));
      const int bc = static_cast<int>(0xFFu - static_cast<u32>(topk & 0xFFu));
      const int sc = static_cast<i16>(static_cast<int>(topk >> 32));
      wave_sync();
      wave_cigar(lds.tb, lds.ctmp, Ln, d, md, sc, br, bc, cig_out, sink(), nops, n_ins, n_del, alen, pos,
                 overflow);
      wave_sync();
    }
    d = edit_distance(scoring, alen, n_ins, n_del);
    const u32 rl = alen - static_cast<u32>(n_ins) + static_cast<u32>(n_del);
    if (end) { n_ops[1] = nops; ref_len[1] = rl; }
    else { n_ops[0] = nops; ref_len[0] = rl; }
  }

  // best_pair, src/abismal.cpp:1722-1831.  List 0 is endA's (the reference's res1), list 1
  // endB's; `swapped` = endA is read 2.
  //
  // The reference walks list 1 (j2) and, for each entry, the window of list-0 entries (j1) whose
  // fragment would be min_dist..max_dist long; both lists are sorted, so the window of entry ib
  // is [lo, hi) with lo = first pos1 + max_dist >= lim and hi = first pos1 + min_dist > lim,
  // lim = pos2 + len2 -- what its rewind/advance loops arrive at.  Here 64 list-1 entries find
  // their windows at once, and all (j2, j1) steps of those windows are laid out in the
  // reference's order and taken 64 at a time; within a chunk only the steps that change
  // pe_element's state (:570-587) are visited one by one.  "*a1 == 0" (first visit of a list-0 entry, or
  // one whose score is 0) is "j1 >= the previous entry's hi", because hi never decreases; the
  // score of the most recent such alignment is what the reference's scr1 holds when a better
  // pair is recorded (:1787-1795), which the traceback's edit distance then uses.
  __device__ __forceinline__ void mate(int endA, u64 r, PairBest &best) {
    const int lane = lane_id();
    const int endB = 1 - endA;
    const bool swapped = endA == 1;
    const int na = lsz[0], nb = lsz[1];
    const u32 lenB = len_of(endB);
    const u32 *posA = pl.lpos[0], *posB = pl.lpos[1];
    int last_sa = 0, keep_sa = 0, keep_sb = 0;
    u32 keep_pa = 0, keep_pb = 0;
    int prev_hi = 0;
    int ib0 = (nb > 0 && static_cast<u32>(uni(static_cast<int>(ld_list<BIG>(posB)))) == 0u) ? 1 : 0;
    sample_list(0);
    for (; ib0 < nb && !best.sure_ambig(); ib0 += 64) {
      const int ib = ib0 + lane;
      u32 pb = 0;
      int db = 0, sb = 0, lo = 0, hi = 0;
      if (ib < nb) {
        pb = ld_list<BIG>(posB + ib);
        const u32 lim = pb + lenB;
        // (32-bit sums, as the reference's: positions are genome offsets below 2^32 - 2^16 and the fragment limits small)
        lo = search_list<false>(0, static_cast<long long>(a.max_frag), static_cast<long long>(lim));       // first pos1 + max_dist >= lim
        hi = search_list<true>(0, static_cast<long long>(a.min_frag), static_cast<long long>(lim), lo);    // first pos1 + min_dist > lim, from lo on
        if (hi > lo) { db = ld_list<BIG>(pl.ld[1] + ib); sb = ld_list<BIG>(pl.lsc[1] + ib); }
      }
      else { lo = hi = na; }
      const u32 cnt = static_cast<u32>(hi - lo);
      u32 total;
      const u32 start = wave_excl_sum(cnt, total);
      // entries before this one have been visited up to the previous entry's hi
      int seen = __shfl_up(hi, 1);
      if (lane == 0) seen = prev_hi;
      prev_hi = rdlane(hi, min(63, nb - 1 - ib0));
      int carry = 0;
      for (u32 c0 = 0; c0 < total && !best.sure_ambig(); c0 += 64) {
        lds.mark[lane] = 0;
        wave_sync();
        if (cnt && start - c0 < 64u) lds.mark[start - c0] = static_cast<u16>(lane + 1);
        wave_sync();
        const int m = wave_incl_max(static_cast<int>(lds.mark[lane]));
        const int owner = m ? m - 1 : carry;
        carry = rdlane(owner, 63);
        const u32 c = c0 + lane;
        const bool valid = c < total;
        const int ia = __shfl(lo, owner) + static_cast<int>(c - __shfl(start, owner));
        const u32 o_pb = __shfl(pb, owner);
        const int o_db = __shfl(db, owner), o_sb = __shfl(sb, owner), o_seen = __shfl(seen, owner);
        u32 pa = 0;
        int da = 0, sa = 0;
        if (valid) {
          pa = ld_list<BIG>(posA + ia);
          da = ld_list<BIG>(pl.ld[0] + ia);
          sa = ld_list<BIG>(pl.lsc[0] + ia);
        }
        // scr1 as of each step: the score of the latest step that (re)aligned its list-0 entry
        const u64 fresh = __ballot(valid && (ia >= o_seen || sa == 0));
        const u64 upto = fresh & ((2ull << lane) - 1);
        const int src = upto ? 63 - __builtin_clzll(upto) : lane;
        const int sa_src = __shfl(sa, src);
        const int scr1 = upto ? sa_src : last_sa;
        // the run of updates
        const int pair = static_cast<i16>(o_sb + sa);
        const long long key = valid ? (static_cast<long long>(pair) << 20) - (da + o_db) : LLONG_MIN;
        // Only a step that beats the standing best, or the first one that ties with it, changes
        // anything; a tie at the best possible score makes the reference stop on the spot
        // (sure_ambig, :1763,:1779) even though a later step might still have had fewer diffs.
        u64 rem = __ballot(valid);
        while (rem) {
          const long long cur = (static_cast<long long>(best.aln_score) << 20) - (best.d1 + best.d2);
          const u64 ge = __ballot(valid && key >= cur) & rem;
          if (!ge) break;
          const u64 gt = __ballot(valid && key > cur) & rem;
          const int wl = __builtin_ctzll(ge);
          if ((gt >> wl) & 1ull) {
            const int w_da = rdlane(da, wl), w_db = rdlane(o_db, wl);
            const u32 w_pa = rdlane(pa, wl), w_pb = rdlane(o_pb, wl);
            if (swapped) { best.d1 = w_db; best.f1 = lflags[1]; best.p1 = w_pb; best.d2 = w_da; best.f2 = lflags[0]; best.p2 = w_pa; }
            else { best.d1 = w_da; best.f1 = lflags[0]; best.p1 = w_pa; best.d2 = w_db; best.f2 = lflags[1]; best.p2 = w_pb; }
            best.aln_score = rdlane(pair, wl);
            keep_sa = rdlane(scr1, wl);
            keep_sb = rdlane(o_sb, wl);
            keep_pa = w_pa;
            keep_pb = w_pb;
            rem &= ~((2ull << wl) - 1);
          }
          else {
            best.f1 |= kFlagAmbig;
            if (best.aln_score == best.max_aln_score) break;
            rem = gt ? (rem & ~((1ull << __builtin_ctzll(gt)) - 1)) : 0ull;  // more ties change nothing
          }
        }
        last_sa = rdlane(scr1, static_cast<int>(min(63u, total - 1 - c0)));
        wave_sync();
      }
    }
    if (keep_pa == 0)
      return;
    int da = swapped ? best.d2 : best.d1, db = swapped ? best.d1 : best.d2;
    u32 fa = swapped ? best.f2 : best.f1, fb = swapped ? best.f1 : best.f2;
    u32 pa = keep_pa, pb = keep_pb, la = 0, lb = 0;
    traceback_end(endA, r, keep_sa, da, fa, pa, la);
    traceback_end(endB, r, keep_sb, db, fb, pb, lb);
    const u32 fe = pb + lb;
    if (fe >= pa + a.min_frag && fe <= pa + a.max_frag) {
      if (swapped) { best.d1 = db; best.f1 = fb; best.p1 = pb; best.d2 = da; best.f2 = fa; best.p2 = pa; }
      else { best.d1 = da; best.f1 = fa; best.p1 = pa; best.d2 = db; best.f2 = fb; best.p2 = pb; }
    }
    else
      best.clear();
  }

  // best_single (:1715-1720): every entry of a list, in order, into a single-end set
  __device__ __forceinline__ void feed_single(SeSet &S, const u32 *lp, const i16 *ldv, int n, u32 flags) {
    for (int k0 = 0; k0 < n && !S.sure_ambig; k0 += 64) {
      const int i = k0 + lane_id();
      const int dl = i < n ? static_cast<int>(ldv[i]) : 0;
      const u32 pv = i < n ? lp[i] : 0u;
      const int m = min(64, n - k0);
      for (int k = 0; k < m && !S.sure_ambig; ++k) {
        S.admit(false, rdlane(dl, k), flags, rdlane(pv, k));
        ++wt.updates;
      }
    }
  }
  // tier 2's log of the lists best_single would have consumed: 8 segments (orientation call x list)
  __device__ __forceinline__ u32 *log_head(int seg) const { return log_base + seg * 4; }
  __device__ __forceinline__ u32 *log_pos(int seg) const { return log_base + 32 + static_cast<u64>(seg) * pl.cap; }
  __device__ __forceinline__ i16 *log_d(int seg) const {
    return reinterpret_cast<i16 *>(log_base + 32 + 8ull * pl.cap) + static_cast<u64>(seg) * pl.cap;
  }
  __device__ __forceinline__ void replay_singles() {
    for (int seg = 0; seg < 8; ++seg) {
      const u32 *h = log_head(seg);
      const int n = uni(static_cast<int>(h[0]));
      if (n == 0) continue;
      const u32 flags = static_cast<u32>(uni(static_cast<int>(h[1])));
      if (uni(static_cast<int>(h[2])) == 0) feed_single(se[0], log_pos(seg), log_d(seg), n, flags);
      else feed_single(se[1], log_pos(seg), log_d(seg), n, flags);
    }
  }

  // map_fragments + select_maps + best_single (:1715-1720, :1833-1885)
  template <bool TIMED> __device__ __forceinline__ bool orientation(int O, int endA, bool ar, u64 r, PairBest &best) {
    const int endB = 1 - endA;
    const bool emptyA = len_of(endA) < a.ix.min_len, emptyB = len_of(endB) < a.ix.min_len;
    if (emptyA && emptyB) {
      // res1/res2 are reset and nothing else happens (:1863-1866)
      return false;
    }
    // endA forward with (rc=0, a_rich=ar); endB reverse-complemented with (rc=1, a_rich=!ar)
    if constexpr (PHASE == kSeed) {  // (one list at a time: each is handed over as soon as it is finished)
      seed_end<TIMED>(0, endA, false, ar);
      emit_list(0, r, 2 * O);
      if (need_big) return true;
      seed_end<TIMED>(0, endB, true, !ar);
      emit_list(0, r, 2 * O + 1);
      return true;
    }
    else if constexpr (PHASE == kMate) {
      load_list(0, r, 2 * O, ar ? kFlagARich : 0u);
      load_list(1, r, 2 * O + 1, kFlagRC | (ar ? 0u : kFlagARich));
    }
    else {
      seed_end<TIMED>(0, endA, false, ar);
      seed_end<TIMED>(1, endB, true, !ar);
      if (need_big && !BIG)
        return true;
    }
    long long t0 = 0, t1 = 0;
    if (worth[0] && worth[1]) {
      ABM_STAMP(t0);
      sort_unique(0, endA);
      sort_unique(1, endB);
      ABM_STAMP(t1);
      if (TIMED) t_sort += t1 - t0;
      score_pairable(endA);
      ABM_STAMP(t0);
      if (TIMED) t_score += t0 - t1;
      mate(endA, r, best);
      ABM_STAMP(t1);
      if (TIMED) t_mate += t1 - t0;
    }
    else {
      heap_order_now(0);
      heap_order_now(1);
    }
    ABM_STAMP(t0);
    const int ends[2] = {endA, endB};
    if (BIG) {
      // best_single is only ever needed if the pair ends up without a reportable concordant hit,
      // which is rare; tier 2 keeps each call's two lists in a per-wave log and replays them into
      // the single-end sets only then (replay_singles), in the same order
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const int seg = 2 * O + which, n = lsz[which];
        u32 *lp = log_pos(seg);
        i16 *ldv = log_d(seg);
        for (int i = lane_id(); i < n; i += 64) { lp[i] = pl.lpos[which][i]; ldv[i] = pl.ld[which][i]; }
        if (lane_id() == 0) { u32 *h = log_head(seg); h[0] = static_cast<u32>(n); h[1] = lflags[which]; h[2] = static_cast<u32>(ends[which]); }
      }
      wave_sync();
    }
    else {
      // best_single: every entry of each set, in array order, into that end's single-end set.  List 0 is endA's:
      // with endA == 1 the two (register-resident) sets trade places around the two calls, so that each call names
      // its set at compile time
      if (endA) { const SeSet t = se[0]; se[0] = se[1]; se[1] = t; }
      feed_single(se[0], pl.lpos[0], pl.ld[0], lsz[0], lflags[0]);
      feed_single(se[1], pl.lpos[1], pl.ld[1], lsz[1], lflags[1]);
      if (endA) { const SeSet t = se[0]; se[0] = se[1]; se[1] = t; }
    }
    ABM_STAMP(t1);
    if (TIMED) t_single += t1 - t0;
    return true;
  }
};

// REC: the seed passes filter on the window records (DevIndex::wrec) -- a launch none of whose ends is longer than they serve
template <bool BIG, bool TIMED, bool COOP, int WPS, bool LONG = false, int PHASE = kWhole, bool REC = false>
__global__ __launch_bounds__(64, WPS) void map_pe_kernel(PeArgs a) {
  static_assert(!REC || (COOP && PHASE != kMate), "window records feed the cooperative filter of the seed passes");
  static_assert(!LONG || (BIG && !COOP && !TIMED && PHASE == kWhole), "the long-end launch: tier 2's lists, nibble filter, no stamps");
  static_assert(PHASE != kSeed || !BIG, "the seed kernel keeps its lists in LDS (and its staging area)");
  static_assert(PHASE != kMate || !COOP, "the mate kernels fetch no candidate windows");
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  PeWave<BIG, COOP, LONG, PHASE, REC> w{a};
  WaveLds &lds = w.lds;
  lds.W = a.W; lds.WB = a.WB; lds.GW = a.GW;
  u32 *after_heap;
  if constexpr (LONG) {
    // read data in this wave's piece of global memory; LDS holds the two window slots, the cache and the job lists
    u64 *q = a.long_q + static_cast<u64>(blockIdx.x) * (8ull * a.W + 8ull * a.WB);
    lds.qpk = q;
    lds.qbits = q + 8 * a.W;
    lds.MB = 0;
    lds.qmask = nullptr;
    lds.gwin = reinterpret_cast<u64 *>(smem);
    lds.max_jobs = 2;
    lds.pcache = lds.gwin + 2 * a.GW;
    lds.tb = a.long_tb + static_cast<u64>(blockIdx.x) * a.long_tb_bytes;
    lds.ctmp = a.long_ctmp + static_cast<u64>(blockIdx.x) * ((a.ctmp_cap + 1) & ~1u);
    lds.jpos = reinterpret_cast<u32 *>(lds.pcache + (1u << kPosCacheBits));
  }
  else if constexpr (PHASE == kSeed) {
    // (pe_seed_lds_bytes) read data, the window cache, the step's distances, then the set: no alignment state at all
    lds.qpk = reinterpret_cast<u64 *>(smem);
    lds.qbits = lds.qpk + 8 * a.W;
    lds.MB = (a.max_len + kPlaneBlock - 1) / kPlaneBlock;
    lds.qmask = lds.qbits + 8 * a.WB;  // [2 ends][4][MB][4]
    lds.pcache = lds.qmask + 8 * lds.MB * 4;
    lds.gwin = nullptr; lds.max_jobs = 0; lds.tb = nullptr; lds.ctmp = nullptr; lds.jpos = nullptr; lds.jdf = nullptr;
    w.pl.jidx = nullptr;
    lds.lbest = reinterpret_cast<int *>(lds.pcache + (1u << kPosCacheBits));
  }
  else {
    lds.qpk = reinterpret_cast<u64 *>(smem);
    if constexpr (PHASE == kMate) {  // (pe_mate_lds_bytes: the packed encodings are all a mate kernel reads of a read)
      lds.qbits = nullptr; lds.MB = 0; lds.qmask = nullptr;
      lds.gwin = lds.qpk + 8 * a.W;
    }
    else {
      lds.qbits = lds.qpk + 8 * a.W;
      lds.MB = (a.max_len + kPlaneBlock - 1) / kPlaneBlock;
      lds.qmask = lds.qbits + 8 * a.WB;  // [2 ends][4][MB][4]
      lds.gwin = lds.qmask + 8 * lds.MB * 4;
    }
    lds.max_jobs = kMaxJobs;
    lds.pcache = lds.gwin + kMaxJobs * a.GW;
    // the traceback table overlays window slots 1.. and the window cache (a traceback uses slot 0 only)
    lds.tb = reinterpret_cast<u8 *>(lds.gwin + a.GW);
    lds.ctmp = reinterpret_cast<u32 *>(reinterpret_cast<u8 *>(lds.pcache + (1u << kPosCacheBits)) + a.tb_extra);
    lds.jpos = lds.ctmp + a.ctmp_cap;
  }
  if constexpr (PHASE != kSeed) {
    lds.jdf = lds.jpos + kSeCap;
    w.pl.jidx = lds.jdf + kSeCap;
    lds.lbest = reinterpret_cast<int *>(w.pl.jidx + kSeCap);
  }
  // tier 1: the set's heap is in LDS.  tier 2: a 32768-entry heap per wave would cap the CU at one
  // wave, so it lives in global memory (L2-resident, touched by this wave only) and occupancy stays normal
  after_heap = reinterpret_cast<u32 *>(lds.lbest + 64);
  if (BIG) w.pl.heap = a.heap_ws + static_cast<u64>(blockIdx.x) * a.cap;
  else { w.pl.heap = after_heap; after_heap += a.cap; }
  w.pl.cap = a.cap;
  if (BIG) {
    u32 *ws = a.list_ws + static_cast<u64>(blockIdx.x) * 4 * a.cap;  // 2 pos arrays + (2 diffs + 2 scores) as i16
    w.pl.lpos[0] = ws; w.pl.lpos[1] = ws + a.cap;
    i16 *h = reinterpret_cast<i16 *>(ws + 2 * a.cap);
    w.pl.ld[0] = h; w.pl.ld[1] = h + a.cap; w.pl.lsc[0] = h + 2 * a.cap; w.pl.lsc[1] = h + 3 * a.cap;
  }
  else if constexpr (PHASE == kSeed) {  // one list at a time (positions, diffs)
    w.pl.lpos[0] = w.pl.lpos[1] = after_heap;
    i16 *h = reinterpret_cast<i16 *>(after_heap + a.cap);
    w.pl.ld[0] = w.pl.ld[1] = h; w.pl.lsc[0] = w.pl.lsc[1] = nullptr;
    after_heap = reinterpret_cast<u32 *>(h + a.cap + (a.cap & 1u));
  }
  else {
    w.pl.lpos[0] = after_heap; w.pl.lpos[1] = after_heap + a.cap;
    i16 *h = reinterpret_cast<i16 *>(after_heap + 2 * a.cap);
    w.pl.ld[0] = h; w.pl.ld[1] = h + a.cap; w.pl.lsc[0] = h + 2 * a.cap; w.pl.lsc[1] = h + 3 * a.cap;
    after_heap = reinterpret_cast<u32 *>(h + 4 * a.cap);
  }
  lds.smark = after_heap;
  lds.sdelta = after_heap + 128;
  lds.mark = reinterpret_cast<u16 *>(after_heap + 256);
  lds.hres = reinterpret_cast<u16 *>(lds.lbest);
  lds.G = a.G;
  if constexpr (PHASE != kMate) { lds.smark[lane] = 0; lds.smark[64 + lane] = 0; }
  else lds.mark = reinterpret_cast<u16 *>(after_heap);  // (no seed passes: no segment marks)
  w.seg_epoch = 0;

  w.P.heap = w.pl.heap;
  w.samp = reinterpret_cast<u32 *>(lds.pcache); w.samp_shift = 0; w.samp_n = 0;
  w.P.spill_pos = nullptr; w.P.spill_d = nullptr; w.P.spill_cap = 0; w.P.spilled = false;
  w.stage_pos = nullptr; w.stage_d = nullptr;
  if constexpr (PHASE == kSeed) {
    if (a.scap > a.cap) {
      w.stage_pos = a.stage_pos + static_cast<u64>(blockIdx.x) * a.scap;
      w.stage_d = a.stage_d + static_cast<u64>(blockIdx.x) * a.scap;
    }
  }
  w.log_base = BIG ? a.log_ws + static_cast<u64>(blockIdx.x) * (32ull + 12ull * a.cap) : nullptr;
  // scratch table for permuting a list: global for tier 2; tier 1 borrows the window cache (idle outside seed passes)
  static_assert(kPeTier1Cap * 4 <= (8u << kPosCacheBits), "tier-1 scratch table must fit the window cache");
  w.pl.tmp = BIG ? a.payload_ws + static_cast<u64>(blockIdx.x) * a.cap : reinterpret_cast<u32 *>(lds.pcache);
  w.P.cap_avail = a.cap;
  w.wt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  w.n_aln = 0;
  w.overflow = false;
  w.t_sort = w.t_score = w.t_mate = w.t_single = 0;
  bool too_long = false;
  long long t_begin = 0, t_fb = 0;
  u32 routed_small = 0, routed_whole = 0, routed_big = 0;  // seed kernel: pairs of this wave by route
  ABM_STAMP(t_begin);

  auto next_item = [&]() -> u64 {
    unsigned long long v = 0;
    if (lane == 0) v = atomicAdd(a.next_read, 1ull);
    return (static_cast<u64>(static_cast<u32>(uni(static_cast<int>(v >> 32)))) << 32) |
           static_cast<u32>(uni(static_cast<int>(v)));
  };
  const u64 n_items = BIG ? static_cast<u64>(*a.subset_count) : a.n_pairs;
  u64 it_next = next_item();
  while (it_next < n_items) {
    const u64 it = it_next;
    it_next = next_item();
    __builtin_amdgcn_s_setprio(0);  // (a heavy end raises its wave's priority: seed_pass)
    const u64 r = BIG ? static_cast<u64>(a.subset[it]) : (a.order ? static_cast<u64>(a.order[it]) : it);
    if constexpr (PHASE == kMate && !BIG) {  // (the small-list mate kernel walks the whole batch: the other routes' pairs are not its)
      if (static_cast<u8>(uni(static_cast<int>(a.need_big[r]))) != kRouteSmall) continue;
    }
    w.L[0] = a.lens1[r];
    w.L[1] = a.lens2[r];
    if (w.L[0] > kMaxReadLen || w.L[1] > kMaxReadLen) { too_long = true; w.L[0] = w.L[1] = 0; }
    // (a pair with an end beyond this launch's length is the long-end launch's: it overwrites what is stored for it here)
    if (!LONG && (w.L[0] > kLdsReadLen || w.L[1] > kLdsReadLen)) w.L[0] = w.L[1] = 0;
    // stage both ends' four encodings and their 2-letter bit strings (LONG: packed by list position)
    #pragma unroll
    for (int e = 0; e < 2; ++e) {
      const u64 *src = (e ? a.packed2 : a.packed1) + (LONG ? it : r) * 4 * a.W;
      for (u32 k = lane; k < 4 * a.W; k += 64) lds.qpk[e * 4 * a.W + k] = src[k];
    }
    wave_sync();
    if constexpr (PHASE != kMate) {
      for (u32 e8 = 0; e8 < 8; ++e8)
        for (u32 wb = 0; wb < a.WB; ++wb) {
          const u32 j = wb * 64 + lane, Le = e8 < 4 ? w.L[0] : w.L[1];
          const bool b = j < Le ? bit2(q_nibble(lds.qpk + e8 * a.W, j)) : true;
          const u64 word = __ballot(b);
          if (lane == 0) lds.qbits[e8 * a.WB + wb] = word;
        }
      if constexpr (COOP) { build_qmasks(w.lds_of(0), w.L[0]); build_qmasks(w.lds_of(1), w.L[1]); }
      wave_sync();
      #pragma unroll
      for (int e = 0; e < 2; ++e)  // 44-46 bases: seeds reach past the end of the read (see ghost_bits)
        if (w.L[e] >= a.ix.min_len && w.L[e] < max(a.ix.window, w.L[e] >> 1) + kKeyWeight - 1)
          ghost_bits(e ? a.packed2 : a.packed1, e ? a.lens2 : a.lens1, r, w.L[e], a.max_len, a.ix.min_len, a.W, a.WB, lds.qbits + e * 4 * a.WB);
    }

    if constexpr (PHASE == kSeed) {
      // both seed passes of every orientation call's two ends; the lists go to the hand-over area, the pair to its route
      w.need_big = false;
      w.max_set = 0;
      const int n_or = a.mode == 2 ? 4 : 2;
#pragma clang loop unroll(disable)
      for (int o = 0; o < n_or; ++o) {
        if (w.need_big) break;
        const bool ar_o = a.mode == 2 ? (o == 1 || o == 2) : ((a.mode == 1) != (o == 1));
        PairBest unused;
        (void)w.template orientation<TIMED>(o, o & 1, ar_o, r, unused);
      }
      const u8 route = w.need_big ? kRouteWhole : (w.max_set > static_cast<int>(kPeTier1Cap) ? kRouteBig : kRouteSmall);
      if (lane == 0) a.need_big[r] = route;
      routed_small += route == kRouteSmall; routed_whole += route == kRouteWhole; routed_big += route == kRouteBig;
      continue;
    }

    PairBest best;
    best.f1 = best.f2 = 0;
    best.clear(w.L[0] >= a.ix.min_len ? w.L[0] : 0u, w.L[1] >= a.ix.min_len ? w.L[1] : 0u);
    w.se[0].begin_read(w.L[0] >= a.ix.min_len ? w.L[0] : 0u);
    w.se[1].begin_read(w.L[1] >= a.ix.min_len ? w.L[1] : 0u);
    w.need_big = false;
    if (BIG && lane < 8) w.log_head(lane)[0] = 0;  // no list logged yet for this pair
    w.max_set = 0;
    long long t_pair = 0;
    ABM_STAMP(t_pair);
    const long long ph0[8] = {w.wt.t_probe, w.wt.t_stream, w.wt.t_replay, w.t_sort, w.t_score, w.t_mate, w.t_single, t_fb};
    w.n_ops[0] = w.n_ops[1] = 0;
    w.ref_len[0] = w.ref_len[1] = 0;

    bool any = false;
    // orientation(endA, alphabet): src/abismal.cpp:1963-1979, :2106-2133
    // ONE copy of the orientation code, looped over (six inlined copies made the tier-1 kernel 155 k lines of assembly)
    {
      const int n_or = a.mode == 2 ? 4 : 2;
#pragma clang loop unroll(disable)
      for (int o = 0; o < n_or; ++o) {
        if (o > 0 && w.need_big && !BIG) break;
        const bool ar_o = a.mode == 2 ? (o == 1 || o == 2) : ((a.mode == 1) != (o == 1));
        any |= w.template orientation<TIMED>(o, o & 1, ar_o, r, best);
      }
    }
    if (w.need_big && !BIG) {
      if (lane == 0) a.need_big[r] = kRouteWhole;
      continue;
    }
    if (!any) {  // :1981-1985
      best.clear();
      w.se[0].begin_read(0); w.se[0].hk = 32767 * 256; w.se[0].cutoff = 32767; w.se[0].best_d = 32767;
      w.se[1].begin_read(0); w.se[1].hk = 32767 * 256; w.se[1].cutoff = 32767; w.se[1].best_d = 32767;
    }
    {  // valid_pair, :624-631, on whatever CIGARs the mating left behind
      const u32 a1 = w.ref_len[0], a2 = w.ref_len[1];
      const bool ok = long_enough(a1, w.L[0], a.ix.min_len) && long_enough(a2, w.L[1], a.ix.min_len) &&
                      static_cast<i16>(best.d1 + best.d2) <= static_cast<i16>(a.valid_frac * (a1 + a2));
      if (!ok) best.clear();
    }
    Hit h1, h2;
    h1.diffs = static_cast<i16>(0.4 * w.L[0]); h1.flags = 0; h1.pos = 0;
    h2.diffs = static_cast<i16>(0.4 * w.L[1]); h2.flags = 0; h2.pos = 0;
    if (!best.should_report(a.allow_ambig != 0)) {  // single-end fallback at half the error budget
      if (BIG) { wave_sync(); w.replay_singles(); }
      long long tf0 = 0, tf1 = 0;
      ABM_STAMP(tf0);
      #pragma unroll
      for (int e = 0; e < 2; ++e) {
        const WaveLds we = w.lds_of(e);
        u32 nops = 0;
        Hit &h = e ? h2 : h1;
        u32 n_single = 0;
        choose_se<LONG>(a.ix, we, w.L[e], a.valid_frac / 2, w.se[e], h, w.cig_of(e, r), w.sink(), nops,
                        w.overflow, w.n_aln, n_single);
        if (nops != 0) w.n_ops[e] = nops;  // whatever traceback ran last owns the slot (A.10)
      }
      ABM_STAMP(tf1);
      if (TIMED) t_fb += tf1 - tf0;
    }
    if (lane == 0) {
      u32 *po = reinterpret_cast<u32 *>(a.pairs) + r * 5;
      po[0] = static_cast<u32>(static_cast<u16>(static_cast<i16>(best.aln_score)));
      po[1] = static_cast<u32>(static_cast<u16>(static_cast<i16>(best.d1))) | ((best.f1 & 0xFFFFu) << 16);
      po[2] = best.p1;
      po[3] = static_cast<u32>(static_cast<u16>(static_cast<i16>(best.d2))) | ((best.f2 & 0xFFFFu) << 16);
      po[4] = best.p2;
      a.se1[r] = h1;
      a.se2[r] = h2;
      a.cig_n1[r] = w.n_ops[0];
      a.cig_n2[r] = w.n_ops[1];
      if (!BIG && PHASE == kWhole) a.need_big[r] = kRouteSmall;
      if (TIMED && a.pair_diag)
        a.pair_diag[r] = (min(static_cast<u32>(w.max_set), 0xFFFFu) << 16) |
                         static_cast<u32>(min((phase_stamp() - t_pair) >> 20, 0xFFFFll));
      if (TIMED && a.pair_phases) {
        const long long ph1[8] = {w.wt.t_probe, w.wt.t_stream, w.wt.t_replay, w.t_sort, w.t_score, w.t_mate, w.t_single, t_fb};
        for (int k = 0; k < 8; ++k) a.pair_phases[r * 8 + k] += static_cast<u32>((ph1[k] - ph0[k]) >> 10);
      }
    }
  }
  if (a.work) {
    auto wsum = [&](u32 v) { u32 t; (void)wave_excl_sum(v, t); return t; };
    const u32 s0 = wsum(w.wt.seed_iters), s1 = wsum(w.wt.probes), s2 = wsum(w.wt.cands), s3 = wsum(w.wt.words);
    const u32 wsum_hits = wsum(w.wt.cache_hits);
    if (lane == 0) {
      atomicAdd(&a.work[0], static_cast<unsigned long long>(s0));
      atomicAdd(&a.work[1], static_cast<unsigned long long>(s1));
      atomicAdd(&a.work[2], static_cast<unsigned long long>(s2));
      atomicAdd(&a.work[3], static_cast<unsigned long long>(s3));
      atomicAdd(&a.work[4], static_cast<unsigned long long>(w.wt.updates));
      atomicAdd(&a.work[5], static_cast<unsigned long long>(w.n_aln));
      atomicAdd(&a.work[11], static_cast<unsigned long long>(wsum_hits));
      if (TIMED) {
        atomicAdd(&a.work[6], static_cast<unsigned long long>(w.wt.t_probe));
        atomicAdd(&a.work[7], static_cast<unsigned long long>(w.wt.t_stream));
        atomicAdd(&a.work[8], static_cast<unsigned long long>(w.wt.t_replay));
        atomicAdd(&a.work[9], static_cast<unsigned long long>(t_fb));
        atomicAdd(&a.work[10], static_cast<unsigned long long>(phase_stamp() - t_begin));
        atomicAdd(&a.work[12], static_cast<unsigned long long>(w.t_sort));
        atomicAdd(&a.work[13], static_cast<unsigned long long>(w.t_score));
        atomicAdd(&a.work[14], static_cast<unsigned long long>(w.t_mate));
        atomicAdd(&a.work[15], static_cast<unsigned long long>(w.t_single));
      }
    }
  }
  if constexpr (PHASE == kSeed) {
    if (a.split_stats && lane == 0) {
      if (routed_small) atomicAdd(&a.split_stats[kRouteSmall], static_cast<unsigned long long>(routed_small));
      if (routed_whole) atomicAdd(&a.split_stats[kRouteWhole], static_cast<unsigned long long>(routed_whole));
      if (routed_big) atomicAdd(&a.split_stats[kRouteBig], static_cast<unsigned long long>(routed_big));
    }
  }
  if (lane == 0 && (w.overflow || too_long))
    atomicOr(a.status, (w.overflow ? 1u : 0u) | (too_long ? 2u : 0u));
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

// compact the pairs of one route (PeArgs::need_big) into a list for that route's launch, heaviest weight class first (a
// counting sort on the class the ordering kernels already computed): the few pairs with huge
// candidate sets run for a long time on their single wave and must not start last
__global__ __launch_bounds__(256) void big_hist_kernel(const u8 *__restrict__ need_big, const u8 *__restrict__ cls,
                                                       u64 n, u8 want, u32 *__restrict__ class_count) {
  __shared__ u32 hist[33];
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n && need_big[r] == want) atomicAdd(&hist[cls[r]], 1u);
  __syncthreads();
  if (threadIdx.x < 33 && hist[threadIdx.x]) atomicAdd(&class_count[threadIdx.x], hist[threadIdx.x]);
}
__global__ void big_bases_kernel(u32 *class_count /*[33] in: counts, out: start of each class*/, u32 *total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    u32 at = 0;
    for (int c = 32; c >= 0; --c) { const u32 k = class_count[c]; class_count[c] = at; at += k; }
    *total = at;
  }
}
__global__ __launch_bounds__(256) void big_scatter_kernel(const u8 *__restrict__ need_big, const u8 *__restrict__ cls,
                                                          u64 n, u8 want, u32 *__restrict__ class_cursor,
                                                          u32 *__restrict__ subset) {
  __shared__ u32 hist[33], base[33];
  if (threadIdx.x < 33) hist[threadIdx.x] = 0;
  __syncthreads();
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  const bool take = r < n && need_big[r] == want;
  u32 c = 0, rank = 0;
  if (take) { c = cls[r]; rank = atomicAdd(&hist[c], 1u); }
  __syncthreads();
  if (threadIdx.x < 33 && hist[threadIdx.x]) base[threadIdx.x] = atomicAdd(&class_cursor[threadIdx.x], hist[threadIdx.x]);
  __syncthreads();
  if (take) subset[base[c] + rank] = static_cast<u32>(r);
}

size_t pe_lds_bytes(u32 W, u32 WB, u32 GW, u32 cig_stride, u32 max_len, double valid_frac, u32 cap, bool big) {
  const u32 MB = (max_len + kPlaneBlock - 1) / kPlaneBlock;
  size_t b = static_cast<size_t>(8) * W * 8 + static_cast<size_t>(8) * WB * 8 + static_cast<size_t>(8) * MB * 4 * 8 +
             (static_cast<size_t>(8) << kPosCacheBits) + static_cast<size_t>(kMaxJobs) * GW * 8 +
             static_cast<size_t>(cig_stride) * 4 + 3 * kSeCap * 4 + 64 * 4 + 2 * 128 * 4 + 64 * 2;
  if (!big) b += static_cast<size_t>(cap) * (4 + 2 * 4 + 4 * 2);
  b += tb_extra_bytes(GW, max_len, valid_frac);
  return (b + 15) & ~static_cast<size_t>(15);
}

int pe_waves_per_simd(size_t lds, bool timed, bool coop) {
  // (LDS of a CU: 160 KB on gfx950)
  return (!timed && coop && lds != 0 && (160u * 1024u) / lds <= 13u) ? 3 : ABM_PE_WAVES_PER_SIMD;
}

int pe_resident_waves(size_t lds, bool big, int wps) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  hipError_t e;
  if (wps == 3) e = big ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<true, false, true, 3>, 64, lds)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<false, false, true, 3>, 64, lds);
  else e = big ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<true, false, true, ABM_PE_WAVES_PER_SIMD>, 64, lds)
               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<false, false, true, ABM_PE_WAVES_PER_SIMD>, 64, lds);
  if (e != hipSuccess) return 0;
  return per_cu * prop.multiProcessorCount;
}

static bool pe_records(const PeArgs &a) { return a.G == 4 && a.ix.wrec != nullptr && a.max_len <= a.ix.wrec_max_len; }
template <bool BIG, bool TIMED>
static void launch_pe_variant(const PeArgs &a, size_t lds, u32 grid, int wps, hipStream_t st) {
  // a.G != 0: the filter reads the genome's bit planes (cooperative window loads) -- or, for ends the window records serve, those
  if (BIG && pe_records(a)) {
    if constexpr (BIG) {
      if constexpr (!TIMED) {
        if (wps == 3) { hipLaunchKernelGGL((map_pe_kernel<true, false, true, 3, false, kWhole, true>), dim3(grid), dim3(64), lds, st, a); return; }
      }
      hipLaunchKernelGGL((map_pe_kernel<true, TIMED, true, ABM_PE_WAVES_PER_SIMD, false, kWhole, true>), dim3(grid), dim3(64), lds, st, a);
    }
  }
  else if (a.G != 0) {
    if constexpr (!TIMED) {
      if (wps == 3) { hipLaunchKernelGGL((map_pe_kernel<BIG, false, true, 3>), dim3(grid), dim3(64), lds, st, a); return; }
    }
    hipLaunchKernelGGL((map_pe_kernel<BIG, TIMED, true, ABM_PE_WAVES_PER_SIMD>), dim3(grid), dim3(64), lds, st, a);
  }
  else hipLaunchKernelGGL((map_pe_kernel<BIG, TIMED, false, ABM_PE_WAVES_PER_SIMD>), dim3(grid), dim3(64), lds, st, a);
}

hipError_t launch_map_pe(const PeArgs &a, size_t lds, u32 grid, bool big, bool timed, int wps, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (big && timed) launch_pe_variant<true, true>(a, lds, grid, wps, st);
  else if (big) launch_pe_variant<true, false>(a, lds, grid, wps, st);
  else if (timed) launch_pe_variant<false, true>(a, lds, grid, wps, st);
  else launch_pe_variant<false, false>(a, lds, grid, wps, st);
  return hipGetLastError();
}

// ---- the long-end launch --------------------------------------------------------------------------------------
// pairs of this batch with an end of kLdsReadLen + 1 .. kMaxReadLen bases
__global__ __launch_bounds__(256) void collect_long_pairs_kernel(const u32 *__restrict__ lens1, const u32 *__restrict__ lens2, u64 n,
                                                                 u32 *__restrict__ list, u32 *__restrict__ count) {
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const u32 a = lens1[r], b = lens2[r];
  if ((a > kLdsReadLen || b > kLdsReadLen) && a <= kMaxReadLen && b <= kMaxReadLen) list[atomicAdd(count, 1u)] = static_cast<u32>(r);
}
hipError_t launch_collect_long_pairs(const u32 *d_lens1, const u32 *d_lens2, u64 n, u32 *d_list, u32 *d_count, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(collect_long_pairs_kernel, dim3(static_cast<u32>((n + 255) / 256)), dim3(256), 0, st, d_lens1, d_lens2, n, d_list, d_count);
  return hipGetLastError();
}
size_t pe_long_lds_bytes(u32 GW) {
  const size_t b = static_cast<size_t>(2) * GW * 8 + (static_cast<size_t>(8) << kPosCacheBits) + 3 * kSeCap * 4 + 64 * 4 + 2 * 128 * 4 + 64 * 2;
  return (b + 15) & ~static_cast<size_t>(15);
}
size_t pe_long_q_words(u32 W, u32 WB) { return 8ull * W + 8ull * WB; }
int pe_long_resident_waves(u32 GW) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<true, false, false, 1, true>, 64, pe_long_lds_bytes(GW)) != hipSuccess) return 0;
  return std::min(per_cu, 1) * prop.multiProcessorCount;  // (one wave per CU: each has megabytes of workspace in global memory)
}
hipError_t launch_map_pe_long(const PeArgs &a, u32 grid, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL((map_pe_kernel<true, false, false, 1, true>), dim3(grid), dim3(64), pe_long_lds_bytes(a.GW), st, a);
  return hipGetLastError();
}

hipError_t launch_collect_big(const u8 *need_big, const u8 *cls, u64 n, u8 want, u32 *class33, u32 *subset, u32 *count,
                              hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(class33, 0, 33 * sizeof(u32), st);
  if (e != hipSuccess) return e;
  const u32 blocks = static_cast<u32>((n + 255) / 256);
  hipLaunchKernelGGL(big_hist_kernel, dim3(blocks), dim3(256), 0, st, need_big, cls, n, want, class33);
  hipLaunchKernelGGL(big_bases_kernel, dim3(1), dim3(64), 0, st, class33, count);
  hipLaunchKernelGGL(big_scatter_kernel, dim3(blocks), dim3(256), 0, st, need_big, cls, n, want, class33, subset);
  return hipGetLastError();
}

// ---- the phase-split launches ------------------------------------------------------------------------------------
// Registers: the seed kernel is built for kPeSeedWps waves per SIMD, the mate kernels for kPeMateWps (what each needs
// without scratch: profiles/r05_pe_split_resources.log)
#ifndef ABM_PE_SEED_WPS
#define ABM_PE_SEED_WPS 4
#endif
#ifndef ABM_PE_MATE_WPS
#define ABM_PE_MATE_WPS 4
#endif
constexpr int kPeSeedWps = ABM_PE_SEED_WPS, kPeMateWps = ABM_PE_MATE_WPS;

size_t pe_seed_lds_bytes(u32 W, u32 WB, u32 max_len, u32 cap) {
  const u32 MB = (max_len + kPlaneBlock - 1) / kPlaneBlock;
  const size_t b = static_cast<size_t>(8) * W * 8 + static_cast<size_t>(8) * WB * 8 + static_cast<size_t>(8) * MB * 4 * 8 +
                   (static_cast<size_t>(8) << kPosCacheBits) + 64 * 4 + static_cast<size_t>(cap) * 4 /* heap */ +
                   static_cast<size_t>(cap) * 4 + static_cast<size_t>(cap + (cap & 1u)) * 2 /* one list */ + 2 * 128 * 4 + 64 * 2;
  return (b + 15) & ~static_cast<size_t>(15);
}
size_t pe_mate_lds_bytes(u32 W, u32 GW, u32 cig_stride, u32 max_len, double valid_frac, u32 cap, bool big) {
  size_t b = static_cast<size_t>(8) * W * 8 + (static_cast<size_t>(8) << kPosCacheBits) + static_cast<size_t>(kMaxJobs) * GW * 8 +
             static_cast<size_t>(cig_stride) * 4 + 3 * kSeCap * 4 + 64 * 4 + 64 * 2;
  if (!big) b += static_cast<size_t>(cap) * (4 + 2 * 4 + 4 * 2);
  b += tb_extra_bytes(GW, max_len, valid_frac);
  return (b + 15) & ~static_cast<size_t>(15);
}
static int resident(const void *fn, size_t lds) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, lds) != hipSuccess) return 0;
  return per_cu * prop.multiProcessorCount;
}
int pe_seed_resident_waves(size_t lds, bool coop) {
  return coop ? resident(reinterpret_cast<const void *>(map_pe_kernel<false, false, true, kPeSeedWps, false, kSeed>), lds)
              : resident(reinterpret_cast<const void *>(map_pe_kernel<false, false, false, kPeSeedWps, false, kSeed>), lds);
}
int pe_mate_resident_waves(size_t lds, bool big) {
  return big ? resident(reinterpret_cast<const void *>(map_pe_kernel<true, false, false, kPeMateWps, false, kMate>), lds)
             : resident(reinterpret_cast<const void *>(map_pe_kernel<false, false, false, kPeMateWps, false, kMate>), lds);
}
hipError_t launch_pe_seed(const PeArgs &a, size_t lds, u32 grid, bool timed, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (pe_records(a)) {
    if (timed) hipLaunchKernelGGL((map_pe_kernel<false, true, true, kPeSeedWps, false, kSeed, true>), dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_pe_kernel<false, false, true, kPeSeedWps, false, kSeed, true>), dim3(grid), dim3(64), lds, st, a);
  }
  else if (a.G != 0) {
    if (timed) hipLaunchKernelGGL((map_pe_kernel<false, true, true, kPeSeedWps, false, kSeed>), dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_pe_kernel<false, false, true, kPeSeedWps, false, kSeed>), dim3(grid), dim3(64), lds, st, a);
  }
  else {
    if (timed) hipLaunchKernelGGL((map_pe_kernel<false, true, false, kPeSeedWps, false, kSeed>), dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_pe_kernel<false, false, false, kPeSeedWps, false, kSeed>), dim3(grid), dim3(64), lds, st, a);
  }
  return hipGetLastError();
}
hipError_t launch_pe_mate(const PeArgs &a, size_t lds, u32 grid, bool big, bool timed, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (big) {
    if (timed) hipLaunchKernelGGL((map_pe_kernel<true, true, false, kPeMateWps, false, kMate>), dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_pe_kernel<true, false, false, kPeMateWps, false, kMate>), dim3(grid), dim3(64), lds, st, a);
  }
  else {
    if (timed) hipLaunchKernelGGL((map_pe_kernel<false, true, false, kPeMateWps, false, kMate>), dim3(grid), dim3(64), lds, st, a);
    else hipLaunchKernelGGL((map_pe_kernel<false, false, false, kPeMateWps, false, kMate>), dim3(grid), dim3(64), lds, st, a);
  }
  return hipGetLastError();
}

}  // namespace abm
