// abismal_amd device-side building blocks shared by the single-end and
// paired-end mapping kernels (gfx950, one wavefront per read / pair).
#pragma once
#include "abm_kernels.hpp"

namespace abm {

// =============================================================================
// Wave-resident single-end candidate set (se_candidates, src/abismal.cpp:334-449).
// The heap array lives across lanes: lane k holds heap slot k as one packed key
// diffs*256 + payload_slot; (flags,pos) payloads never move -- lane p holds the
// payload whose slot number is p.  A sift step therefore moves one register.
// =============================================================================
struct SeSet {
  static constexpr bool kFifo = true;
  static constexpr bool kAppend = false;
  int hk;   // per lane k: heap[k] = diffs*256 + payload slot
  u32 pf;   // per lane p: flags of payload p
  u32 pp;   // per lane p: pos of payload p
  int sz, cutoff, good_cutoff;
  int best_d;
  u32 best_f, best_p;
  bool sure_ambig;

  __device__ __forceinline__ static int key_d(int k) { return k >> 8; }
  __device__ __forceinline__ void begin_read(u32 readlen) {
    const int worst = static_cast<i16>(0.4 * readlen);  // se_element::reset(readlen), :292-296
    hk = worst * 256; pf = 0; pp = 0;  // sentinel: heap[0] -> payload 0 = {pos 0}
    sz = 1;
    cutoff = worst;
    good_cutoff = static_cast<i16>(readlen / 10u);
    best_d = worst; best_f = 0; best_p = 0;
    sure_ambig = false;
  }
  __device__ __forceinline__ int top_d() const { return key_d(rdlane(hk, 0)); }
  // libstdc++ __push_heap with `key` entering at `hole`, comparator diffs<
  __device__ __forceinline__ void sift_up(int hole, int key) {
    int parent = (hole - 1) / 2;
    while (hole > 0) {
      const int pk = rdlane(hk, parent);
      if (!(key_d(pk) < key_d(key))) break;
      wrlane(hk, hole, pk);
      hole = parent;
      parent = (hole - 1) / 2;
    }
    wrlane(hk, hole, key);
  }
  // libstdc++ pop_heap on [0,n): returns the payload slot of the evicted maximum;
  // the caller overwrites heap[n-1] and pushes, so only __adjust_heap of the
  // displaced last element is performed here
  __device__ __forceinline__ int pop_max(int n) {
    const int len = n - 1;
    const int freed = rdlane(hk, 0) & 255;
    const int vk = rdlane(hk, len);
    int hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      int sk = rdlane(hk, second);
      const int lk = rdlane(hk, second - 1);
      if (key_d(sk) < key_d(lk)) { --second; sk = lk; }
      wrlane(hk, hole, sk);
      hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      wrlane(hk, hole, rdlane(hk, second - 1));
      hole = second - 1;
    }
    sift_up(hole, vk);
    return freed;
  }
  // pop_heap + overwrite + push_heap on the FULL set (50 elements) in straight-line code: the evicted maximum's
  // payload slot takes (f, p) and the key d enters.  Same result as pop_max + sift_up above, derived from
  // libstdc++'s __adjust_heap / __push_heap:
  //  * the hole left by the root walks down 0 -> p1 -> .. -> pm, always to the right child unless it compares
  //    less than the left one, while the node has two children in the 49-element heap (node < 24): every sibling
  //    comparison is made at once (one DPP move + one ballot) and the walk is bit tests on the result;
  //  * the displaced last element vk then rises from pm while the element above compares less -- and the element
  //    above hole p(i) at that point is the one that started in p(i) -- so the net effect on the path is: nodes
  //    above p(j) take their path child's key, p(j) takes vk, nodes below keep theirs, where j is the deepest
  //    path index >= 1 whose original key does not compare less than vk (0 if there is none);
  //  * the newcomer enters at 49 and rises along 24, 11, 5, 2, 0 while the parent compares less.
  // Returns the key left at the root.
  __device__ __forceinline__ int replace_top(int d, u32 f, u32 p) {
    static_assert(kSeCap == 50, "paths derived for a 50-element heap");
    const int left = __builtin_amdgcn_update_dpp(0, hk, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    const u64 right_wins = __ballot(!(key_d(hk) < key_d(left)));  // bit r, r even: the hole moves to r rather than r - 1
    auto down = [&](int node) { const int r = 2 * node + 2; return ((right_wins >> r) & 1ull) ? r : r - 1; };
    const int root = rdlane(hk, 0), vk = rdlane(hk, 49);
    const int p1 = down(0), p2 = down(p1), p3 = down(p2), p4 = down(p3);
    const bool five = p4 < 24;
    const int p5 = five ? down(p4) : 63;  // (lane 63 is not part of the set: a harmless stand-in)
    const int k1 = rdlane(hk, p1), k2 = rdlane(hk, p2), k3 = rdlane(hk, p3), k4 = rdlane(hk, p4), k5 = rdlane(hk, p5);
    const int dv = key_d(vk);
    int j = 0;
    if (key_d(k1) >= dv) j = 1;
    if (key_d(k2) >= dv) j = 2;
    if (key_d(k3) >= dv) j = 3;
    if (key_d(k4) >= dv) j = 4;
    if (five && key_d(k5) >= dv) j = 5;
    wrlane(hk, 0, j == 0 ? vk : k1);
    wrlane(hk, p1, j > 1 ? k2 : (j == 1 ? vk : k1));
    wrlane(hk, p2, j > 2 ? k3 : (j == 2 ? vk : k2));
    wrlane(hk, p3, j > 3 ? k4 : (j == 3 ? vk : k3));
    wrlane(hk, p4, j > 4 ? k5 : (j == 4 ? vk : k4));
    wrlane(hk, p5, j == 5 ? vk : k5);
    const int slot = root & 255;
    wrlane(pf, slot, f);
    wrlane(pp, slot, p);
    const int nk = d * 256 + slot;
    const int c24 = rdlane(hk, 24), c11 = rdlane(hk, 11), c5 = rdlane(hk, 5), c2 = rdlane(hk, 2), c0 = rdlane(hk, 0);
    const bool b24 = key_d(c24) < d, b11 = b24 && key_d(c11) < d, b5 = b11 && key_d(c5) < d, b2 = b5 && key_d(c2) < d,
               b0 = b2 && key_d(c0) < d;
    wrlane(hk, 49, b24 ? c24 : nk);
    wrlane(hk, 24, b24 ? (b11 ? c11 : nk) : c24);
    wrlane(hk, 11, b11 ? (b5 ? c5 : nk) : c11);
    wrlane(hk, 5, b5 ? (b2 ? c2 : nk) : c5);
    wrlane(hk, 2, b2 ? (b0 ? c0 : nk) : c2);
    const int top = b0 ? nk : c0;
    wrlane(hk, 0, top);
    return top;
  }

  // A RUN of survivors that tie with the cutoff while the set is full and heap[0], heap[24] and heap[49] all sit at
  // the cutoff c (tandem repeats, homopolymer reads: hundreds of thousands of such updates per read).  Each is
  // pop_heap + overwrite + push_heap as in replace_top, and under these conditions that is a shift register:
  //  * the displaced last element heap[49] has distance c, the maximum, so it comes to rest at the deepest node of
  //    the hole's path that holds a c (j in replace_top; the c's of a path are a prefix of it, the heap being a
  //    heap), and everything above moves up one node: the path's c-prefix p0..pj takes p1..pj, heap[49];
  //  * the newcomer (distance c) enters at 49 and stays there, its parent 24 not comparing less;
  //  * every node keeps its distance, so the sibling comparisons -- hence the path -- are the same for the next tie.
  // So a tie is: evict heap[0]'s payload, shift keys along the chain p0 <- p1 <- .. <- pj <- 49, put the newcomer
  // (with the evicted payload slot) at 49.  After n = chain length updates every old element is gone and the payload
  // slots are back in the same places, so of a run of m only the last n + m % n (at most 13) need to be played.
  // `ties` = lanes whose candidate has distance c; the leading ones of `todo` that are ties are consumed.
  // Returns how many updates were applied (0: conditions not met, nothing done).
  __device__ __forceinline__ int tie_run(u64 &todo, u64 ties, u32 cand_pos, u32 f) {
    static_assert(kSeCap == 50, "paths derived for a 50-element heap");
    const int lane = lane_id();
    const int c = cutoff;
    if (sz != static_cast<int>(kSeCap) || key_d(rdlane(hk, 0)) != c || key_d(rdlane(hk, 24)) != c || key_d(rdlane(hk, 49)) != c)
      return 0;
    const u64 brk = todo & ~ties;
    u64 run = brk ? (todo & ((brk & (0 - brk)) - 1)) : todo;
    if (run == 0) return 0;
    const int m = __popcll(run);
    todo &= ~run;
    // the hole's path and its c-prefix
    const int left = __builtin_amdgcn_update_dpp(0, hk, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    const u64 right_wins = __ballot(!(key_d(hk) < key_d(left)));
    auto down = [&](int node) { const int r = 2 * node + 2; return ((right_wins >> r) & 1ull) ? r : r - 1; };
    const int p1 = down(0), p2 = down(p1), p3 = down(p2), p4 = down(p3);
    const int p5 = p4 < 24 ? down(p4) : 63;
    const bool e1 = key_d(rdlane(hk, p1)) == c, e2 = e1 && key_d(rdlane(hk, p2)) == c, e3 = e2 && key_d(rdlane(hk, p3)) == c,
               e4 = e3 && key_d(rdlane(hk, p4)) == c, e5 = e4 && p4 < 24 && key_d(rdlane(hk, p5)) == c;
    const int n = 2 + (e1 ? 1 : 0) + (e2 ? 1 : 0) + (e3 ? 1 : 0) + (e4 ? 1 : 0) + (e5 ? 1 : 0);  // chain length incl. 49
    // where each lane's key comes from in one shift
    int src = lane;
    wrlane(src, 0, e1 ? p1 : 49);
    if (e1) wrlane(src, p1, e2 ? p2 : 49);
    if (e2) wrlane(src, p2, e3 ? p3 : 49);
    if (e3) wrlane(src, p3, e4 ? p4 : 49);
    if (e4) wrlane(src, p4, e5 ? p5 : 49);
    if (e5) wrlane(src, p5, 49);
    // the survivors that leave a trace: the last n + m % n of the run
    int play = m;
    while (play >= 2 * n) play -= n;
    u64 last = 0, rest = run;
    for (int k = 0; k < play; ++k) {
      const int top = 63 - __builtin_clzll(rest);
      last |= 1ull << top;
      rest &= ~(1ull << top);
    }
    const int c256 = c * 256;
    while (last) {
      const int l = __builtin_ctzll(last);
      last &= last - 1;
      const int slot = rdlane(hk, 0) & 255;
      const u32 p = rdlane(cand_pos, l);
      hk = __builtin_amdgcn_ds_bpermute(src << 2, hk);
      wrlane(hk, 49, c256 + slot);
      wrlane(pp, slot, p);
      wrlane(pf, slot, f);
    }
    return m;
  }

  // se_candidates::update, :394-404
  __device__ __forceinline__ void admit(bool specific, int d, u32 f, u32 p) {
    if (d == 0) {
      if (best_p == 0) { best_d = 0; best_f = f; best_p = p; }
      else if (p != best_p || f != best_f) best_f |= kFlagAmbig;
    }
    int top;
    if (d != 0 && sz == static_cast<int>(kSeCap)) top = key_d(replace_top(d, f, p));
    else {
      if (d != 0) {
        const int slot = sz++;
        wrlane(pf, slot, f);
        wrlane(pp, slot, p);
        sift_up(sz - 1, d * 256 + slot);
      }
      top = top_d();
    }
    sure_ambig = (best_f & kFlagAmbig) && best_d == 0;
    cutoff = specific ? min(cutoff, top) : top;
  }
};

// Per-wave LDS carve-up
struct WaveLds {
  u64 *qpk;    // [4][W] packed encodings
  u64 *qbits;  // [4][WB] 2-letter bit strings, bit j = bit2(nibble j), 1 past the end
  u64 *qmask;  // [4][MB][4] per block of 64 read bases: which of them admit genome code 0, 1, 2, 3 (cooperative filter)
  u16 *mark;   // [64] (paired-end mating)
  u32 *smark;  // [128] seed pass: where the segments of a flattened block begin inside the current 128-candidate step (locate128)
  u32 *sdelta; // [128] seed pass: per segment, index-array position minus candidate number of its first entry
  u32 *ctmp;   // [ctmp_cap] reversed CIGAR scratch
  u8 *tb;      // traceback bytes
  u64 *gwin;   // [kMaxJobs][GW] genome windows of the alignments in flight
  u32 *jpos;   // [kSeCap] alignment job list: position
  u32 *jdf;    // [kSeCap] alignment job list: diffs<<16 | flags
  int *lbest;  // [64]
  u64 *pcache; // [1 << kPosCacheBits] candidate cache: pos | diffs<<32 | max-prefix-diffs<<48
  u16 *hres;   // [128] distances of the candidates of one step (cooperative window loads)
  u32 W, WB, GW, MB;
  u32 G;       // lanes that share one candidate's window (4 or 8), 0 = one lane per window
  u32 max_jobs;  // window slots in gwin: kMaxJobs, or 2 in the long-read kernel (its bands are 61 lanes wide: one slot of lanes)
};
constexpr u32 kPosCacheBits = 8;
constexpr u32 kMaxJobs = 21;  // 64 lanes / narrowest band (3)

__device__ __forceinline__ u32 q_nibble(const u64 *qpk, u32 k) {
  return static_cast<u32>(qpk[k >> 4] >> ((k & 15u) << 2)) & 15u;
}

// 16 consecutive read nibbles starting at base i (nibbles at or past L read as 0)
__device__ __forceinline__ u64 q_window16(const u64 *qpk, u32 W, u32 i, u32 L) {
  const u32 w = i >> 4, s = (i & 15u) << 2;
  u64 x = qpk[w] >> s;
  if (s && w + 1 < W) x |= qpk[w + 1] << (64 - s);
  const u32 have = i < L ? min(16u, L - i) : 0u;
  if (have < 16) x &= (have == 0 ? 0ull : ((1ull << (have << 2)) - 1));
  return x;
}

// The genome letter at position q as the narrowing loops need it: its 2-letter bit / its 3-letter sort symbol
// (from the nibble array: a probe can land on an N, which the bit planes cannot express).
__device__ __forceinline__ u32 genome_bit2(const DevIndex &ix, u64 q) { return bit2(gnib(ix.genome, q)); }
__device__ __forceinline__ u32 genome_sortsym3(const DevIndex &ix, u64 q, bool g_to_a) { return sortsym3(gnib(ix.genome, q), g_to_a); }

// find_candidates (src/abismal.cpp:1163-1194) and find_candidates_three (:1214-1259) of one seed offset, advanced
// TOGETHER.  Both narrow a bucket letter by letter with std::lower_bound bisections (one per letter on the 2-letter
// table; two per letter -- first symbol >= mid, first symbol >= top -- on a 3-letter table, which start at the same
// midpoint and share probes until their paths part).  Every round issues the next probe of the 2-letter bisection
// and of both 3-letter bisections before any of them is waited for (three dependent chains of index-entry ->
// genome-letter loads in flight per lane instead of one after the other); each bisection sees exactly
// std::lower_bound's probe sequence, so the boundaries are the reference's even where a bucket's tail is unsorted.
// A chain with nothing to probe in a round reads entry 0 of its table and ignores it.  The read's letters of the
// 2-letter chain come from its bit string qb (n_bits of it; 1 beyond): the loop ends at p == limit, and for reads
// of 44-46 bases limit = L - i is BELOW the starting length for the last offsets, so the reference keeps extending
// past the end of the read, through whatever its reused buffer holds there (the ghost bits put into qb: ghost_bits).
//
// REC: the letters come from the window records (DevIndex::wrec): entry k's record holds the genome from wrec_back bases
// before the entry's position, so the letter p bases after it is bit wrec_back + p of record k -- ONE load that does not
// wait for the index entry, where the nibble array takes two in a row (the entry, then the letter behind it).  Two bits
// cannot say N: the record's one spare base (its last; the filter never reaches it) says whether the record's stretch
// holds a blank nibble, and such a probe -- like one beyond the record's reach, which only a read of 44-46 bases
// extending past its end can ask for -- goes the old way.
__device__ __forceinline__ u32 record_nibble(const DevIndex &ix, const u32 *__restrict__ tbl, u32 rec0, bool live, u32 k, u32 p) {
  const u32 at = ix.wrec_back + p, last = ix.wrec_blocks * 64u - 1u;
  const bool in_reach = at < last;
  const u64 *r = ix.wrec + 2ull * (live ? static_cast<u64>(k + rec0) * ix.wrec_blocks : 0ull);
  const u32 blk = in_reach ? at >> 6 : 0u;
  const u64 lo = r[2 * blk], hi = r[2 * blk + 1], flag = r[2 * (ix.wrec_blocks - 1)];
  u32 nib = 1u << ((static_cast<u32>(lo >> (at & 63u)) & 1u) | ((static_cast<u32>(hi >> (at & 63u)) & 1u) << 1));
  if (live && (!in_reach || (flag >> 63))) nib = gnib(ix.genome, static_cast<u64>(tbl[k]) + p);
  return nib;
}
template <bool REC = false>
__device__ __forceinline__ void narrow_both(const DevIndex &ix, const u32 *__restrict__ tbl3, bool g_to_a, const u64 *qb,
                                            u32 n_bits, const u64 *qpk, u32 qbase, u32 limit, u32 maxc, bool run2, u32 &lo2,
                                            u32 &hi2, u32 &len2, bool run3, u32 &lo3, u32 &hi3, u32 &len3, u32 &probes, u32 rec3 = 0) {
  // (len2 / len3 on entry: the letters each range has been narrowed by already -- the key weights, or more where the
  // seed-extension tables have taken the first steps; run2 / run3 false: the table has finished that chain, nothing
  // is done to its range)
  const u32 *__restrict__ tbl2 = ix.index;
  const u32 mid_sym = g_to_a ? 2u : 1u, top_sym = g_to_a ? 8u : 4u;
  u32 pA = len2, ploA = lo2, phiA = hi2, blA = 0;
  int bnA = 0;
  auto goA = [&]() { return pA != limit && (hi2 - lo2) > maxc && qbase + pA < n_bits + 4096u; };
  bool actA = run2 && goA();
  if (actA) { blA = lo2; bnA = static_cast<int>(hi2 - lo2); }
  u32 pB = len3, ploB = lo3, phiB = hi3, l1 = 0, l2 = 0;
  int n1 = 0, n2 = 0;
  auto goB = [&]() { return pB != limit && (hi3 - lo3) > maxc; };
  bool actB = run3 && goB();
  if (actB) { l1 = l2 = lo3; n1 = n2 = static_cast<int>(hi3 - lo3); }
  while (actA || actB) {
    const int hA = bnA >> 1, h1 = n1 >> 1, h2 = n2 >> 1;
    const u32 kA = blA + static_cast<u32>(hA), m1 = l1 + static_cast<u32>(h1), m2 = l2 + static_cast<u32>(h2);
    const bool dA = actA, d1 = actB && n1 > 0, d2 = actB && n2 > 0 && !(d1 && m2 == m1);
    u32 sA, s1, s2;
    if constexpr (REC) {
      sA = bit2(record_nibble(ix, tbl2, 0u, dA, kA, pA));
      s1 = sortsym3(record_nibble(ix, tbl3, rec3, d1, m1, pB), g_to_a);
      s2 = sortsym3(record_nibble(ix, tbl3, rec3, d2, m2, pB), g_to_a);
    }
    else {
      const u32 eA = tbl2[dA ? kA : 0u], e1 = tbl3[d1 ? m1 : 0u], e2 = tbl3[d2 ? m2 : 0u];
      sA = genome_bit2(ix, static_cast<u64>(eA) + pA);
      s1 = genome_sortsym3(ix, static_cast<u64>(e1) + pB, g_to_a);
      s2 = genome_sortsym3(ix, static_cast<u64>(e2) + pB, g_to_a);
    }
    probes += (dA ? 1u : 0u) + (d1 ? 1u : 0u) + (d2 ? 1u : 0u);
    if (actA) {  // one step of first_not (std::lower_bound's probe sequence), then find_candidates' update
      if (sA < 1u) { blA = kA + 1; bnA -= hA + 1; } else bnA = hA;
      if (bnA <= 0) {
        const u32 at = qbase + pA;
        const bool one = at < n_bits ? ((qb[at >> 6] >> (at & 63u)) & 1ull) != 0 : true;
        lo2 = one ? blA : lo2;
        hi2 = one ? hi2 : blA;
        ++pA;
        actA = goA();
        if (actA) { ploA = lo2; phiA = hi2; blA = lo2; bnA = static_cast<int>(hi2 - lo2); }
      }
    }
    if (actB) {
      if (n2 > 0 && n1 > 0 && m2 == m1) s2 = s1;
      if (n1 > 0) { if (s1 < mid_sym) { l1 = m1 + 1; n1 -= h1 + 1; } else n1 = h1; }
      if (n2 > 0) { if (s2 < top_sym) { l2 = m2 + 1; n2 -= h2 + 1; } else n2 = h2; }
      if (n1 <= 0 && n2 <= 0) {
        const u32 sym = sortsym3(q_nibble(qpk, qbase + pB), g_to_a);
        const u32 nlo = sym == 0 ? lo3 : (sym == mid_sym ? l1 : l2), nhi = sym == 0 ? l1 : (sym == mid_sym ? l2 : hi3);
        lo3 = nlo;
        hi3 = nhi;
        ++pB;
        actB = goB();
        if (actB) { ploB = lo3; phiB = hi3; l1 = l2 = lo3; n1 = n2 = static_cast<int>(hi3 - lo3); }
      }
    }
  }
  if (run2 && lo2 == hi2) { --pA; lo2 = ploA; hi2 = phiA; }
  if (run3 && lo3 == hi3) { --pB; lo3 = ploB; hi3 = phiB; }
  len2 = pA;
  len3 = pB;
}

__device__ __forceinline__ u32 every_fourth_bit(u64 t) {  // bits 0, 4, 8, ... of t -> 16 contiguous bits
  t &= 0x1111111111111111ull;
  t = (t | (t >> 3)) & 0x0303030303030303ull;
  t = (t | (t >> 6)) & 0x000F000F000F000Full;
  t = (t | (t >> 12)) & 0x000000FF000000FFull;
  t = (t | (t >> 24)) & 0xFFFFull;
  return static_cast<u32>(t);
}

// ---- direct narrowing of big buckets ---------------------------------------------------------------------------
// find_candidates / find_candidates_three extend a seed letter by letter while its range holds more than
// max_candidates entries, each letter a bisection of the range: for a seed inside a high-copy repeat that is dozens of
// letters times log2(range) dependent probes, each a random line -- a quarter of the kernel's line requests at hg38
// scale.  The loop's outcome can be had with far fewer.  Let m(k) be the number of letters, from the current depth p0
// on and at most T = limit - p0, that entry k shares with the read.  The range after t more letters is {k: m(k) >= t};
// the loop stops at the first t whose range holds at most maxc entries (stepping back one letter if that range is
// empty) or at T.  With V the (maxc + 1)-th largest m and Mmax the largest, that is t_f = V + 1 if V < T and Mmax > V,
// else V.  A bucket is sorted by those very letters (src/AbismalIndex.cpp:857-978, 256 of them), so m rises towards the
// place `ins` where the read would be inserted and falls after it: ins costs one bisection with whole-string
// comparisons (64 letters per step from the bit planes), V is the best min(m(s), m(s + maxc)) over windows of
// maxc + 1 entries around ins (a bisection on a monotone difference), and the final range's ends are two more
// bisections -- log2(range) + ~30 probes in all, whatever the depth.  Exact wherever the bisections are: same sortedness
// they rely on; entries within reach of an N (which the planes cannot express: nmap) send their lane back to the
// letter-by-letter loop.  mode 0: 2-letter table; 1 / 2: 3-letter table, C->T / G->A alphabet (letter classes as
// find_candidates_three orders them: below mid, mid, top and above).
__device__ __forceinline__ void classes16(int mode, u64 x, u32 &c1, u32 &c2) {
  const u64 y = mode == 1 ? x : x >> 1;  // sym = nib & 5: 1 -> class 1, 4 or 5 -> class 2;  sym = nib & 10: 2 -> class 1, 8 or 10 -> class 2
  c2 = every_fourth_bit(y >> 2);
  c1 = every_fourth_bit(y & ~(y >> 2));
}
// the read's letter classes for positions [s, s + 64) as two bit strings (2-letter table: its bit string and 0)
__device__ __forceinline__ void read_classes64(int mode, const u64 *qb, u32 n_bits, const u64 *qpk, u32 W, u32 L, u32 s, u64 &r1, u64 &r2) {
  if (mode == 0) {
    const u32 w = s >> 6, sh = s & 63u;
    const u64 a = s < n_bits ? qb[w] : ~0ull, b = s + 64 < n_bits ? qb[w + 1] : ~0ull;
    r1 = sh ? (a >> sh) | (b << (64 - sh)) : a;
    r2 = 0;
    return;
  }
  r1 = r2 = 0;
#pragma unroll
  for (u32 j = 0; j < 4; ++j) {
    u32 c1, c2;
    classes16(mode, q_window16(qpk, W, s + 16 * j, L), c1, c2);
    r1 |= static_cast<u64>(c1) << (16 * j);
    r2 |= static_cast<u64>(c2) << (16 * j);
  }
}
// (Inlined into the single-end kernel this search cost more than it saved: that kernel runs at the register budget of
// its filter loop, and 200 bytes per lane of scratch instead of 100 made every read slower -- 729 -> 780 ms per 10 M
// reads although the probes fell from 585 to 241 per read; profiles/r03_exp_direct_narrowing_*.  The pair kernels, built
// for 128 / 170 registers and bound by line requests of which a quarter are narrowing probes, are where it runs.)
struct DirectArgs { const u64 *planes0; const u32 *nmap; const u64 *qb; const u64 *qpk; u32 n_bits, W, L, maxc; };
struct DirectRange { u32 lo, hi, len, probes; bool ok; };
__device__ __forceinline__ DirectRange narrow_direct(int mode, DirectArgs da, const u32 *__restrict__ tbl, u32 qbase, u32 limit, u32 lo, u32 hi, u32 len) {
  const u64 *qb = da.qb, *qpk = da.qpk;
  const u32 n_bits = da.n_bits, W = da.W, L = da.L, maxc = da.maxc;
  u32 probes = 0;
  const u32 p0 = len, T = limit - p0, s0 = qbase + p0;
  u64 R1, R2;
  read_classes64(mode, qb, n_bits, qpk, W, L, s0, R1, R2);
  bool bad = false;
  // m(k) and how entry k compares with the read (cmp < 0: before it)
  auto eval = [&](u32 k, int &cmp) -> u32 {
    ++probes;
    const u64 q = static_cast<u64>(tbl[k]) + p0;
    if ((da.nmap[q >> (kPlaneChunkBits + 5)] >> ((q >> kPlaneChunkBits) & 31u)) & 1u) bad = true;
    for (u32 w = 0; 64 * w < T; ++w) {
      const u64 at = q + 64 * w;
      const u64 *b = da.planes0 + 2 * (at / kPlaneBlock);
      const u32 sh = static_cast<u32>(at % kPlaneBlock);
      u64 gl = b[0] >> sh, gh = b[1] >> sh;
      if (sh) { gl |= b[2] << (64 - sh); gh |= b[3] << (64 - sh); }
      u64 r1 = R1, r2 = R2;
      if (w) read_classes64(mode, qb, n_bits, qpk, W, L, s0 + 64 * w, r1, r2);
      const u64 g1 = mode == 0 ? gl : (mode == 1 ? ~gl & ~gh : gl & ~gh);
      const u64 g2 = mode == 0 ? 0ull : (mode == 1 ? ~gl & gh : gl & gh);
      u64 x = (g1 ^ r1) | (g2 ^ r2);
      const u32 have = T - 64 * w;
      if (have < 64) x &= (1ull << have) - 1;
      if (x) {
        const u32 j = static_cast<u32>(__builtin_ctzll(x));
        const u32 gv = static_cast<u32>((g1 >> j) & 1ull) + 2u * static_cast<u32>((g2 >> j) & 1ull);
        const u32 rv = static_cast<u32>((r1 >> j) & 1ull) + 2u * static_cast<u32>((r2 >> j) & 1ull);
        cmp = gv < rv ? -1 : 1;
        return 64 * w + j;
      }
    }
    cmp = 0;
    return T;
  };
  // One loop, one probe per turn, whatever step of the search a lane is at (lanes of a wave are at different ones):
  //  kIns   bisection for `ins`, where the read would be inserted
  //  kWinL / kWinR  bisection for the first window start s1 with m(s1) >= m(s1 + maxc): m(s) - m(s + maxc) is <= 0 for
  //         windows left of ins, >= 0 from ins on and monotone in between, so the best window is at the sign change
  //  kV1 / kV2  V = max(m(s1 + maxc), m(s1 - 1)), whichever exist
  //  kM1 / kM2  the largest m: next to ins
  //  kLeft / kRight  the ends of {k: m(k) >= tf} (m does not fall towards ins from either side)
  enum { kIns, kWinL, kWinR, kV1, kV2, kM1, kM2, kTf, kLeft, kRight, kDone };
  const u32 d_lo = lo, d_hi = hi - 1 - maxc;  // window starts: s in [d_lo, d_hi]  (hi - lo > maxc)
  int phase = kIns;
  u32 base = lo, cnt = hi - lo, ins = lo, s1 = 0, sx = 0, ml = 0, V = 0, mmax = 0, tf = 0, left = lo, right = hi, r_end = hi;
  while (phase != kDone) {
    // transitions that need no probe
    if (phase == kIns && cnt == 0) {
      ins = base;
      const u32 a = ins > maxc + 1 ? max(d_lo, ins - maxc - 1) : d_lo, b_end = min(d_hi, ins);
      base = min(a, b_end); cnt = b_end + 1 - base;
      phase = kWinL;
    }
    if (phase == kWinL && cnt == 0) { s1 = base; phase = kV1; }
    if (phase == kV1 && s1 > d_hi) phase = kV2;
    if (phase == kV2 && s1 <= d_lo) phase = kM1;
    if (phase == kM1 && ins <= lo) phase = kM2;
    if (phase == kM2 && ins >= hi) phase = kTf;
    if (phase == kTf) {
      tf = (V < T && mmax > V) ? V + 1 : V;
      u32 l0 = lo;
      r_end = hi;
      if (tf == V + 1) { l0 = ins > maxc ? max(lo, ins - maxc) : lo; r_end = min(hi, ins + maxc); }  // (at most maxc entries, all next to ins)
      if (tf == 0) { left = lo; right = hi; phase = kDone; }
      else { base = l0; cnt = ins - l0; phase = kLeft; }
    }
    if (phase == kLeft && cnt == 0) { left = base; base = ins; cnt = r_end - ins; phase = kRight; }
    if (phase == kRight && cnt == 0) { right = base; phase = kDone; }
    if (phase == kDone) break;
    // this turn's probe
    const u32 half = cnt >> 1;
    u32 k;
    if (phase == kIns || phase == kLeft || phase == kRight) k = base + half;
    else if (phase == kWinL) { sx = base + half; k = sx; }
    else if (phase == kWinR) k = sx + maxc;
    else if (phase == kV1) k = s1 + maxc;
    else if (phase == kV2) k = s1 - 1;
    else if (phase == kM1) k = ins - 1;
    else k = ins;
    int c;
    const u32 m = eval(k, c);
    if (phase == kIns) { if (c < 0) { base += half + 1; cnt -= half + 1; } else cnt = half; }
    else if (phase == kWinL) { ml = m; phase = kWinR; }
    else if (phase == kWinR) { if (ml < m) { base = sx + 1; cnt -= half + 1; } else cnt = half; phase = kWinL; }
    else if (phase == kV1) { V = max(V, m); phase = kV2; }
    else if (phase == kV2) { V = max(V, m); phase = kM1; }
    else if (phase == kM1) { mmax = max(mmax, m); phase = kM2; }
    else if (phase == kM2) { mmax = max(mmax, m); phase = kTf; }
    else if (phase == kLeft) { if (m < tf) { base += half + 1; cnt -= half + 1; } else cnt = half; }
    else { if (m >= tf) { base += half + 1; cnt -= half + 1; } else cnt = half; }
  }
  DirectRange out;
  out.lo = left; out.hi = right; out.len = p0 + tf; out.probes = probes; out.ok = !bad;
  return out;
}


// full_compare (src/abismal.cpp:1105-1122).  The reference adds one word's mismatches at a
// time and gives up as soon as the running sum exceeds the cutoff, so a candidate is admitted
// iff EVERY prefix sum stays within the cutoff, and then with the complete distance.  Both are
// returned: d (complete) and dmax (largest prefix sum; equals d unless an IUPAC genome letter
// makes a word's contribution negative).
__device__ __forceinline__ int hamming(const u64 *__restrict__ genome, const u64 *qpk, u32 nwords,
                                       u32 pos, int &dmax) {
  const u64 *g = genome + (pos >> 4);
  const u32 sh = (pos & 15u) << 2;
  int d = 0, m = 0;
  u64 g0 = g[0];
  for (u32 w = 0; w < nwords; ++w) {
    const u64 g1 = g[w + 1];
    const u64 win = (g0 >> sh) | ((g1 << (63 - sh)) << 1);
    d += 16 - __popcll(qpk[w] & win);
    m = max(m, d);
    g0 = g1;
  }
  dmax = static_cast<i16>(m);
  return static_cast<i16>(d);
}
__device__ __forceinline__ int hamming(const u64 *__restrict__ genome, const u64 *qpk, u32 nwords, u32 pos) {
  int unused;
  return hamming(genome, qpk, nwords, pos, unused);
}
// the same for two windows at once: the genome words are fetched two at a time (16-byte loads,
// half as many requests as word-by-word), eight words of each window in flight per step (a lane
// whose `want` flag is off issues no loads and its results are meaningless)
__device__ __forceinline__ void hamming2(const u64 *__restrict__ genome, const u64 *qpk, u32 nwords,
                                         u32 pos_a, bool want_a, u32 pos_b, bool want_b, int &d_a,
                                         int &dmax_a, int &d_b, int &dmax_b) {
  typedef u64 pair_t __attribute__((ext_vector_type(2), aligned(8)));
  const u64 *ga = genome + (pos_a >> 4), *gb = genome + (pos_b >> 4);
  const u32 sa = (pos_a & 15u) << 2, sb = (pos_b & 15u) << 2;
  int da = 0, ma = 0, db = 0, mb = 0;
  u64 last_a = 0, last_b = 0;  // word c-1 of each window, carried into the next step
  auto word = [&](u32 w, u64 a0, u64 a1, u64 b0, u64 b1) {
    const u64 q = qpk[w];
    da += 16 - __popcll(q & ((a0 >> sa) | ((a1 << (63 - sa)) << 1)));
    ma = max(ma, da);
    db += 16 - __popcll(q & ((b0 >> sb) | ((b1 << (63 - sb)) << 1)));
    mb = max(mb, db);
  };
  for (u32 c = 0; c <= nwords; c += 8) {
    u64 xa[8], xb[8];
#pragma unroll
    for (u32 p = 0; p < 4; ++p) {
      const bool in = c + 2 * p <= nwords;
      pair_t va = {0ull, 0ull}, vb = {0ull, 0ull};
      if (want_a && in) va = *reinterpret_cast<const pair_t *>(ga + c + 2 * p);
      if (want_b && in) vb = *reinterpret_cast<const pair_t *>(gb + c + 2 * p);
      xa[2 * p] = va.x; xa[2 * p + 1] = va.y;
      xb[2 * p] = vb.x; xb[2 * p + 1] = vb.y;
    }
    if (c > 0 && c - 1 < nwords) word(c - 1, last_a, xa[0], last_b, xb[0]);
#pragma unroll
    for (u32 j = 0; j < 7; ++j)
      if (c + j < nwords) word(c + j, xa[j], xa[j + 1], xb[j], xb[j + 1]);
    last_a = xa[7];
    last_b = xb[7];
  }
  d_a = static_cast<i16>(da); dmax_a = static_cast<i16>(ma);
  d_b = static_cast<i16>(db); dmax_b = static_cast<i16>(mb);
}

// ---- candidate filter, cooperative form ----------------------------------------------
// A lane that fetches its own candidate's window issues one load per word, and with ~20 waves
// sharing a CU's L1 the line has often been evicted again before the next word asks for it: a
// window then costs several L1 misses.  Here G lanes (4 for reads up to 192 bp, 8 up to 448 bp)
// share a candidate and fetch its window from the genome's bit planes (DevIndex::planes), one
// 16-byte block of 64 bases per lane: a whole window is covered by a single instruction whose
// lanes coalesce into one line request.  Each lane counts the mismatches of its 64 read bases (the
// bits past its block come from the next lane), the group adds up, and the sums travel through LDS
// to the lanes that own the candidates.  The 128 candidates of a step are slots 0..63 (lane's
// first) and 64..127 (lane's second); 64/G of them are fetched per round, kCoopRounds rounds in
// flight (few: registers are better spent on a fifth wave per SIMD).
__device__ __forceinline__ int dpp_row_shl1(int v) {  // lane i <- lane i+1 within a row of 16 (last lane: 0)
  return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, false);
}
__device__ __forceinline__ int group_sum(int v, u32 G) {  // sum over aligned groups of 4 or 8 lanes, in every lane
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm 1,0,3,2*/, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E /*quad_perm 2,3,0,1*/, 0xf, 0xf, false);
  if (G == 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141 /*row_half_mirror*/, 0xf, 0xf, false);
  return v;
}
#ifndef ABM_COOP_ROUNDS
#define ABM_COOP_ROUNDS 2  // measured best with 20 waves per CU (round 2, profiles/r02_experiments.md): 1, 2, 4, 8 -> 937, 923, 959, 1386 ms
#endif
constexpr u32 kCoopRounds = ABM_COOP_ROUNDS;  // rounds of window loads in flight per lane

// Per block of 64 read bases and per encoding: the four masks "read base j admits genome code c" (bit c of the
// read's nibble; everything past the read admits every code, like the 0xF padding of the packed words).
__device__ __forceinline__ void build_qmasks(const WaveLds &lds, u32 L) {
  const int lane = lane_id();
  const u32 nblk = (L + kPlaneBlock - 1) / kPlaneBlock;
  for (u32 e = 0; e < 4; ++e)
    for (u32 s = 0; s < nblk; ++s) {
      const u32 i = s * kPlaneBlock + static_cast<u32>(lane);
      const u32 nib = i < L ? q_nibble(lds.qpk + e * lds.W, i) : 15u;
      const u64 m0 = __ballot(nib & 1u), m1 = __ballot(nib & 2u), m2 = __ballot(nib & 4u), m3 = __ballot(nib & 8u);
      if (lane == 0) {
        u64 *q = lds.qmask + (e * lds.MB + s) * 4;
        q[0] = m0; q[1] = m1; q[2] = m2; q[3] = m3;
      }
    }
}

// Hamming distances of 128 candidates (two per lane: pos_a of lane c = candidate c, pos_b = candidate 64 + c)
// against the genome's bit planes (DevIndex::planes).  G lanes share a candidate: lane s of the group loads block
// (pos >> 6) + s of the window -- one 16-byte load, every window inside one 128-byte line of one of the two
// copies -- and counts the mismatches of read bases [64 s, 64 s + 64): the genome bits it needs beyond its own
// block arrive by DPP from the next lane.  mismatches = 64 - popcount(the mask of the code each genome base has).
// Identical to full_compare's sum over whole words for a one-hot genome (src/abismal.cpp:1093-1122; the early exit
// there changes a distance only when the hit is rejected anyway).
template <u32 kRounds = kCoopRounds>
__device__ __forceinline__ void hamming_planes(const DevIndex &ix, const WaveLds &lds, const u64 *qm, u32 L,
                                               u32 pos_a, bool want_a, u32 pos_b, bool want_b, int &d_a, int &d_b) {
  const int lane = lane_id();
  const u32 G = lds.G, sub = lane & (G - 1), grp = lane / G, per_round = 64 / G;
  const u64 wa = __ballot(want_a), wb = __ballot(want_b);
  const bool counts = sub * kPlaneBlock < L;  // this lane has read bases to compare
  u64 m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  if (counts) { const u64 *q = qm + sub * 4; m0 = q[0]; m1 = q[1]; m2 = q[2]; m3 = q[3]; }
  for (u32 pass = 0; pass * kRounds * per_round < 128; ++pass) {
    {  // a pass none of whose slots is wanted (most steps of an ordinary read hold a few candidates) costs nothing
      const u32 slot0 = pass * kRounds * per_round;
      const u64 w = (slot0 >= 64 ? wb : wa) >> (slot0 & 63u);
      if (kRounds * per_round < 64 && (w & ((1ull << (kRounds * per_round)) - 1)) == 0) continue;
      if (kRounds * per_round >= 64 && w == 0) continue;
    }
    u64 xl[kRounds], xh[kRounds];
    u64 shifts = 0;  // (pos & 63) of the rounds' candidates, eight bits each (up to eight rounds)
#pragma unroll
    for (u32 r = 0; r < kRounds; ++r) {
      const u32 slot = (pass * kRounds + r) * per_round + grp;  // < 128; a round lies entirely in one half
      const bool second = (pass * kRounds + r) * per_round >= 64;
      const u32 c = slot & 63u;
      const u32 cp = static_cast<u32>(__shfl(static_cast<int>(second ? pos_b : pos_a), static_cast<int>(c)));
      shifts |= static_cast<u64>(cp & 63u) << (8 * r);
      // (a group of four always fetches four blocks, 64 contiguous bytes: the memory pipeline merges the loads of a
      // full quad of lanes into one request, and those of a partly active quad not at all -- measured, 2.5 requests per
      // window against 1.2; a group of eight fetches the blocks its window has)
      const u32 b0 = cp / kPlaneBlock, b1 = G == 4 ? b0 + 3 : (cp + L - 1) / kPlaneBlock;
      const bool act = (((second ? wb : wa) >> c) & 1ull) && b0 + sub <= b1;
      // Every lane loads, unconditionally, so that the rounds' loads are issued back to back (a load under a
      // divergent branch is waited for inside it): a lane with nothing to fetch reads the array's first line.
      // What it gets is never looked at -- its bits could only reach read positions past the end, whose masks
      // admit every code, or candidates whose result is discarded.
      const u64 *g = act ? ix.planes[(b0 / kPlaneLineBlocks) != (b1 / kPlaneLineBlocks) ? 1 : 0] + 2 * static_cast<u64>(b0 + sub)
                         : ix.planes[0];
      xl[r] = g[0];
      xh[r] = g[1];
    }
#pragma unroll
    for (u32 r = 0; r < kRounds; ++r) {
      const u32 slot = (pass * kRounds + r) * per_round + grp;
      const u32 sh = static_cast<u32>(shifts >> (8 * r)) & 63u;
      const u64 nl = (static_cast<u64>(static_cast<u32>(dpp_row_shl1(static_cast<int>(xl[r] >> 32)))) << 32) |
                     static_cast<u32>(dpp_row_shl1(static_cast<int>(xl[r])));
      const u64 nh = (static_cast<u64>(static_cast<u32>(dpp_row_shl1(static_cast<int>(xh[r] >> 32)))) << 32) |
                     static_cast<u32>(dpp_row_shl1(static_cast<int>(xh[r])));
      const u64 gl = (xl[r] >> sh) | ((nl << (63 - sh)) << 1), gh = (xh[r] >> sh) | ((nh << (63 - sh)) << 1);
      const u64 match = (~gh & ((~gl & m0) | (gl & m1))) | (gh & ((~gl & m2) | (gl & m3)));
      int d = counts ? 64 - __popcll(match) : 0;
      d = group_sum(d, G);
      if (sub == 0) lds.hres[slot] = static_cast<u16>(d);
    }
  }
  wave_sync();
  d_a = static_cast<i16>(lds.hres[lane]);
  d_b = static_cast<i16>(lds.hres[64 + lane]);
  wave_sync();
}

// Reads of up to 128 bases: TWO lanes per candidate, each fetching two consecutive blocks (32 bytes: lane s of the
// pair takes blocks s and s + 1 of the window, so the middle block is asked for twice and costs nothing more) and
// counting 64 read bases against them -- no exchange between the lanes, and 32 candidates per round instead of 16:
// the kernel is bound by vector-instruction issue once its windows are single lines (76 % of the SIMDs' cycles),
// and this halves the filter's instructions per candidate.  One round (32 windows, 8 registers) in flight per pass.
__device__ __forceinline__ void hamming_planes_pairs(const DevIndex &ix, const WaveLds &lds, const u64 *qm, u32 L,
                                                     u32 pos_a, bool want_a, u32 pos_b, bool want_b, int &d_a, int &d_b) {
  const int lane = lane_id();
  const u32 sub = lane & 1u, grp = lane >> 1;
  const u64 wa = __ballot(want_a), wb = __ballot(want_b);
  const bool counts = sub * kPlaneBlock < L;
  u64 m0 = 0, m1 = 0, m2 = 0, m3 = 0;
  if (counts) { const u64 *q = qm + sub * 4; m0 = q[0]; m1 = q[1]; m2 = q[2]; m3 = q[3]; }
#pragma unroll 1
  for (u32 pass = 0; pass < 4; ++pass) {
    const u32 slot = pass * 32 + grp;  // < 128; passes 0-1 are the lanes' first candidates, 2-3 their second
    const bool second = pass >= 2;
    // (a pass none of whose 32 slots is wanted costs nothing: most steps of an ordinary read hold a few candidates)
    if (static_cast<u32>((second ? wb : wa) >> ((pass & 1u) * 32u)) == 0u) continue;
    const u32 c = slot & 63u;
    const u32 cp = static_cast<u32>(__shfl(static_cast<int>(second ? pos_b : pos_a), static_cast<int>(c)));
    const u32 sh = cp & 63u;
    const u32 b0 = cp / kPlaneBlock, b1 = b0 + 2;  // a window of up to 128 bases has at most three blocks
    const bool act = ((second ? wb : wa) >> c) & 1ull;
    // (unconditional loads, as in hamming_planes: a lane with nothing to fetch reads the array's first line)
    const u64 *g = act ? ix.planes[(b0 / kPlaneLineBlocks) != (b1 / kPlaneLineBlocks) ? 1 : 0] + 2 * static_cast<u64>(b0 + sub)
                       : ix.planes[0];
    const u64 xl = g[0], xh = g[1], nl = g[2], nh = g[3];
    const u64 gl = (xl >> sh) | ((nl << (63 - sh)) << 1), gh = (xh >> sh) | ((nh << (63 - sh)) << 1);
    const u64 match = (~gh & ((~gl & m0) | (gl & m1))) | (gh & ((~gl & m2) | (gl & m3)));
    int d = counts ? 64 - __popcll(match) : 0;
    d += __builtin_amdgcn_update_dpp(0, d, 0xB1 /*quad_perm 1,0,3,2*/, 0xf, 0xf, false);
    if (sub == 0) lds.hres[slot] = static_cast<u16>(d);
  }
  wave_sync();
  d_a = static_cast<i16>(lds.hres[lane]);
  d_b = static_cast<i16>(lds.hres[64 + lane]);
  wave_sync();
}

// Window records: ONE lane per candidate.  A record is a few blocks laid end to end, so a lane takes its own candidate's
// window with three (reads up to 128 bases) or four (up to 192) 16-byte loads of consecutive addresses -- the lanes of a
// segment read consecutive records -- and compares the read's 64-base blocks itself: no shuffles to hand candidates to
// groups of lanes, no sums through LDS, 64 windows in flight per half step instead of 32 or 16 -- 0.7 vector instructions
// per candidate of a 100-base read against 1.4 for the two-lane groups, 1 against 2.8 for the four-lane groups of a
// 150-base read.  (The bit planes need the groups: a window there is a random gather, which only a group's single load
// fetches in one request.)  a / b: the lane's first and (if `two`) second candidate of the step.  The blocks a lane loads
// past its candidate's record (at most one) only reach read positions past the end of the read, whose masks admit every code.
template <bool WIDE>
__device__ __forceinline__ void hamming_records_lane(const DevIndex &ix, const u64 *qm, u32 L, u32 rec_a, u32 x_a, bool want_a,
                                                     u32 rec_b, u32 x_b, bool want_b, bool two, int &d_a, int &d_b) {
  auto funnel = [](u64 lo, u64 hi, u32 sh) { return (lo >> sh) | ((hi << (63 - sh)) << 1); };
  auto mismatches = [&](u64 gl, u64 gh, const u64 *q) {
    const u64 match = (~gh & ((~gl & q[0]) | (gl & q[1]))) | (gh & ((~gl & q[2]) | (gl & q[3])));
    return 64 - __popcll(match);
  };
  if constexpr (!WIDE) {
    // both candidates' loads first: six registers each
    const u64 *ra = ix.wrec + (want_a ? 2 * (static_cast<u64>(rec_a) + x_a / kPlaneBlock) : 0ull);
    const u64 a0l = ra[0], a0h = ra[1], a1l = ra[2], a1h = ra[3], a2l = ra[4], a2h = ra[5];
    u64 b0l = 0, b0h = 0, b1l = 0, b1h = 0, b2l = 0, b2h = 0;
    if (two) {
      const u64 *rb = ix.wrec + (want_b ? 2 * (static_cast<u64>(rec_b) + x_b / kPlaneBlock) : 0ull);
      b0l = rb[0]; b0h = rb[1]; b1l = rb[2]; b1h = rb[3]; b2l = rb[4]; b2h = rb[5];
    }
    const bool wide = L > kPlaneBlock;  // (uniform: the read has a second block of masks)
    {
      const u32 sh = x_a & 63u;
      int d = mismatches(funnel(a0l, a1l, sh), funnel(a0h, a1h, sh), qm);
      if (wide) d += mismatches(funnel(a1l, a2l, sh), funnel(a1h, a2h, sh), qm + 4);
      d_a = d;
    }
    d_b = 0x7fff;
    if (two) {
      const u32 sh = x_b & 63u;
      int d = mismatches(funnel(b0l, b1l, sh), funnel(b0h, b1h, sh), qm);
      if (wide) d += mismatches(funnel(b1l, b2l, sh), funnel(b1h, b2h, sh), qm + 4);
      d_b = d;
    }
  }
  else {
    // up to three blocks of read: four blocks of record, one candidate after the other (eight registers of window)
    auto one = [&](u32 rec, u32 x, bool want) {
      const u64 *r = ix.wrec + (want ? 2 * (static_cast<u64>(rec) + x / kPlaneBlock) : 0ull);
      const u64 l0 = r[0], h0 = r[1], l1 = r[2], h1 = r[3], l2 = r[4], h2 = r[5], l3 = r[6], h3 = r[7];
      const u32 sh = x & 63u;
      int d = mismatches(funnel(l0, l1, sh), funnel(h0, h1, sh), qm);
      if (L > kPlaneBlock) d += mismatches(funnel(l1, l2, sh), funnel(h1, h2, sh), qm + 4);
      if (L > 2 * kPlaneBlock) d += mismatches(funnel(l2, l3, sh), funnel(h2, h3, sh), qm + 8);
      return d;
    };
    d_a = one(rec_a, x_a, want_a);
    d_b = two ? one(rec_b, x_b, want_b) : 0x7fff;
  }
}

// ---- flattened candidates of one 64-offset block ---------------------------------------
// Per lane (= seed offset g0 + lane): its checked 2-letter bucket [lo2, lo2 + na) and 3-letter bucket
// [lo3, lo3 + nb), laid end to end in the reference's visiting order (offset ascending, 2-letter
// before 3-letter, index order): candidate c of the block is entry start_a + ... of whichever
// segment contains c.  Segment number = 2 * lane + (1 for the 3-letter table).
struct Segs {
  u32 start_a, na, lo2, nb, lo3;
  __device__ __forceinline__ u32 start_b() const { return start_a + na; }
};
constexpr u32 kSegEpochLimit = (1u << 24) - 4;
// once per block: where each segment's entries lie relative to their candidate numbers
__device__ __forceinline__ void publish_segs(const WaveLds &lds, const Segs &sg) {
  typedef u32 u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = {sg.lo2 - sg.start_a, sg.lo3 - sg.start_b()};
  *reinterpret_cast<u32x2 *>(lds.sdelta + 2 * lane_id()) = v;
}
// Which segment each of the 128 candidates c0 + lane (a) and c0 + 64 + lane (b) belongs to, and where its index
// entry is.  Every segment that begins inside the step leaves a mark at its first candidate, a running maximum
// spreads it over the candidates that follow, and candidates before the first mark belong to `carry`, the segment
// open at c0 (0 at c0 == 0; updated for c0 + 128).  Marks are tagged with the call's epoch (seg_epoch: one counter
// per wave, owned by the kernel) and never cleared: a stale mark is smaller than any mark of this call and loses
// the maximum.
__device__ __forceinline__ void locate128(const WaveLds &lds, u32 &seg_epoch, const Segs &sg, u32 c0, u32 total, u32 &carry, bool &va,
                                          bool &vb, u32 &seg_a, u32 &seg_b, u32 &ea_at, u32 &eb_at) {
  const int lane = lane_id();
  const u32 epoch = ++seg_epoch, tag = epoch << 8;
  const u32 da = sg.start_a - c0, db = sg.start_b() - c0;
  if (sg.na && da < 128u) lds.smark[da] = tag | static_cast<u32>(2 * lane + 1);
  if (sg.nb && db < 128u) lds.smark[db] = tag | static_cast<u32>(2 * lane + 2);
  wave_sync();
  const bool two = c0 + 64 < total;  // (uniform)
  u32 ma = lds.smark[lane], mb = two ? lds.smark[64 + lane] : 0u;
  ma = wave_incl_max(ma);
  seg_a = (ma >> 8) == epoch ? (ma & 255u) - 1u : carry;
  carry = rdlane(seg_a, 63);
  va = c0 + lane < total;
  ea_at = c0 + lane + lds.sdelta[seg_a];
  vb = false; seg_b = 0; eb_at = 0;
  if (two) {
    mb = wave_incl_max(mb);
    seg_b = (mb >> 8) == epoch ? (mb & 255u) - 1u : carry;
    carry = rdlane(seg_b, 63);
    vb = c0 + 64 + lane < total;
    eb_at = c0 + 64 + lane + lds.sdelta[seg_b];
  }
}

// Ghost bits.  The reference hashes a read's first max(20, L/2) seed offsets and extends the last
// ones while their buckets are too big; for reads of 44-46 bases that reaches PAST the end of the read,
// into whatever the reused per-thread buffer of that encoding still holds: position k carries the
// nibble the nearest earlier read longer than k left there (prep_read only resizes;
// src/abismal.cpp:1163-1194, :1302-1308, :1377-1386; zero where nothing was written).
// This rebuilds those 2-letter bits for positions L .. L+63 of the four encodings of read r from the
// reads handed over before it (lens / packed of the same batch, input order = the reference at -t 1)
// and writes them into the bit strings qbits (>= 2 words per encoding).  Lane j <-> position L + j.
__device__ __forceinline__ void ghost_bits(const u64 *__restrict__ packed, const u32 *__restrict__ lens, u64 r, u32 L,
                                           u32 max_len, u32 min_len, u32 W, u32 WB, u64 *qbits) {
  const int lane = lane_id();
  const u32 k = L + static_cast<u32>(lane);
  bool found = k >= max_len;  // no read of this batch ever wrote that far: still the zero fill
  u64 src = 0;
  for (u64 base = r; base > 0 && __ballot(!found) != 0; base = base > 64 ? base - 64 : 0) {
    const u32 len_q = base > static_cast<u64>(lane) ? lens[base - 1 - lane] : 0u;
    for (int t = 0; t < 64; ++t) {
      const u32 lt = rdlane(len_q, t);
      if (!found && lt >= min_len && lt > k) { found = true; src = base - 1 - t; }
    }
  }
  const bool have = found && k < max_len && src < r && lens[src] > k;
#pragma unroll
  for (u32 e = 0; e < 4; ++e) {
    u32 nib = 0;
    if (have) nib = static_cast<u32>(packed[(src * 4 + e) * W + (k >> 4)] >> ((k & 15u) << 2)) & 15u;
    const u64 m = __ballot(bit2(nib) != 0);  // bit j = 2-letter bit at position L + j (1 for the zero fill)
    if (lane == 0) {
      const u64 real = qbits[e * WB] & ((1ull << L) - 1);
      qbits[e * WB] = real | (m << L);
      qbits[e * WB + 1] = (m >> (64 - L)) | (~0ull << L);
    }
  }
  wave_sync();
}

// One (strand, alphabet) call of process_seeds (src/abismal.cpp:1269-1375) for
// the whole wave.  Lanes are seed offsets while probing/narrowing, then become
// candidates (all checked buckets of 64 offsets flattened in reference order)
// for the Hamming filter; survivors are replayed in order into the set.
#ifndef ABM_SE_DIRECT_NARROWING
// (the single-end kernel narrows big ranges directly as well -- since the window records: until then it was neutral to
// slower there at either register budget; with the records the probes had become the largest phase of a
// 150-base read.  10 M x 100 bp 461-464 -> 438-440 ms, 4 M x 150 bp random PBAT 7.4-7.5 -> 8.3-8.5 M reads/s;
// profiles/r05_exp_se_direct_narrowing.log)
#define ABM_SE_DIRECT_NARROWING true
#endif
#ifndef ABM_SE_RECORD_PROBES
#define ABM_SE_RECORD_PROBES false
#endif
#ifndef ABM_HEAVY_BLOCK
#define ABM_HEAVY_BLOCK 4096
#endif
constexpr u32 kHeavyBlock = ABM_HEAVY_BLOCK;  // candidates in one block of 64 seed offsets from which a read counts as heavy
struct WorkTally {
  u32 seed_iters, probes, cands, words, updates, cache_hits;
  u32 fifo_updates, steps, light_steps;  // diagnostic build only
  // diagnostic build only (TIMED): shader cycles per phase, from s_memtime
  long long t_probe, t_stream, t_replay, t_align, t_total;
};
// a stamp drains outstanding memory traffic first so that waits are charged to the
// phase that issued them (the scheduler may otherwise hoist s_memtime above the wait)
__device__ __forceinline__ long long phase_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
  return static_cast<long long>(t);
}
#define ABM_STAMP(var) do { if (TIMED) var = phase_stamp(); } while (0)

// REC: the launch filters on the window records (DevIndex::wrec; every read of the launch is short enough for them).  A
// candidate's window is then addressed by its entry NUMBER, so the step neither waits for the index entries nor reads them
// at all, except for the few candidates whose distance is within the cutoff: those alone need a position -- for the
// set, and to ask whether an N is within reach of the window (the planes' distance never exceeds the true one -- a blank
// nibble matches nothing, the planes' code 0 there may match -- so a candidate beyond the cutoff on the planes is beyond
// it on the nibble array too).  No position cache either: a repeated window costs a third of a line, not a line.
template <bool SPECIFIC, bool TIMED, bool COOP, bool REC, class Set>
__device__ __forceinline__ void seed_pass(const DevIndex &ix, const WaveLds &lds, u32 enc, bool g_to_a,
                                          u32 flags, u32 L, Set &S, WorkTally &wt, u32 &seg_epoch) {
  const int lane = lane_id();
  const u64 *qpk = lds.qpk + enc * lds.W;
  const u64 *qb = lds.qbits + enc * lds.WB;
  const u32 *cnt3 = g_to_a ? ix.counter_a : ix.counter_t;
  const u32 *idx3 = g_to_a ? ix.index_a : ix.index_t;
  const u32 maxc = ix.max_candidates;
  const u32 nwords = (L + 15) >> 4;
  const u32 spec_len = min(L - ix.window, L >> 1);
  const u32 n_off = SPECIFIC ? max(ix.window, L >> 1) : L - kKeyWeight + 1;
  // the tables answer for offsets whose seed can be extended by all their letters inside the read (L - i >= depth for
  // every offset of the pass) and for the max_candidates they were built with; otherwise the loops start at the counters
  const bool use_ext = SPECIFIC && ix.ext2 != nullptr && ix.ext_maxc == maxc &&
                       L - n_off + 1 >= max(kKeyWeight + ix.e2, kKeyWeight3 + ix.e3);
  static_assert(!REC || COOP, "window records are filtered cooperatively");
  const u32 rec3 = g_to_a ? ix.wrec_a0 : ix.wrec_t0;  // (REC) record number of the 3-letter table's first entry

  // work tallies: the diagnostic builds keep them, the production kernels do not (they cost the single-end kernel
  // registers: 100 -> 64 bytes per lane of scratch without them; the pair kernels kept them until round 5, when five
  // lane-resident counters were what the seed kernel spilled in its offset loop)
  constexpr bool TALLY = TIMED;
  // direct narrowing of big ranges: in the pair kernels only (see narrow_direct)
  constexpr bool kDirect = Set::kAppend ? ABM_PE_DIRECT_NARROWING : ABM_SE_DIRECT_NARROWING;
  long long ta = 0, tb_ = 0, tc = 0, td = 0;
  if (SPECIFIC && !REC) {  // a new (strand, alphabet) call: the cache belongs to one encoding
    for (u32 k = lane; k < (1u << kPosCacheBits); k += 64) lds.pcache[k] = 0;
    wave_sync();
  }
  for (u32 g0 = 0; g0 < n_off && !S.sure_ambig; g0 += 64) {
      ABM_STAMP(ta);
    const u32 i = g0 + lane;
    const bool live = i < n_off;
    u32 lo2 = 0, hi2 = 0, lo3 = 0, hi3 = 0;
    bool chk2 = false, chk3 = false;
    if (live) {
      // 25-bit 2-letter key, MSB first (get_1bit_hash, src/AbismalIndex.hpp:285-294)
      const u32 wq = i >> 6, sq = i & 63u;
      u64 bits = qb[wq] >> sq;
      if (sq) bits |= qb[wq + 1] << (64 - sq);
      const u32 k2 = __brev(static_cast<u32>(bits) & 0x1FFFFFFu) >> 7;
      // 16-digit base-3 key (get_base_3_hash, src/AbismalIndex.hpp:296-305)
      const u64 win = q_window16(qpk, lds.W, i, L);
      u32 k3 = 0;
#pragma unroll
      for (u32 j = 0; j < 16; ++j)
        k3 = k3 * 3u + trit(static_cast<u32>(win >> (j << 2)) & 15u, g_to_a);
      // (keeping the specific pass's counters in LDS for the sensitive pass, which looks the first max(window, L/2)
      // offsets up again, took 4.4 % of the line requests away and made the kernel 1.6 % slower: measured, removed)
      // (likewise keeping the specific pass's distances, one byte per candidate, so that the sensitive pass fetches no
      // window for a bucket entry it has seen: 290 of 2,850 candidates per read found there, -7 % lines, and 2.3 KB more LDS
      // per wave cost 15 % of the time -- profiles/r02_exp_distance_memo.log)
      if (SPECIFIC) {
        u32 probes = 0;
        u32 len2 = kKeyWeight, len3 = kKeyWeight3;
        bool run2 = true, run3 = true;
        if (use_ext) {
          // seed-extension tables (abm_ext.hip): one independent 8-byte load per table says where the narrowing
          // loops stand after the next e2 / e3 letters -- for most offsets, finished
          const u32 D2 = kKeyWeight + ix.e2;
          const u32 K2 = __brev(static_cast<u32>(bits) & (D2 >= 32 ? 0xFFFFFFFFu : (1u << D2) - 1u)) >> (32 - D2);
          const u64 more = q_window16(qpk, lds.W, i + 16, L);
          u32 K3 = k3;
          for (u32 j = 0; j < ix.e3; ++j) K3 = K3 * 3u + trit(static_cast<u32>(more >> (j << 2)) & 15u, g_to_a);
          const uint2 x2 = ix.ext2[K2], x3 = (g_to_a ? ix.ext3a : ix.ext3t)[K3];
          const u32 st2 = x2.y >> 30, st3 = x3.y >> 30;
          lo2 = x2.x; hi2 = x2.x + (x2.y & 0x7FFFFFFu); len2 = kKeyWeight + ((x2.y >> 27) & 7u); run2 = st2 != 0;
          lo3 = x3.x; hi3 = x3.x + (x3.y & 0x7FFFFFFu); len3 = kKeyWeight3 + ((x3.y >> 27) & 7u); run3 = st3 != 0;
          if (st2 == 2) { lo2 = ix.counter[k2]; hi2 = ix.counter[k2 + 1]; len2 = kKeyWeight; }  // (not tabulated)
          if (st3 == 2) { lo3 = cnt3[k3]; hi3 = cnt3[k3 + 1]; len3 = kKeyWeight3; }
        }
        else {
          lo2 = ix.counter[k2]; hi2 = ix.counter[k2 + 1];
          lo3 = cnt3[k3];       hi3 = cnt3[k3 + 1];
        }
        if constexpr (kDirect) {
          if (ix.direct_min != 0 && L <= kSortDepth && ix.planes[0] != nullptr) {
            // big ranges -- seeds inside high-copy repeats -- are narrowed directly (narrow_direct); small ones, and
            // lanes whose probes come within reach of an N, by the letter-by-letter loop below
            const u32 limit = L - i;
            const DirectArgs da = {ix.planes[0], ix.nmap, qb, qpk, 64u * lds.WB, lds.W, L, maxc};
#pragma unroll 1
            for (int chain = 0; chain < 2; ++chain) {  // (one copy of the search in the code: 2-letter table, then the 3-letter one)
              const bool three = chain != 0;
              const u32 clo = three ? lo3 : lo2, chi = three ? hi3 : hi2, clen = three ? len3 : len2;
              // (clen < limit: the last offsets of a 44-46-base read start BEYOND the end of the read -- limit = L - i is below the
              // key weight -- and the reference extends those through whatever its reused buffer holds: the letter loop's
              // business, with the ghost bits)
              if ((three ? run3 : run2) && chi - clo >= ix.direct_min && chi - clo > maxc && clen < limit) {
                const DirectRange r = narrow_direct(three ? (g_to_a ? 2 : 1) : 0, da, three ? idx3 : ix.index, i, limit, clo, chi, clen);
                probes += r.probes;
                if (r.ok) {
                  if (three) { lo3 = r.lo; hi3 = r.hi; len3 = r.len; run3 = false; }
                  else { lo2 = r.lo; hi2 = r.hi; len2 = r.len; run2 = false; }
                }
              }
            }
          }
        }
        // (letters from the window records in the pair kernels only: the single-end kernel, short of registers, is 6 %
        // slower with them -- 527 -> 558 ms per 10 M reads -- where the pair kernels' seed passes gain 1.5 %;
        // profiles/r05_exp_record_probes.log)
        narrow_both<REC && (Set::kAppend || ABM_SE_RECORD_PROBES)>(ix, idx3, g_to_a, qb, 64u * lds.WB, qpk, i, L - i, maxc, run2, lo2, hi2, len2, run3, lo3, hi3, len3, probes, rec3);
        chk2 = (hi2 - lo2) <= maxc || len2 >= spec_len;
        chk3 = (hi3 - lo3) <= maxc || len3 >= spec_len;
        if (TALLY) wt.probes += probes;
      }
      else {
        lo2 = ix.counter[k2]; hi2 = ix.counter[k2 + 1];
        lo3 = cnt3[k3];       hi3 = cnt3[k3 + 1];
        const u32 d2 = hi2 - lo2, d3 = hi3 - lo3;
        chk2 = d2 != 0 && d2 <= maxc && (d3 == 0 || d2 <= 10u * d3);
        chk3 = d3 != 0 && d3 <= maxc;
      }
      if (TALLY) ++wt.seed_iters;
    }
    Segs sg;
    sg.na = chk2 ? hi2 - lo2 : 0u;
    sg.nb = chk3 ? hi3 - lo3 : 0u;
    sg.lo2 = lo2;
    sg.lo3 = lo3;
    if constexpr (TALLY && Set::kAppend) {
      // pair kernels: the 128-byte lines the checked buckets' index-entry runs span (32 entries a line) -- their share of a
      // pair's line requests (DESIGN 4.3's table); the words of the fetched windows follow from the candidate count
      if (sg.na) wt.words += ((lo2 + sg.na - 1) >> 5) - (lo2 >> 5) + 1;
      if (sg.nb) wt.words += ((lo3 + sg.nb - 1) >> 5) - (lo3 >> 5) + 1;
    }
    u32 total;
    sg.start_a = wave_excl_sum(sg.na + sg.nb, total);
    ABM_STAMP(tb_);
    if (TIMED) wt.t_probe += tb_ - ta;
    if (total == 0) continue;
    // A read with this many candidates in one block of offsets is one of the launch's costliest (a homopolymer, a
    // satellite): its wave is served first from here on (until the kernel's next read), so that it runs at the pace of
    // a wave alone on its SIMD instead of a fifth of it -- such reads are what a launch's last milliseconds wait for.
    if (total >= kHeavyBlock) __builtin_amdgcn_s_setprio(3);
    if (seg_epoch >= kSegEpochLimit) {  // (sixteen million steps on: start the mark tags over)
      lds.smark[lane] = 0; lds.smark[64 + lane] = 0;
      seg_epoch = 0;
    }
    publish_segs(lds, sg);

    // ordered replay of one 64-candidate sub-chunk (check_hits + update, :1133-1149, :394-404)
    auto replay = [&](bool valid, int h, int hmax, u32 pos) {
      u64 todo = __ballot(valid && hmax <= S.cutoff);
      if constexpr (Set::kAppend) if (SPECIFIC && !S.heaped && todo) {
        // growing paired-end set: every survivor goes in, the whole chunk at once (see PeSet)
        const int offered = __popcll(todo);
        todo = S.append(todo, h, pos);
        if (TALLY) wt.updates += static_cast<u32>(offered - __popcll(todo));
      }
      while (todo && !S.sure_ambig) {
        if constexpr (Set::kFifo) {
          // full set, survivors that tie with the cutoff: applied a run at a time (SeSet::tie_run)
          const int applied = S.tie_run(todo, __ballot(valid && h == S.cutoff), pos, flags);
          if (applied) {
            if (TALLY) wt.updates += static_cast<u32>(applied);
            if (TIMED) wt.fifo_updates += static_cast<u32>(applied);
            continue;
          }
        }
        const int l = __builtin_ctzll(todo);
        const int before = S.cutoff;
        S.admit(true, rdlane(h, l), flags, rdlane(pos, l));
        if (TALLY) ++wt.updates;
        todo &= ~(((1ull << l) << 1) - 1);
        if (S.cutoff < before) todo &= __ballot(valid && hmax <= S.cutoff);
      }
    };

    // Candidates are taken 128 at a time, two per lane (a: c0+lane, b: c0+64+lane): both index entries,
    // then both genome windows, are in flight together, which halves the dependent round trips of a
    // read with very many candidates.  The set still sees them strictly in the reference's order.
    // The index entries of the NEXT 128 candidates are requested before this step's windows, so a
    // step costs one dependent memory round trip instead of two (what a read with millions of
    // candidates -- one wave, no other latency hiding at the end of a launch -- is made of).
    // A step with at most 64 candidates (most steps of an ordinary read) does nothing for its b half.
    if constexpr (REC) {
      u32 carry = 0;
      for (u32 c0 = 0; c0 < total && !S.sure_ambig; c0 += 128) {
        ABM_STAMP(tc);
        if (TIMED) { ++wt.steps; if (total - c0 <= 64) ++wt.light_steps; }
        const bool two = c0 + 64 < total;  // (uniform: this step has a b half)
        bool va, vb;
        u32 sa, sb, ea, eb;
        locate128(lds, seg_epoch, sg, c0, total, carry, va, vb, sa, sb, ea, eb);
        const u32 ia = g0 + (sa >> 1), ib = g0 + (sb >> 1);
        // record start (in 16-byte blocks) and the bit of the record the window begins at
        const u32 ra = (ea + ((sa & 1u) ? rec3 : 0u)) * ix.wrec_blocks, rb = (eb + ((sb & 1u) ? rec3 : 0u)) * ix.wrec_blocks;
        int ha, hb = 0x7fff;
        if (lds.G == 2)
          hamming_records_lane<false>(ix, lds.qmask + enc * lds.MB * 4, L, ra, ix.wrec_back - ia, va, rb, ix.wrec_back - ib, vb, two, ha, hb);
        else
          hamming_records_lane<true>(ix, lds.qmask + enc * lds.MB * 4, L, ra, ix.wrec_back - ia, va, rb, ix.wrec_back - ib, vb, two, ha, hb);
        // the candidates that may still enter the set: these alone need their position
        const bool ca = va && ha <= S.cutoff, cb = two && vb && hb <= S.cutoff;
        u32 pa = 0, pb = 0;
        if (__any(ca || cb)) {
          if (ca) pa = ((sa & 1u) ? idx3[ea] : ix.index[ea]) - ia;
          if (cb) pb = ((sb & 1u) ? idx3[eb] : ix.index[eb]) - ib;
          const u32 na = ca ? ix.nmap[pa >> (kPlaneChunkBits + 5)] >> ((pa >> kPlaneChunkBits) & 31u) : 0u;
          const u32 nb = cb ? ix.nmap[pb >> (kPlaneChunkBits + 5)] >> ((pb >> kPlaneChunkBits) & 31u) : 0u;
          if (__any((na | nb) & 1u)) {  // rare: redone on the nibble array, where an N is an N
            if (na & 1u) ha = hamming(ix.genome, qpk, nwords, pa);
            if (nb & 1u) hb = hamming(ix.genome, qpk, nwords, pb);
          }
        }
        if (TALLY) {
          wt.cands += (va ? 1u : 0u) + (vb ? 1u : 0u);
          if constexpr (!Set::kAppend) wt.words += ((va ? 1u : 0u) + (vb ? 1u : 0u)) * (lds.G == 2 ? 6u : 8u);
        }
        ABM_STAMP(td);
        if (TIMED) wt.t_stream += td - tc;
        replay(ca, ha, ha, pa);
        if (two && !S.sure_ambig) replay(cb, hb, hb, pb);
        ABM_STAMP(tc);
        if (TIMED) wt.t_replay += tc - td;
      }
      continue;
    }
    u32 carry = 0;
    bool nva = false, nvb = false;
    u32 nsa = 0, nsb = 0, nea = 0, neb = 0;
    auto fetch_entries = [&](u32 c0) {
      u32 ea_at, eb_at;
      locate128(lds, seg_epoch, sg, c0, total, carry, nva, nvb, nsa, nsb, ea_at, eb_at);
      nea = 0; neb = 0;
      if (nva) nea = (nsa & 1u) ? idx3[ea_at] : ix.index[ea_at];
      if (nvb) neb = (nsb & 1u) ? idx3[eb_at] : ix.index[eb_at];
    };
    fetch_entries(0);
    for (u32 c0 = 0; c0 < total && !S.sure_ambig; c0 += 128) {
          ABM_STAMP(tc);
      if (TIMED) { ++wt.steps; if (total - c0 <= 64) ++wt.light_steps; }
      const bool two = c0 + 64 < total;  // (uniform: this step has a b half)
      const bool va = nva, vb = nvb;
      const u32 pa = nea - (g0 + (nsa >> 1)), pb = neb - (g0 + (nsb >> 1));
      if (c0 + 128 < total) fetch_entries(c0 + 128);
      // the same genome position is proposed again and again (neighbouring seeds of one hit, the
      // sensitive pass repeating the specific one): a small per-call cache of (pos -> distances)
      // saves the 1-2 HBM lines of a window.  Distances are a pure function of (pos, encoding),
      // so a cache hit is exact by construction.  (Without it: 7 % more windows fetched, 18 instead of 25 spilled
      // VGPRs, 10 M reads 1.1 % faster and 1 M reads 2.4 % slower -- profiles/r02_exp_position_cache.log; kept.)
      u64 *slot_a = lds.pcache + ((pa * 2654435761u) >> (32 - kPosCacheBits));
      u64 *slot_b = lds.pcache + ((pb * 2654435761u) >> (32 - kPosCacheBits));
      const u64 ca = va ? *slot_a : 0ull, cb = vb ? *slot_b : 0ull;
      const bool hit_a = va && static_cast<u32>(ca) == pa, hit_b = vb && static_cast<u32>(cb) == pb;
      int ha, hma, hb = 0x7fff, hmb = 0x7fff;
      if constexpr (COOP) {  // (distances are complete sums: no genome letter here makes a word's share negative)
        // (is an N within reach of the window?  asked before the windows so that the answers arrive with them)
        const u32 na = va && !hit_a ? ix.nmap[pa >> (kPlaneChunkBits + 5)] >> ((pa >> kPlaneChunkBits) & 31u) : 0u;
        const u32 nb = vb && !hit_b ? ix.nmap[pb >> (kPlaneChunkBits + 5)] >> ((pb >> kPlaneChunkBits) & 31u) : 0u;
        if (lds.G == 2)
          hamming_planes_pairs(ix, lds, lds.qmask + enc * lds.MB * 4, L, pa, va && !hit_a, pb, vb && !hit_b, ha, hb);
        else
          hamming_planes(ix, lds, lds.qmask + enc * lds.MB * 4, L, pa, va && !hit_a, pb, vb && !hit_b, ha, hb);
        if (__any((na | nb) & 1u)) {  // rare: redone on the nibble array, where an N is an N
          if (na & 1u) ha = hamming(ix.genome, qpk, nwords, pa);
          if (nb & 1u) hb = hamming(ix.genome, qpk, nwords, pb);
        }
        hma = ha; hmb = hb;
      }
      else
        hamming2(ix.genome, qpk, nwords, pa, va && !hit_a, pb, vb && !hit_b, ha, hma, hb, hmb);
      if (hit_a) { ha = static_cast<i16>(static_cast<u16>(ca >> 32)); hma = static_cast<i16>(static_cast<u16>(ca >> 48)); }
      if (va && !hit_a)
        *slot_a = static_cast<u64>(pa) | (static_cast<u64>(static_cast<u16>(ha)) << 32) | (static_cast<u64>(static_cast<u16>(hma)) << 48);
      if (!va) { ha = hma = 0x7fff; }
      if (two) {
        if (hit_b) { hb = static_cast<i16>(static_cast<u16>(cb >> 32)); hmb = static_cast<i16>(static_cast<u16>(cb >> 48)); }
        if (vb && !hit_b)
          *slot_b = static_cast<u64>(pb) | (static_cast<u64>(static_cast<u16>(hb)) << 32) | (static_cast<u64>(static_cast<u16>(hmb)) << 48);
        if (!vb) { hb = hmb = 0x7fff; }
      }
      if (TALLY) {
      wt.cands += (va ? 1u : 0u) + (vb ? 1u : 0u);
      // 8-byte words fetched per window: the read's words on the nibble array; on the bit planes a group of four
      // lanes fetches four 16-byte blocks, a group of eight the blocks its window has (at most L / 64 + 2)
      const u32 fetched = COOP ? (lds.G == 2 ? 6u : (lds.G == 4 ? 8u : 2u * ((L + kPlaneBlock - 1) / kPlaneBlock + 1))) : nwords;
      if constexpr (!Set::kAppend) wt.words += ((va ? 1u : 0u) + (vb ? 1u : 0u)) * fetched;
      wt.cache_hits += (hit_a ? 1u : 0u) + (hit_b ? 1u : 0u);
      }
      ABM_STAMP(td);
      if (TIMED) wt.t_stream += td - tc;
      replay(va, ha, hma, pa);
      if (two && !S.sure_ambig) replay(vb, hb, hmb, pb);
      ABM_STAMP(tc);
      if (TIMED) wt.t_replay += tc - td;
    }
  }
}

// =============================================================================
// Banded local alignment (AbismalAlign::align, src/AbismalAlign.hpp:320-386) as
// an anti-diagonal wavefront.  Cell (i,j) of the reference's band table (row i =
// target base t_beg+i-1, column j, read index q = i+j-bw) depends on (i-1,j)
// [substitution], (i-1,j+1) [from_above] and (i,j-1) [from_left]; on the
// anti-diagonal t = 2i+j all three are already known, from the same lane two
// steps ago and from the two neighbouring lanes one step ago.  So lane = band
// column, one DPP move per neighbour per step, and no in-row scan.  A band is
// at most 61 lanes wide and usually ~21 (2*min(diffs,max_diffs)+1), so several
// candidate alignments ("jobs") share one wave side by side.
// Cell validity (left/right of the reference's row loop) reduces to 0 <= q < L;
// invalid cells read as 0, exactly like the zero-filled table.
// =============================================================================
__device__ __forceinline__ int band_for(int diffs, int max_diffs) {
  const int v = 2 * min(diffs, max_diffs) + 1;
  return v < 0 ? static_cast<int>(kMaxBand) : min(static_cast<int>(kMaxBand), v);
}
// (bound_ctrl: the lane without a source reads 0 and no lane keeps its old value, so the destination needs no initialising move)
__device__ __forceinline__ int from_prev_lane(int v) {  // lane j <- lane j-1 (lane 0 <- 0)
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ int from_next_lane(int v) {  // lane j <- lane j+1 (lane 63 <- 0)
  return __builtin_amdgcn_update_dpp(0, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true);
}

// 16 nibbles starting at nibble index `start` of an LDS word array; nibbles
// outside [0, 16*nwords) read as 0 (they only ever feed invalid cells)
__device__ __forceinline__ u64 nibbles16(const u64 *words, int nwords, int start) {
  if (start <= -16 || start >= 16 * nwords)
    return 0ull;
  if (start < 0)
    return words[0] << ((-start) << 2);
  const int w = start >> 4, s = (start & 15) << 2;
  u64 x = words[w] >> s;
  if (s && w + 1 < nwords) x |= words[w + 1] << (64 - s);
  return x;
}

struct AlnJob {  // per lane: the job whose band column this lane is
  int bw;        // 0 = lane unassigned
  int jl;        // column within the band
  int qoff;      // word offset of the query encoding in lds.qpk
  int g;         // genome-window slot in lds.gwin
  int t0nib;     // t_beg & 15: nibble offset of row 1's target base inside the window
};

// Runs every job assigned in `job` to completion.  Returns, per lane, the best
// cell value of its own column (and its first row); with TB also stores one
// byte per cell: arrow (0 M, 1 I, 2 D, 3 none) | 4 if the cell's score is > 0.
// WRAP: every candidate score is narrowed to 16 bits before it competes, as the reference's score_t arithmetic does
// (src/AbismalAlign.hpp:35, :239-262: `const score_t score = ... + *cur_row`); it only ever matters for reads beyond
// 16383 bases, whose perfect score 2 L no longer fits -- the long-read kernel's instantiations.
template <bool WRAP> __device__ __forceinline__ int score16(int v) { return WRAP ? static_cast<int>(static_cast<i16>(v)) : v; }

template <bool TB, bool WRAP = false>
__device__ __forceinline__ void wavefront(const WaveLds &lds, const AlnJob &job, int L, int bw_min,
                                          int bw_max, int &bestv, int &bestrow) {
  const int jl = job.jl, bw = job.bw;
  const bool assigned = bw != 0;
  const u64 *qw = lds.qpk + job.qoff;
  const u64 *gw = lds.gwin + job.g * lds.GW;
  const int t_start = max(0, bw_min - 1), t_end = 2 * (L - 1 + bw_max);
  int cur = 0;
  u64 M = 0;
  bestv = 0; bestrow = 0;
  for (int t = t_start; t <= t_end; ++t) {
    if (((t - t_start) & 31) == 0) {
      // match bits for this lane's next 16 cells: nibble k set <=> q[i0+k+jl-bw] & T[i0+k-1] != 0
      const int ta = t + ((t - jl) & 1);
      const int i0 = (ta - jl) >> 1;
      u64 x = 0;
      if (assigned)
        x = nibbles16(qw, static_cast<int>(lds.W), i0 + jl - bw) &
            nibbles16(gw, static_cast<int>(lds.GW), job.t0nib + i0 - 1);
      x |= x >> 1;
      x |= x >> 2;
      M = x & 0x1111111111111111ull;
    }
    const int dlt = t - jl;
    const bool active = assigned && (dlt & 1) == 0;
    const int i = dlt >> 1;
    const int q = i + jl - bw;
    const bool valid = active && q >= 0 && q < L;
    const int lf = from_prev_lane(cur), up = from_next_lane(cur);
    const int sdiag = score16<WRAP>(cur + ((static_cast<u32>(M) & 1u) ? 2 : -3));
    int c = max(sdiag, 0);
    int arrow = (c == sdiag) ? 0 : 3;
    if (jl < bw - 1 && q < L - 1) {   // from_above: j in [left, right-1)
      const int s = score16<WRAP>(up - 4);
      c = max(c, s);
      if (TB && c == s) arrow = 2;
    }
    if (jl > 0 && q > 0) {            // from_left: j in [left+1, right)
      const int s = score16<WRAP>(lf - 4);
      c = max(c, s);
      if (TB && c == s) arrow = 1;
    }
    if (active) {
      M >>= 4;
      cur = valid ? c : 0;
      if (valid && c > bestv) { bestv = c; bestrow = i; }
      if (TB && i >= 0 && i < L + bw)
        lds.tb[i * bw + jl] = static_cast<u8>(valid ? (arrow | (c > 0 ? 4 : 0)) : 3);
    }
  }
}

// The traceback run row by row: ONE job on lanes [0, bw), every lane computes its cell of row i in the same step
// (the anti-diagonal schedule keeps a lone job's lanes idle every other step).  The one dependency inside a row --
// from_left, c(i,j) = max(base(i,j), c(i,j-1) - 4) -- unrolls to c(i,j) = max over k <= j of base(i,k) - 4 (j - k), an
// inclusive running maximum of base + 4 k over the band's lanes: ceil(log2 bw) DPP steps (two for the band of three
// a read with one mismatch gets).  Cells left of the read (q < 0) compute to 0 like the reference's zero-filled
// table, cells right of it (q >= L) cannot reach a valid cell through from_left and are masked when published.
// Arrows as the reference decides them: left wins ties, then above, then the diagonal.  Same table, scores, first
// maximum per column as wavefront<TB>; 32-bit scores (the long-read kernel keeps wavefront<TB, true>).
#ifndef ABM_TB_ROWS
#define ABM_TB_ROWS true  // (false: the traceback run on the anti-diagonal schedule, for same-box comparisons)
#endif
constexpr bool kTracebackByRows = ABM_TB_ROWS;
template <bool TB>
__device__ __forceinline__ void wavefront_rows(const WaveLds &lds, const AlnJob &job, int L, int bw, int &bestv,
                                               int &bestrow) {
  const int jl = job.jl;
  const bool assigned = job.bw != 0;
  const u64 *qw = lds.qpk + job.qoff;
  const u64 *gw = lds.gwin + job.g * lds.GW;
  const int off4 = 4 * jl;
  const bool up_lane = assigned && jl < bw - 1;
  const int rows = L + bw;
  int cur = 0;
  u64 M = 0;
  bestv = 0; bestrow = 0;
  for (int i = 0; i < rows; ++i) {
    if ((i & 15) == 0) {  // match bits of this lane's next 16 rows
      u64 x = 0;
      if (assigned)
        x = nibbles16(qw, static_cast<int>(lds.W), i + jl - bw) & nibbles16(gw, static_cast<int>(lds.GW), job.t0nib + i - 1);
      x |= x >> 1;
      x |= x >> 2;
      M = x & 0x1111111111111111ull;
    }
    const int q = i + jl - bw;
    const bool valid = assigned && q >= 0 && q < L;
    const int up = from_next_lane(cur) - 4;
    const int sdiag = cur + ((static_cast<u32>(M) & 1u) ? 2 : -3);
    M >>= 4;
    int base = max(sdiag, 0);
    int arrow = (base == sdiag) ? 0 : 3;
    if (up_lane && q < L - 1) {  // from_above
      base = max(base, up);
      if (TB && base == up) arrow = 2;
    }
    u32 x = static_cast<u32>(base + off4);  // (>= 0: the scans' identity 0 never wins)
    if (bw > 1) x = max(x, dpp_from<0x111, 0xf>(0u, x));
    if (bw > 2) x = max(x, dpp_from<0x112, 0xf>(0u, x));
    if (bw > 4) x = max(x, dpp_from<0x114, 0xf>(0u, x));
    if (bw > 8) x = max(x, dpp_from<0x118, 0xf>(0u, x));
    if (bw > 16) x = max(x, dpp_from<0x142, 0xa>(0u, x));
    if (bw > 32) x = max(x, dpp_from<0x143, 0xc>(0u, x));
    const int c = static_cast<int>(x) - off4;
    if (TB && c == from_prev_lane(c) - 4) arrow = 1;  // from_left (an invalid or missing neighbour offers -4)
    cur = valid ? c : 0;
    if (valid && c > bestv) { bestv = c; bestrow = i; }
    if (TB && assigned) lds.tb[i * bw + jl] = static_cast<u8>(valid ? (arrow | (c > 0 ? 4 : 0)) : 3);
  }
}

// where CIGARs go: fixed slots of `stride` ops per read; one with more ops goes whole into the launch's
// overflow arena (bump-allocated), its slot's first word = where, and its count (> stride) says so
struct CigarSink {
  u32 stride;
  u32 ctmp_cap;        // entries of the reversed-ops scratch in LDS (longest read + 2: no CIGAR has more ops)
  u32 *arena;          // [arena_cap] or null
  u32 *arena_count;    // ops handed out so far
  u32 arena_cap;
  u32 *fin;            // optional: the CIGAR's first kSeCap ops in LDS as well (the single-end kernel's SAM text is made from them)
};

// build_cigar_len_and_pos + get_traceback (src/AbismalAlign.hpp:166-193, :388-440).
// Runs uniformly on the wave; ops are collected reversed in LDS then emitted.
__device__ __forceinline__ void wave_cigar(const u8 *tb, u32 *ctmp, int L, int diffs, int max_diffs,
                                           int score, int best_r, int best_c, u32 *cig_out,
                                           const CigarSink &sink, u32 &n_ops, int &ins, int &del, u32 &aln_len,
                                           u32 &t_pos, bool &overflow) {
  const int lane = lane_id();
  ins = del = 0;  // count_total_ops<I>/<D> with oplen() narrowed to uint8_t (abismal_cigar_utils.hpp:50-53)
  if (score == 0 || diffs == 0) {
    if (lane == 0) { store_out(cig_out, static_cast<u32>(L) << 4); if (sink.fin) sink.fin[0] = static_cast<u32>(L) << 4; }
    n_ops = 1;
    aln_len = static_cast<u32>(L);
    return;
  }
  const int bw = band_for(diffs, max_diffs);
  int r = best_r, c = best_c;
  const int clip_tail = (L + (bw - 1)) - (r + c);
  u32 n = 0;
  auto emit = [&](u32 run, int op) {
    if (n < sink.ctmp_cap) { if (lane == 0) ctmp[n] = (run << 4) | static_cast<u32>(op); }
    const int r8 = static_cast<int>(static_cast<u8>(run));
    ins = op == 1 ? static_cast<i16>(ins + r8) : ins;  // (value selects: conditional stores through the two references go to scratch)
    del = op == 2 ? static_cast<i16>(del + r8) : del;
    ++n;
  };
  auto step = [&](int a) {
    if (a != 1) --r;
    if (a == 1) --c;
    if (a == 2) ++c;
  };
  int op = uni(tb[r * bw + c]) & 3;
  step(op);
  u32 run = 1;
  for (;;) {
    const int cell = uni(tb[r * bw + c]);
    if (!(cell & 4)) break;
    const int a = cell & 3;
    step(a);
    if (a != op) { emit(run, op); run = 0; }
    ++run;
    op = a;
  }
  emit(run, op);
  const int clip_head = (r + c) - (bw - 1);
  wave_sync();
  // final order: [head clip] reversed(ops) [tail clip]
  const u32 full = n + (clip_head > 0) + (clip_tail > 0);  // the CIGAR's op count
  u32 *dst = cig_out;
  u32 body = n, total = full;
  if (full > sink.stride) {  // too long for the slot: the whole CIGAR goes to the arena
    u32 at = 0xFFFFFFFFu;
    if (sink.arena != nullptr && n <= sink.ctmp_cap) {
      if (lane == 0) at = atomicAdd(sink.arena_count, full);
      at = static_cast<u32>(uni(static_cast<int>(at)));
      if (at > sink.arena_cap || full > sink.arena_cap - at) at = 0xFFFFFFFFu;
    }
    if (at != 0xFFFFFFFFu) {
      dst = sink.arena + at;
      if (lane == 0) store_out(cig_out, at);
    }
    else {  // no room: the slot gets what fits and the launch is flagged
      overflow = true;
      body = min(min(n, sink.ctmp_cap), sink.stride);
      total = min(body + (clip_head > 0) + (clip_tail > 0), sink.stride);
    }
  }
  for (u32 k = lane; k < total; k += 64) {
    u32 v;
    const u32 kk = k - (clip_head > 0 ? 1u : 0u);
    if (clip_head > 0 && k == 0) v = (static_cast<u32>(clip_head) << 4) | 4u;
    else if (kk < body) v = ctmp[body - 1 - kk];
    else v = (static_cast<u32>(clip_tail) << 4) | 4u;
    store_out(dst + k, v);
    if (sink.fin && k < kSeCap) sink.fin[k] = v;
  }
  n_ops = full;  // > stride: the ops are in the arena at slot[0] (or, with ABM_STATUS_CIGAR_OVERFLOW set, nowhere complete)
  aln_len = static_cast<u32>(L - clip_tail - clip_head);
  t_pos = t_pos - static_cast<u32>((bw - 1) / 2) + static_cast<u32>(r);
}

// simple_aln::edit_distance with the reference's integer types
// (src/AbismalAlign.hpp:73-89); ins/del are the op totals gathered in wave_cigar
__device__ __forceinline__ int edit_distance(int scr, u32 len, int ins, int del) {
  if (scr == 0)
    return static_cast<i16>(len);
  const int A = static_cast<i16>(scr + 4 * (ins + del));
  const u32 num = 2u * (len - static_cast<u32>(ins)) - static_cast<u32>(A);
  const int mism = static_cast<i16>(num / 5u);
  return static_cast<i16>(mism + ins + del);
}

__device__ __forceinline__ bool long_enough(u32 aln_len, u32 readlen, u32 kMinReadLen) {
  const double min_frac = 1.0 - 0.4;  // src/abismal.cpp:307-314
  return aln_len >= max(kMinReadLen, static_cast<u32>(min_frac * readlen));
}

// encoding index of the query a hit was found with (src/abismal.cpp:1463-1464)
__device__ __forceinline__ u32 enc_of(u32 flags) {
  const u32 rc = (flags & kFlagRC) ? 1u : 0u, ar = (flags & kFlagARich) ? 1u : 0u;
  return rc * 2u + (rc ^ ar);
}

// Stage the genome windows of jobs [first, first+n) of the LDS job list into
// lds.gwin (one coalesced sweep, loads issued before the stores)
__device__ __forceinline__ void stage_windows(const DevIndex &ix, const WaveLds &lds, int first, int n,
                                              int md) {
  const int lane = lane_id();
  const int GW = static_cast<int>(lds.GW), total = n * GW;
  for (int k0 = 0; k0 < total; k0 += 256) {
    u64 v[4];
    int at[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + lane + 64 * r;
      at[r] = -1;
      v[r] = 0;
      if (k < total) {
        const int g = k / GW, w = k - g * GW;
        const u32 pos = lds.jpos[first + g];
        const int bw = band_for(static_cast<int>(lds.jdf[first + g]) >> 16, md);
        const u64 t_beg = static_cast<u64>(pos) - static_cast<u64>((bw - 1) / 2);
        v[r] = ix.genome[(t_beg >> 4) + w];
        at[r] = k;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (at[r] >= 0) lds.gwin[at[r]] = v[r];
  }
}

// ---- scoring runs (align<false>): two job sets per wave ---------------------------
// On the anti-diagonal schedule a lane computes a cell only every other step.  A second
// set of jobs, laid out over the same lanes and started one step later, fills the idle
// steps: whatever a lane's neighbours published in the previous step then always belongs
// to the set the lane is working on now, so both sets share the two DPP moves and no
// select is needed -- each lane just alternates between its two jobs (`jp` on even steps
// counted from t_start, `jq` on odd ones; dp/dq = 1 if that job belongs to the delayed set).
template <bool WRAP = false>
__device__ __forceinline__ void wavefront_pair(const WaveLds &lds, const AlnJob &jp, const AlnJob &jq, int dp,
                                               int dq, int L, int bw_min, int bw_max, int &bestp, int &bestq) {
  const int t_start = max(0, bw_min - 1), t_end = 2 * (L - 1 + bw_max) + 1;
  const u64 *qp = lds.qpk + jp.qoff, *gp = lds.gwin + jp.g * lds.GW;
  const u64 *qq = lds.qpk + jq.qoff, *gq = lds.gwin + jq.g * lds.GW;
  int curp = 0, curq = 0, pub = 0;
  u64 Mp = 0, Mq = 0;
  bestp = bestq = 0;
  auto refill = [&](const AlnJob &j, const u64 *qw, const u64 *gw, int time) -> u64 {
    const int i0 = (time - j.jl) >> 1;
    u64 x = 0;
    if (j.bw)
      x = nibbles16(qw, static_cast<int>(lds.W), i0 + j.jl - j.bw) &
          nibbles16(gw, static_cast<int>(lds.GW), j.t0nib + i0 - 1);
    x |= x >> 1;
    x |= x >> 2;
    return x & 0x1111111111111111ull;
  };
  auto cell = [&](const AlnJob &j, int time, int &cur, u64 &M, int &best) {
    const int i = (time - j.jl) >> 1;
    const int q = i + j.jl - j.bw;
    const bool valid = j.bw != 0 && q >= 0 && q < L;
    const int lf = from_prev_lane(pub), up = from_next_lane(pub);
    int c = max(score16<WRAP>(cur + ((static_cast<u32>(M) & 1u) ? 2 : -3)), 0);
    if (j.jl < j.bw - 1 && q < L - 1) c = max(c, score16<WRAP>(up - 4));  // from_above
    if (j.jl > 0 && q > 0) c = max(c, score16<WRAP>(lf - 4));             // from_left
    M >>= 4;
    cur = valid ? c : 0;
    best = max(best, cur);
    pub = cur;
  };
  for (int t = t_start; t <= t_end; t += 2) {
    if (((t - t_start) & 31) == 0) {
      Mp = refill(jp, qp, gp, t - dp);
      Mq = refill(jq, qq, gq, t + 1 - dq);
    }
    cell(jp, t - dp, curp, Mp, bestp);
    cell(jq, t + 1 - dq, curq, Mq, bestq);
  }
}

__device__ __forceinline__ int wave_max_i32(int x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x = max(x, __shfl_xor(x, d));
  return x;
}

// Scores jobs [first, ...) of the LDS job list (jpos / jdf = diffs<<16 | flags), as many as fit
// one wave: consecutive jobs share a slot of lanes (one in each set), slots sit side by side.
// Returns one past the last job taken; the score of job first+k is left in lds.lbest[k].
template <bool WRAP = false>
__device__ __forceinline__ int score_round(const DevIndex &ix, const WaveLds &lds, int first, int n_jobs,
                                           int L, int md, int qbase) {
  const int lane = lane_id();
  AlnJob ja = {0, 0, 0, 0, 0}, jb = {0, 0, 0, 0, 0};
  int used = 0, s = first, bw_min = 64, bw_max = 0, my_o = 0;
  while (s < n_jobs && s - first + 2 <= static_cast<int>(lds.max_jobs)) {
    const bool two = s + 1 < n_jobs;
    const u32 dfa = lds.jdf[s], dfb = two ? lds.jdf[s + 1] : 0u;
    const int bwa = band_for(static_cast<int>(dfa) >> 16, md);
    const int bwb = two ? band_for(static_cast<int>(dfb) >> 16, md) : 0;
    const int width = max(bwa, bwb);
    if (used + width > 64) break;
    const int o = lane - used;
    if (o >= 0 && o < width) my_o = o;
    if (o >= 0 && o < bwa) {
      const u64 t_beg = static_cast<u64>(lds.jpos[s]) - static_cast<u64>((bwa - 1) / 2);
      ja.bw = bwa; ja.qoff = qbase + static_cast<int>(enc_of(dfa & 0xFFFFu) * lds.W);
      ja.g = s - first; ja.t0nib = static_cast<int>(t_beg & 15u);
    }
    if (o >= 0 && o < bwb) {
      const u64 t_beg = static_cast<u64>(lds.jpos[s + 1]) - static_cast<u64>((bwb - 1) / 2);
      jb.bw = bwb; jb.qoff = qbase + static_cast<int>(enc_of(dfb & 0xFFFFu) * lds.W);
      jb.g = s + 1 - first; jb.t0nib = static_cast<int>(t_beg & 15u);
    }
    used += width;
    bw_min = min(bw_min, two ? min(bwa, bwb) : bwa);
    bw_max = max(bw_max, width);
    s += two ? 2 : 1;
  }
  stage_windows(ix, lds, first, s - first, md);
  wave_sync();
  const int t_start = max(0, bw_min - 1);
  const bool even = ((t_start - my_o) & 1) == 0;  // which of the lane's jobs is due on even steps
  ja.jl = jb.jl = my_o;
  int bp, bq;
  wavefront_pair<WRAP>(lds, even ? ja : jb, even ? jb : ja, even ? 0 : 1, even ? 1 : 0, L, bw_min, bw_max, bp, bq);
  lds.lbest[lane] = (even ? bp : bq) | ((even ? bq : bp) << 16);
  wave_sync();
  int base = 0, mine = 0;
  for (int k = first; k < s;) {
    const bool two = k + 1 < s;
    const int bwa = band_for(static_cast<int>(lds.jdf[k]) >> 16, md);
    const int bwb = two ? band_for(static_cast<int>(lds.jdf[k + 1]) >> 16, md) : 0;
    const int v = lane < max(bwa, bwb) ? lds.lbest[base + lane] : 0;
    const int sa = wave_max_i32(lane < bwa ? (v & 0xFFFF) : 0);
    if (lane == k - first) mine = sa;
    if (two) {
      const int sb = wave_max_i32(lane < bwb ? (v >> 16) : 0);
      if (lane == k + 1 - first) mine = sb;
    }
    base += max(bwa, bwb);
    k += two ? 2 : 1;
  }
  wave_sync();
  lds.lbest[lane] = mine;
  wave_sync();
  return s;
}

// ---- scoring runs, four jobs per slot of lanes ----------------------------------------
// Scores fit 16 bits (2 L <= 2048 in these kernels), so a lane can carry TWO jobs of a set in the halves of one
// register and advance both with one packed instruction (v_pk_add_i16 / v_pk_max_i16): with the two sets of the
// anti-diagonal schedule that is four jobs per slot.  For the halves to share the lane's control flow the slot is laid
// out around the band's centre: lane offset o, c = o - K (K = the slot's half width, its widest job's), and a job of
// half width k <= K owns the lanes |c| <= k.  In these coordinates the read index q = i + o - width and the row's
// target base pos + i - K - 1 are the same for every job of the slot (the reference's column is j = c + k, its row
// i_ref = i - (K - k): q_ref = i_ref + j - bw = q, target = t_beg + i_ref - 1), so validity by q, the row clock and the DPP
// moves are shared, and what differs per job is loop-invariant: the band mask (halves 0xFFFF inside the job's band), the
// edge masks of from_above (c < k) and from_left (c > -k), and the offset of its staged window (row 1's target base
// sits K - k nibbles earlier).  A masked-off candidate is 0, which never beats a cell value (>= 0).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)); }
__device__ __forceinline__ u32 pk_max(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
// maximum of both halves over the wave (values >= 0), uniform
__device__ __forceinline__ u32 wave_max_pk(u32 x) {
  x = pk_max(x, dpp_from<0x111, 0xf>(0u, x));
  x = pk_max(x, dpp_from<0x112, 0xf>(0u, x));
  x = pk_max(x, dpp_from<0x114, 0xf>(0u, x));
  x = pk_max(x, dpp_from<0x118, 0xf>(0u, x));
  x = pk_max(x, dpp_from<0x142, 0xa>(0u, x));
  x = pk_max(x, dpp_from<0x143, 0xc>(0u, x));
  return rdlane(x, 63);
}

struct QuadJob {         // per lane: the two jobs of one set that share this lane's slot (low half, high half)
  u32 mv, ma, ml;        // halves 0xFFFF: lane inside the job's band / may take from_above / may take from_left
  int qoff_lo, qoff_hi;  // word offsets of the query encodings in lds.qpk
  int g_lo, g_hi;        // genome-window slots in lds.gwin
  int gs_lo, gs_hi;      // nibble index of row 1's target base in the window, moved to the slot's band
};

// The loop (round 5).  At iteration n (anti-diagonal steps t_start + 2 n and + 2 n + 1) set P's cell has read index
// q = n + q0p and set Q's q = n + q0q, with per-lane constants q0: ((a + 2 n) >> 1 = (a >> 1) + n).  A cell needs its
// position only for three tests -- from_above exists if q < L - 1, from_left if q > 0, the cell itself if 0 <= q < L --
// and for all lanes at once they can only fail in the first and last few iterations (the lanes' q differ by at most a
// band's width).  So the iterations are cut in three: a head and a tail that make the tests, and a middle -- 1 <= q <=
// L - 2 in every lane, five sixths of the iterations of a 150-base read -- that does not: 15 vector instructions per
// packed cell against 28.  The match bits of a lane's next 16 cells sit one per cell in the halves of a 32-bit word
// (bit r: low job, bit 16 + r: high job), so "this cell's +2 / -3" is an AND and a packed multiply-add.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int wave_min_i32(int x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x = min(x, __shfl_xor(x, d));
  return x;
}
__device__ __forceinline__ void wavefront_quad(const WaveLds &lds, const QuadJob &jp, const QuadJob &jq, int dp, int dq,
                                               int o, int width, int L, int w_min, int w_max, u32 &bestp, u32 &bestq) {
  const int t_start = max(0, w_min - 1), t_end = 2 * (L - 1 + w_max) + 1;
  const int n_total = (t_end - t_start) / 2 + 1;
  const int W = static_cast<int>(lds.W), GW = static_cast<int>(lds.GW);
  const int q0p = ((t_start - dp - o) >> 1) + o - width, q0q = ((t_start + 1 - dq - o) >> 1) + o - width;
  // iterations [n_mid0, n_mid1): 1 <= q <= L - 2 for both cells of every lane that belongs to a slot (uniform)
  const bool in_slot = width != 0;
  const int q_lo = wave_min_i32(in_slot ? min(q0p, q0q) : 0x3fffffff), q_hi = wave_max_i32(in_slot ? max(q0p, q0q) : -0x3fffffff);
  const int n_mid0 = min(max(1 - q_lo, 0), n_total), n_mid1 = min(max(L - 1 - q_hi, n_mid0), n_total);
  u32 curp = 0, curq = 0, pub = 0, Mp = 0, Mq = 0;
  bestp = bestq = 0;
  auto refill = [&](const QuadJob &j, int q) -> u32 {  // match bits of the 16 cells from read index q on
    const int i0 = q - o + width;
    u64 lo = 0, hi = 0;
    if (j.mv & 0xFFFFu) lo = nibbles16(lds.qpk + j.qoff_lo, W, q) & nibbles16(lds.gwin + j.g_lo * GW, GW, j.gs_lo + i0 - 1);
    if (j.mv >> 16) hi = nibbles16(lds.qpk + j.qoff_hi, W, q) & nibbles16(lds.gwin + j.g_hi * GW, GW, j.gs_hi + i0 - 1);
    lo |= lo >> 1; lo |= lo >> 2;
    hi |= hi >> 1; hi |= hi >> 2;
    return every_fourth_bit(lo) | (every_fourth_bit(hi) << 16);
  };
  // +2 on a match, -3 otherwise, in both halves: 5 * bit - 3
  auto step_delta = [&](u32 &M) -> u32 {
    const u16x2 bits = __builtin_bit_cast(u16x2, M & 0x00010001u);
    M >>= 1;
    const u16x2 five = {5, 5}, minus3 = {0xFFFD, 0xFFFD};
    return __builtin_bit_cast(u32, static_cast<u16x2>(bits * five + minus3));
  };
  auto fast = [&](const QuadJob &j, u32 &cur, u32 &M, u32 &best) {
    const u32 lf = static_cast<u32>(from_prev_lane(static_cast<int>(pub))), up = static_cast<u32>(from_next_lane(static_cast<int>(pub)));
    u32 c = pk_max(pk_add(cur, step_delta(M)), 0u);
    c = pk_max(c, pk_add(up, 0xFFFCFFFCu) & j.ma);  // from_above
    c = pk_max(c, pk_add(lf, 0xFFFCFFFCu) & j.ml);  // from_left
    cur = c & j.mv;
    best = pk_max(best, cur);
    pub = cur;
  };
  auto slow = [&](const QuadJob &j, int q, u32 &cur, u32 &M, u32 &best) {
    const u32 lf = static_cast<u32>(from_prev_lane(static_cast<int>(pub))), up = static_cast<u32>(from_next_lane(static_cast<int>(pub)));
    u32 c = pk_max(pk_add(cur, step_delta(M)), 0u);
    c = pk_max(c, pk_add(up, 0xFFFCFFFCu) & (q < L - 1 ? j.ma : 0u));  // from_above
    c = pk_max(c, pk_add(lf, 0xFFFCFFFCu) & (q > 0 ? j.ml : 0u));      // from_left
    cur = static_cast<u32>(q) < static_cast<u32>(L) ? (c & j.mv) : 0u;
    best = pk_max(best, cur);
    pub = cur;
  };
  int n = 0;
#pragma clang loop unroll(disable)
  for (int seg = 0; seg < 3; ++seg) {
    const int end = seg == 0 ? n_mid0 : (seg == 1 ? n_mid1 : n_total);
    while (n < end) {
      if ((n & 15) == 0) { Mp = refill(jp, n + q0p); Mq = refill(jq, n + q0q); }
      const int stop = min(end, (n | 15) + 1);
      if (seg == 1) {
        for (; n < stop; ++n) { fast(jp, curp, Mp, bestp); fast(jq, curq, Mq, bestq); }
      }
      else {
        for (; n < stop; ++n) { slow(jp, n + q0p, curp, Mp, bestp); slow(jq, n + q0q, curq, Mq, bestq); }
      }
    }
  }
}

// score_round with four jobs per slot: job first+4m+h of slot m is set (h & 1), half (h >> 1)
__device__ __forceinline__ int score_round_quad(const DevIndex &ix, const WaveLds &lds, int first, int n_jobs, int L, int md,
                                                int qbase) {
  const int lane = lane_id();
  QuadJob ja = {0, 0, 0, 0, 0, 0, 0, 0, 0}, jb = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int used = 0, s = first, w_min = 64, w_max = 0, my_o = 0, my_w = 0;
  while (s < n_jobs) {
    const int cnt = min(4, n_jobs - s);
    if (s - first + cnt > static_cast<int>(lds.max_jobs)) break;
    int bwh[4], width = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      bwh[h] = h < cnt ? band_for(static_cast<int>(lds.jdf[s + h]) >> 16, md) : 0;
      width = max(width, bwh[h]);
    }
    if (used + width > 64) break;
    const int o = lane - used, K = (width - 1) / 2, c = o - K;
    const bool in = o >= 0 && o < width;
    if (in) { my_o = o; my_w = width; }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h >= cnt) break;
      const int k = (bwh[h] - 1) / 2;
      if (in && c >= -k && c <= k) {
        const u32 df = lds.jdf[s + h];
        const u64 t_beg = static_cast<u64>(lds.jpos[s + h]) - static_cast<u64>(k);
        const u32 half = (h >> 1) ? 0xFFFF0000u : 0x0000FFFFu;
        QuadJob &j = (h & 1) ? jb : ja;
        j.mv |= half;
        if (c < k) j.ma |= half;
        if (c > -k) j.ml |= half;
        const int qoff = qbase + static_cast<int>(enc_of(df & 0xFFFFu) * lds.W);
        const int gs = static_cast<int>(t_beg & 15u) - (K - k);
        if (h >> 1) { j.qoff_hi = qoff; j.g_hi = s + h - first; j.gs_hi = gs; }
        else { j.qoff_lo = qoff; j.g_lo = s + h - first; j.gs_lo = gs; }
      }
    }
    used += width;
    w_min = min(w_min, width);
    w_max = max(w_max, width);
    s += cnt;
  }
  stage_windows(ix, lds, first, s - first, md);
  wave_sync();
  const int t_start = max(0, w_min - 1);
  const bool even = ((t_start - my_o) & 1) == 0;  // which of the lane's sets is due on even steps
  u32 bp, bq;
  wavefront_quad(lds, even ? ja : jb, even ? jb : ja, even ? 0 : 1, even ? 1 : 0, my_o, my_w, L, w_min, w_max, bp, bq);
  const u32 best_a = even ? bp : bq, best_b = even ? bq : bp;
  int base = 0, mine = 0;
  for (int k = first; k < s;) {
    const int cnt = min(4, s - k);
    int width = 0;
#pragma unroll
    for (int h = 0; h < 4; ++h)
      if (h < cnt) width = max(width, band_for(static_cast<int>(lds.jdf[k + h]) >> 16, md));
    const bool in = lane >= base && lane < base + width;
    const u32 xa = wave_max_pk(in ? best_a : 0u), xb = cnt > 1 ? wave_max_pk(in ? best_b : 0u) : 0u;
    if (lane == k - first) mine = static_cast<int>(xa & 0xFFFFu);
    if (lane == k + 1 - first && cnt > 1) mine = static_cast<int>(xb & 0xFFFFu);
    if (lane == k + 2 - first && cnt > 2) mine = static_cast<int>(xa >> 16);
    if (lane == k + 3 - first && cnt > 3) mine = static_cast<int>(xb >> 16);
    base += width;
    k += cnt;
  }
  wave_sync();
  lds.lbest[lane] = mine;
  wave_sync();
  return s;
}

// a round of scoring runs: four jobs per slot (packed cells) when at least kQuadMinJobs are left, else two (plain cells);
// the long-read kernel's 16-bit wrapping arithmetic stays on the plain cells.  The packed cell costs a quarter more
// than the plain one, so by instruction count it pays from a slot's third job on -- but the kernel that holds only one
// of the two loops is the faster one (10 M reads, same box, profiles/r03_exp_quad_scoring.log: never 701-712 ms, from
// three jobs on 685-695 ms, always 678-681 ms).
#ifndef ABM_QUAD_MIN_JOBS
#define ABM_QUAD_MIN_JOBS 1
#endif
constexpr int kQuadMinJobs = ABM_QUAD_MIN_JOBS;
template <bool WRAP = false>
__device__ __forceinline__ int score_jobs(const DevIndex &ix, const WaveLds &lds, int first, int n_jobs, int L, int md, int qbase) {
  if constexpr (!WRAP) {
    if (n_jobs - first >= kQuadMinJobs) return score_round_quad(ix, lds, first, n_jobs, L, md, qbase);
  }
  return score_round<WRAP>(ix, lds, first, n_jobs, L, md, qbase);
}

// align_se_candidates (src/abismal.cpp:1435-1497) on the wave-resident set
template <bool WRAP = false>
__device__ __forceinline__ void choose_se(const DevIndex &ix, const WaveLds &lds, u32 L, double frac,
                                          SeSet &S, Hit &best, u32 *cig_out, const CigarSink &sink,
                                          u32 &n_ops, bool &overflow, u32 &n_aln, u32 &n_single) {
  const int lane = lane_id();
  const int Ls = static_cast<i16>(L);
  const int md = static_cast<i16>(frac * static_cast<u32>(Ls));  // valid_diffs_cutoff
  const int perfect = static_cast<i16>(2 * L);
  n_ops = 0;
  if (S.best_p != 0) {  // exact match: no alignment needed
    best.diffs = static_cast<i16>(S.best_d); best.flags = static_cast<u16>(S.best_f); best.pos = S.best_p;
    if (lane == 0) { store_out(cig_out, L << 4); if (sink.fin) sink.fin[0] = L << 4; }
    n_ops = 1;
    return;
  }
  // prepare_for_alignments: order by (pos, flags), drop duplicates.  Lane k looks
  // at heap entry k; its payload sits in lane (key & 255).
  const bool mine = lane < S.sz;
  const int slot_of = S.hk & 255;
  const u32 e_pos = __shfl(S.pp, slot_of), e_flags = __shfl(S.pf, slot_of);
  const int e_d = SeSet::key_d(S.hk);
  const u64 key = (static_cast<u64>(e_pos) << 16) | e_flags;
  bool dup = false;
  for (int k = 0; k < S.sz; ++k) {
    const u64 kk = rdlane(key, k);
    dup |= (mine && k < lane && kk == key);
  }
  // jobs = unique entries that the reference would align: non-empty, diffs < 0.4L
  const int invalid_at = static_cast<i16>(0.4 * Ls);  // valid_hit, :323-326
  const bool is_job = mine && !dup && e_pos != 0 && e_d < invalid_at;
  const u64 jobs = __ballot(is_job);
  int rank = 0;
  for (int k = 0; k < S.sz; ++k) {
    const u64 kk = rdlane(key, k);
    rank += ((jobs >> k) & 1) && kk < key;
  }
  const int n_jobs = __popcll(jobs);
  if (is_job) {
    lds.jpos[rank] = e_pos;
    lds.jdf[rank] = (static_cast<u32>(e_d) << 16) | e_flags;
  }
  wave_sync();

  int top = 0;
  u32 top_pos = 0, b_pos = 0, b_flags = 0;
  int b_diffs = 0x7fff;
  // A set with ONE alignable entry (most reads that map uniquely with a mismatch or two): the reference scores it
  // (align<false>) and then aligns it again with traceback (align<true>) -- the same table twice, so the traceback
  // run's score is the scoring run's, and the scoring run is skipped.
  const bool single = n_jobs == 1;
  if (single) {
    const u32 df = lds.jdf[0];
    b_diffs = static_cast<int>(df) >> 16; b_flags = df & 0xFFFFu; b_pos = lds.jpos[0];
    ++n_aln;
    ++n_single;
  }
  for (int s = 0; s < n_jobs && !single;) {
    const int first = s;
    s = score_jobs<WRAP>(ix, lds, first, n_jobs, static_cast<int>(L), md, 0);
    // apply the reference's selection in job order
    for (int k = first; k < s; ++k) {
      const u32 df = lds.jdf[k];
      const int d = static_cast<int>(df) >> 16;
      const int sc = static_cast<i16>(lds.lbest[k - first]);
      const u32 pos = lds.jpos[k], flags = df & 0xFFFFu;
      ++n_aln;
      if (sc > top) { b_diffs = d; b_flags = flags; b_pos = pos; top = sc; top_pos = pos; }
      else if (sc == top) {
        const u32 gap = pos > top_pos ? pos - top_pos : top_pos - pos;
        if (sc == perfect ? pos != top_pos : gap > 3u) b_flags |= kFlagAmbig;
      }
    }
    wave_sync();
  }
  best.diffs = 0x7fff; best.flags = static_cast<u16>(b_flags); best.pos = 0;
  if (b_pos == 0)
    return;
  // traceback run of the winner: one job, band in lanes [0, bw)
  const int bw = band_for(b_diffs, md);
  AlnJob job = {0, 0, 0, 0, 0};
  const u64 t_beg = static_cast<u64>(b_pos) - static_cast<u64>((bw - 1) / 2);
  if (lane < bw) {
    job.bw = bw;
    job.jl = lane;
    job.qoff = static_cast<int>(enc_of(b_flags) * lds.W);
    job.t0nib = static_cast<int>(t_beg & 15u);
  }
  if (lane == 0) { lds.jpos[0] = b_pos; lds.jdf[0] = (static_cast<u32>(b_diffs) << 16) | (b_flags & 0xFFFFu); }
  wave_sync();
  stage_windows(ix, lds, 0, 1, md);
  wave_sync();
  int bv, brow;
  if constexpr (WRAP || !kTracebackByRows) wavefront<true, WRAP>(lds, job, static_cast<int>(L), bw, bw, bv, brow);
  else wavefront_rows<true>(lds, job, static_cast<int>(L), bw, bv, brow);
  // first maximum in row-major order: max value, then smallest row, then smallest column
  const u64 k64 = (static_cast<u64>(static_cast<u32>(bv)) << 32) |
                  (static_cast<u64>(0xFFFFu - static_cast<u32>(brow)) << 8) |
                  static_cast<u64>(0xFFu - static_cast<u32>(lane));
  const u64 topk = wave_max_u64(lane < bw ? k64 : 0ull);
  const int br = static_cast<int>(0xFFFFu - static_cast<u32>((topk >> 8) & 0xFFFFu));
  const int bc = static_cast<int>(0xFFu - static_cast<u32>(topk & 0xFFu));
  const int sc = static_cast<i16>(static_cast<int>(topk >> 32));
  wave_sync();
  if (single) {  // what the scoring run would have found
    top = sc;
    if (sc <= 0) { best.flags = static_cast<u16>(kFlagAmbig); return; }  // (a zero score ties with "nothing yet" at position 0)
  }
  u32 alen = 0, pos = b_pos;
  int n_ins = 0, n_del = 0;
  wave_cigar(lds.tb, lds.ctmp, static_cast<int>(L), b_diffs, md, sc, br, bc, cig_out, sink, n_ops,
             n_ins, n_del, alen, pos, overflow);
  wave_sync();
  // NM from the score found by the scoring pass (best_scr), as the reference does
  const int nm = edit_distance(top, alen, n_ins, n_del);
  if (long_enough(alen, static_cast<u32>(Ls), ix.min_len) && nm <= md) {
    best.diffs = static_cast<i16>(nm);
    best.pos = pos;
  }
  // (a hit that fails here leaves best.pos == 0; n_ops still counts the CIGAR the traceback wrote into the slot,
  // like the reference's r.cig, so that count and slot -- or arena reference -- always belong together)
}


}  // namespace abm
