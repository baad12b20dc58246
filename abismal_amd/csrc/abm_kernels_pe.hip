// abismal_amd HIP kernels for gfx950: paired-end mapping, one wavefront per pair.
// Restates map_paired_ended / map_paired_ended_rand's per-pair body
// (src/abismal.cpp:1950-1999, :2094-2155): two or four orientation calls
// (map_fragments :1849-1885), each seeding both ends into growable candidate
// sets (pe_candidates :775-863), mating them (best_pair :1722-1831), feeding the
// single-end sets (best_single :1715-1720), then valid_pair and the single-end
// fallback.  Two tiers run the same code: tier 1 keeps sets of up to 256
// entries in LDS at normal occupancy; pairs whose sets outgrow that are redone
// by tier 2 with one wave per CU and a 32768-entry set in LDS.
#include "abm_kernels_core.hpp"

namespace abm {

struct PeLds {
  u32 *heap;     // [cap] live set as a binary heap of diffs<<16 | handle; reused as sort buffer
  u32 *lpos[2];  // finished sets of the two ends of this orientation call: positions
  i16 *ld[2];    //   ... diffs
  i16 *lsc[2];   //   ... alignment scores of the entries that can pair up
  u32 *jidx;     // [kSeCap] which list entry an alignment job belongs to
  u32 cap;
};

template <bool BIG> __device__ __forceinline__ u32 ld_list(const u32 *p) {
  return BIG ? __builtin_nontemporal_load(p) : *p;
}
template <bool BIG> __device__ __forceinline__ int ld_list(const i16 *p) {
  return BIG ? static_cast<int>(__builtin_nontemporal_load(p)) : static_cast<int>(*p);
}

// pe_candidates, src/abismal.cpp:775-863.  Wave-uniform state + heap in LDS;
// positions live in a per-wave global table indexed by a recycled handle.
struct PeSet {
  static constexpr bool kFifo = false;
  u32 *heap;
  u32 *payload;
  u32 cap_avail;
  int sz, capacity, cutoff, good_cutoff;
  bool sure_ambig, overflow;

  __device__ __forceinline__ static int key_d(u32 k) { return static_cast<int>(k) >> 16; }
  __device__ __forceinline__ u32 rd(int i) const { return static_cast<u32>(uni(static_cast<int>(heap[i]))); }
  __device__ __forceinline__ void begin_read(u32 readlen) {
    const int worst = static_cast<i16>(0.4 * readlen);
    heap[0] = static_cast<u32>(worst) << 16;  // sentinel, handle 0 -> pos 0
    if (lane_id() == 0) payload[0] = 0;
    sz = 1;
    capacity = static_cast<int>(kPeCapSmall);
    cutoff = worst;
    good_cutoff = static_cast<i16>(readlen / 10);
    sure_ambig = false;
    overflow = false;
  }
  __device__ __forceinline__ bool wants_sensitive() const {
    return capacity == static_cast<int>(kPeCapSmall) || cutoff > good_cutoff;
  }
  __device__ __forceinline__ void sift_up(int hole, u32 key) {
    while (hole > 0) {
      const int parent = (hole - 1) / 2;
      const u32 pk = rd(parent);
      if (!(key_d(pk) < key_d(key))) break;
      heap[hole] = pk;
      hole = parent;
    }
    heap[hole] = key;
  }
  __device__ __forceinline__ int pop_max(int n) {  // libstdc++ pop_heap on [0,n); see SeSet::pop_max
    const int len = n - 1;
    const int freed = static_cast<int>(rd(0) & 0x7FFFu);
    const u32 vk = rd(len);
    int hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      u32 sk = rd(second);
      const u32 lk = rd(second - 1);
      if (key_d(sk) < key_d(lk)) { --second; sk = lk; }
      heap[hole] = sk;
      hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      heap[hole] = rd(second - 1);
      hole = second - 1;
    }
    sift_up(hole, vk);
    return freed;
  }
  // pe_candidates::update, :824-842
  __device__ __forceinline__ void admit(bool specific, int d, u32 /*flags*/, u32 p) {
    if (overflow) return;
    int handle;
    if (sz == capacity) {
      if (specific && capacity != static_cast<int>(kPeCapLarge) && d <= good_cutoff) {
        if (capacity == static_cast<int>(cap_avail)) {  // tier 1 ran out of room: redo in tier 2
          overflow = true;
          sure_ambig = true;
          return;
        }
        ++capacity;
        handle = sz;
      }
      else {
        handle = pop_max(sz);
        --sz;
      }
    }
    else
      handle = sz;
    if (lane_id() == 0) payload[handle] = p;
    ++sz;
    sift_up(sz - 1, (static_cast<u32>(d) << 16) | static_cast<u32>(handle));
    const int top = key_d(rd(0));
    cutoff = specific ? min(cutoff, top) : top;
    sure_ambig = (sz == capacity) && cutoff == 0;
  }
};

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

template <bool BIG> struct PeWave {
  const PeArgs &a;
  WaveLds lds;      // qpk/qbits point at end 0; end 1 follows at +4W / +4WB
  PeLds pl;
  WorkTally wt;
  u32 n_aln;
  bool overflow, need_big;
  u32 L[2];
  SeSet se[2];
  PeSet P;
  // finished-set bookkeeping for the orientation call in flight (index 0 = endA)
  int lsz[2];
  bool worth[2];
  u32 lflags[2];
  // CIGAR bookkeeping per read end
  u32 n_ops[2], ref_len[2];

  __device__ __forceinline__ WaveLds lds_of(int end) const {
    WaveLds w = lds;
    w.qpk = lds.qpk + end * 4 * lds.W;
    w.qbits = lds.qbits + end * 4 * lds.WB;
    return w;
  }
  __device__ __forceinline__ u32 *cig_of(int end, u64 r) const { return (end ? a.cig2 : a.cig1) + r * a.cig_stride; }

  // ---- one end of one orientation call: both seed passes, then freeze the set ----
  template <bool TIMED> __device__ __forceinline__ void seed_end(int which, int end, bool rc, bool ar) {
    const WaveLds w = lds_of(end);
    const bool g_to_a = rc != ar;
    const u32 enc = (rc ? 2u : 0u) + (g_to_a ? 1u : 0u);
    const u32 flags = (rc ? kFlagRC : 0u) | (ar ? kFlagARich : 0u);
    lflags[which] = flags;
    P.begin_read(L[end]);
    if (L[end] >= kMinReadLen) {
      P.cutoff = P.good_cutoff;  // set_specific
      seed_pass<true, TIMED>(a.ix, w, enc, g_to_a, flags, L[end], P, wt);
      if (!P.overflow && P.wants_sensitive()) {
        P.cutoff = PeSet::key_d(P.rd(0));  // set_sensitive
        seed_pass<false, TIMED>(a.ix, w, enc, g_to_a, flags, L[end], P, wt);
      }
    }
    need_big |= P.overflow;
    // freeze: the list keeps heap-array order (what best_single replays if no mating happens)
    const int n = P.sz;
    __syncthreads();
    for (int i = lane_id(); i < n; i += 64) {
      const u32 e = P.heap[i];
      pl.lpos[which][i] = __builtin_nontemporal_load(P.payload + (e & 0x7FFFu));
      pl.ld[which][i] = static_cast<i16>(static_cast<int>(e) >> 16);
    }
    lsz[which] = n;
    worth[which] = n != static_cast<int>(kPeCapLarge) || P.cutoff != 0;  // should_align, :799-802
    __syncthreads();
  }

  // prepare_for_mating (:844-852): sort by position, drop duplicates; diffs are
  // then recomputed (the Hamming distance is a function of the position alone)
  __device__ __forceinline__ void sort_unique(int which, int end) {
    const int lane = lane_id();
    const int n = lsz[which];
    u32 *buf = pl.heap;  // the live heap is dead by now
    int m = 1;
    while (m < n) m <<= 1;
    for (int i = lane; i < m; i += 64) buf[i] = i < n ? ld_list<BIG>(pl.lpos[which] + i) : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < m; i += 64) {
          const int p = i ^ j;
          if (p > i) {
            const u32 x = buf[i], y = buf[p];
            const bool up = (i & k) == 0;
            if ((x > y) == up) { buf[i] = y; buf[p] = x; }
          }
        }
        __syncthreads();
      }
    // unique + recompute diffs
    const WaveLds w = lds_of(end);
    const u64 *qpk = w.qpk + enc_of(lflags[which]) * w.W;
    const u32 nwords = (L[end] + 15) >> 4;
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
                                                    : static_cast<int>(static_cast<i16>(0.4 * L[end])));
      }
      out += __popcll(kept);
    }
    lsz[which] = out;
    __syncthreads();
  }

  // first index whose position is >= key (lists are sorted here)
  __device__ __forceinline__ int lower_bound_pos(int which, long long key) const {
    int lo = 0, n = lsz[which];
    while (n > 0) {
      const int half = n >> 1;
      if (static_cast<long long>(ld_list<BIG>(pl.lpos[which] + lo + half)) < key) { lo += half + 1; n -= half + 1; }
      else n = half;
    }
    return lo;
  }

  // scores of every entry that has at least one concordant partner: the only
  // alignments best_pair can ever request (it computes them lazily, :1783-1790)
  __device__ __forceinline__ void score_pairable(int endA) {
    const int lane = lane_id();
    const int ends[2] = {endA, 1 - endA};
    const long long lenB = L[ends[1]];
    // pass 1: mark.  score 0 = cannot pair, 2L = exact hit (align() returns at once), -1 = needs the DP
    for (int which = 0; which < 2; ++which) {
      const int other = 1 - which, n = lsz[which];
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
            mark = ld_list<BIG>(pl.ld[which] + i) == 0 ? static_cast<int>(static_cast<i16>(2 * L[ends[which]])) : -1;
        }
        pl.lsc[which][i] = static_cast<i16>(mark);
      }
    }
    __syncthreads();
    // pass 2: run the marked entries through the wavefront DP, up to kSeCap jobs at a time
    for (int which = 0; which < 2; ++which) {
      const int end = ends[which], n = lsz[which];
      const int md = static_cast<i16>(a.valid_frac * L[end]);
      const int qoff = static_cast<int>((end * 4 + enc_of(lflags[which])) * lds.W);
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
          __syncthreads();
          if (take < cnt) break;  // list full: the rest of this chunk is picked up next time
          cursor += 64;
        }
        if (n_jobs == 0) break;
        for (int s = 0; s < n_jobs;) {  // rounds of side-by-side bands
          AlnJob job = {0, 0, 0, 0, 0};
          int used = 0, first = s, bw_min = 64, bw_max = 0;
          while (s < n_jobs) {
            const u32 df = lds.jdf[s];
            const int bw = band_for(static_cast<int>(df) >> 16, md);
            if (used + bw > 64) break;
            if (lane >= used && lane < used + bw) {
              const u64 t_beg = static_cast<u64>(lds.jpos[s]) - static_cast<u64>((bw - 1) / 2);
              job.bw = bw; job.jl = lane - used; job.qoff = qoff; job.g = s - first;
              job.t0nib = static_cast<int>(t_beg & 15u);
            }
            used += bw; bw_min = min(bw_min, bw); bw_max = max(bw_max, bw);
            ++s;
          }
          stage_windows(a.ix, lds, first, s - first, md);
          __syncthreads();
          int bv, br;
          wavefront<false>(lds, job, static_cast<int>(L[end]), bw_min, bw_max, bv, br);
          lds.lbest[lane] = bv;
          __syncthreads();
          int base = 0;
          for (int k = first; k < s; ++k) {
            const int bw = band_for(static_cast<int>(lds.jdf[k]) >> 16, md);
            int sc = lane < bw ? lds.lbest[base + lane] : 0;
            sc = static_cast<i16>(static_cast<int>(wave_max_u64(static_cast<u64>(static_cast<u32>(sc)))));
            base += bw;
            if (lane == 0) pl.lsc[which][pl.jidx[k]] = static_cast<i16>(sc);
            ++n_aln;
          }
          __syncthreads();
        }
      }
    }
  }

  // traceback of one end's winning hit; updates (pos, diffs) and the end's CIGAR
  __device__ __forceinline__ void traceback_end(int end, u64 r, int scoring, int &d, u32 flags, u32 &pos,
                                                u32 &alen) {
    const int lane = lane_id();
    const int Ln = static_cast<int>(L[end]);
    const int md = static_cast<i16>(a.valid_frac * L[end]);
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
      __syncthreads();
      stage_windows(a.ix, lds, 0, 1, md);
      __syncthreads();
      int bv, brow;
      wavefront<true>(lds, job, Ln, bw, bw, bv, brow);
      const u64 k64 = (static_cast<u64>(static_cast<u32>(bv)) << 32) |
                      (static_cast<u64>(0xFFFFu - static_cast<u32>(brow)) << 8) |
                      static_cast<u64>(0xFFu - static_cast<u32>(lane));
      const u64 topk = wave_max_u64(lane < bw ? k64 : 0ull);
      const int br = static_cast<int>(0xFFFFu - static_cast<u32>((topk >> 8) & 0xFFFFu));
      const int bc = static_cast<int>(0xFFu - static_cast<u32>(topk & 0xFFu));
      const int sc = static_cast<i16>(static_cast<int>(topk >> 32));
      __syncthreads();
      wave_cigar(lds.tb, lds.ctmp, Ln, d, md, sc, br, bc, cig_out, a.cig_stride, nops, n_ins, n_del, alen, pos,
                 overflow);
      __syncthreads();
    }
    d = edit_distance(scoring, alen, n_ins, n_del);
    n_ops[end] = nops;
    ref_len[end] = alen - static_cast<u32>(n_ins) + static_cast<u32>(n_del);
  }

  // best_pair, src/abismal.cpp:1722-1831.  List 0 is endA's (the reference's
  // res1), list 1 endB's; `swapped` = endA is read 2.
  __device__ __forceinline__ void mate(int endA, u64 r, PairBest &best) {
    const int endB = 1 - endA;
    const bool swapped = endA == 1;
    const long long na = lsz[0], nb = lsz[1];
    const u32 lenB = L[endB];
    auto posA = [&](long long i) { return static_cast<u32>(uni(static_cast<int>(ld_list<BIG>(pl.lpos[0] + i)))); };
    auto posB = [&](long long i) { return static_cast<u32>(uni(static_cast<int>(ld_list<BIG>(pl.lpos[1] + i)))); };
    long long ia = 0, ib = 0;
    while (ia != na && posA(ia) == 0) ++ia;
    while (ib != nb && posB(ib) == 0) ++ib;
    int last_sa = 0, keep_sa = 0, keep_sb = 0;
    u32 keep_pa = 0, keep_pb = 0;
    // The reference aligns an A entry at its first visit and memoises the score (an entry that
    // scored 0 is realigned at every visit, with the same result); `last_sa` is the score of the
    // most recent such alignment, which is what best_scr1 ends up holding (:1787-1795).
    for (; ib != nb && !best.sure_ambig(); ++ib) {
      const u32 pb = posB(ib);
      const int db = uni(ld_list<BIG>(pl.ld[1] + ib));
      int sb = 0;
      const u32 frag_end = pb + lenB;
      while (ia == na || (ia != 0 && posA(ia) + a.max_frag >= frag_end)) --ia;
      while (ia != na && posA(ia) + a.max_frag < frag_end) ++ia;
      for (; ia != na && posA(ia) + a.min_frag <= frag_end && !best.sure_ambig(); ++ia) {
        const u32 pa = posA(ia);
        const int da = uni(ld_list<BIG>(pl.ld[0] + ia));
        if (sb == 0) sb = uni(ld_list<BIG>(pl.lsc[1] + ib));
        // "if (*a1 == 0) { scr1 = align(...); *a1 = scr1; }": visited[] marks first visits
        const int sa = uni(ld_list<BIG>(pl.lsc[0] + ia));
        if (!visited_test_and_set(ia) || sa == 0) last_sa = sa;
        const int pair = static_cast<i16>(sb + sa);
        const bool better = swapped ? best.offer(pair, db, lflags[1], pb, da, lflags[0], pa)
                                    : best.offer(pair, da, lflags[0], pa, db, lflags[1], pb);
        if (better) { keep_sa = last_sa; keep_sb = sb; keep_pa = pa; keep_pb = pb; }
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

  // visited bitmap for list 0, kept in the (now idle) score array's sign bit would alter
  // scores, so a separate bitmap lives at the tail of the sort buffer (heap area)
  __device__ __forceinline__ bool visited_test_and_set(long long i) {
    u32 *bm = pl.heap;  // cap/32 words are enough; the heap is dead during mating
    const u32 w = static_cast<u32>(uni(static_cast<int>(bm[i >> 5])));
    const u32 bit = 1u << (i & 31);
    if (w & bit) return true;
    bm[i >> 5] = w | bit;
    return false;
  }

  // map_fragments + select_maps + best_single (:1715-1720, :1833-1885)
  template <bool TIMED> __device__ __forceinline__ bool orientation(int endA, bool ar, u64 r, PairBest &best) {
    const int endB = 1 - endA;
    const bool emptyA = L[endA] < kMinReadLen, emptyB = L[endB] < kMinReadLen;
    if (emptyA && emptyB) {
      // res1/res2 are reset and nothing else happens (:1863-1866)
      return false;
    }
    // endA forward with (rc=0, a_rich=ar); endB reverse-complemented with (rc=1, a_rich=!ar)
    seed_end<TIMED>(0, endA, false, ar);
    seed_end<TIMED>(1, endB, true, !ar);
    if (need_big && !BIG)
      return true;
    if (worth[0] && worth[1]) {
      sort_unique(0, endA);
      sort_unique(1, endB);
      score_pairable(endA);
      // clear the visited bitmap
      for (int i = lane_id(); i < (lsz[0] + 31) / 32; i += 64) pl.heap[i] = 0;
      __syncthreads();
      mate(endA, r, best);
    }
    // best_single: every entry of each set, in array order, into that end's single-end set
    const int ends[2] = {endA, endB};
    for (int which = 0; which < 2; ++which) {
      SeSet &S = se[ends[which]];
      for (int k = 0; k < lsz[which] && !S.sure_ambig; ++k) {
        const int d = uni(ld_list<BIG>(pl.ld[which] + k));
        const u32 p = static_cast<u32>(uni(static_cast<int>(ld_list<BIG>(pl.lpos[which] + k))));
        S.admit(false, d, lflags[which], p);
        ++wt.updates;
      }
    }
    return true;
  }
};

template <bool BIG, bool TIMED>
__global__ __launch_bounds__(64) void map_pe_kernel(PeArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  PeWave<BIG> w{a};
  WaveLds &lds = w.lds;
  lds.W = a.W; lds.WB = a.WB; lds.GW = a.GW;
  lds.qpk = reinterpret_cast<u64 *>(smem);
  lds.qbits = lds.qpk + 8 * a.W;
  lds.pcache = lds.qbits + 8 * a.WB;
  lds.gwin = lds.pcache + (1u << kPosCacheBits);
  lds.ctmp = reinterpret_cast<u32 *>(lds.gwin + kMaxJobs * a.GW);
  lds.jpos = lds.ctmp + a.cig_stride;
  lds.jdf = lds.jpos + kSeCap;
  w.pl.jidx = lds.jdf + kSeCap;
  lds.lbest = reinterpret_cast<int *>(w.pl.jidx + kSeCap);
  // tier 1: the set's heap is in LDS.  tier 2: a 32768-entry heap per wave would cap the CU at one
  // wave, so it lives in global memory (L2-resident, touched by this wave only) and occupancy stays normal
  u32 *after_heap = reinterpret_cast<u32 *>(lds.lbest + 64);
  if (BIG) w.pl.heap = a.heap_ws + static_cast<u64>(blockIdx.x) * a.cap;
  else { w.pl.heap = after_heap; after_heap += a.cap; }
  w.pl.cap = a.cap;
  if (BIG) {
    u32 *ws = a.list_ws + static_cast<u64>(blockIdx.x) * 4 * a.cap;  // 2 pos arrays + (2 diffs + 2 scores) as i16
    w.pl.lpos[0] = ws; w.pl.lpos[1] = ws + a.cap;
    i16 *h = reinterpret_cast<i16 *>(ws + 2 * a.cap);
    w.pl.ld[0] = h; w.pl.ld[1] = h + a.cap; w.pl.lsc[0] = h + 2 * a.cap; w.pl.lsc[1] = h + 3 * a.cap;
  }
  else {
    w.pl.lpos[0] = after_heap; w.pl.lpos[1] = after_heap + a.cap;
    i16 *h = reinterpret_cast<i16 *>(after_heap + 2 * a.cap);
    w.pl.ld[0] = h; w.pl.ld[1] = h + a.cap; w.pl.lsc[0] = h + 2 * a.cap; w.pl.lsc[1] = h + 3 * a.cap;
    after_heap = reinterpret_cast<u32 *>(h + 4 * a.cap);
  }
  lds.mark = reinterpret_cast<u16 *>(after_heap);
  lds.tb = reinterpret_cast<u8 *>(lds.mark + 64);

  w.P.heap = w.pl.heap;
  w.P.payload = a.payload_ws + static_cast<u64>(blockIdx.x) * a.cap;
  w.P.cap_avail = a.cap;
  w.wt = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  w.n_aln = 0;
  w.overflow = false;
  bool too_long = false;

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
    const u64 r = BIG ? static_cast<u64>(a.subset[it]) : (a.order ? static_cast<u64>(a.order[it]) : it);
    w.L[0] = a.lens1[r];
    w.L[1] = a.lens2[r];
    if (w.L[0] > kMaxReadLen || w.L[1] > kMaxReadLen) { too_long = true; w.L[0] = w.L[1] = 0; }
    // stage both ends' four encodings and their 2-letter bit strings
    for (int e = 0; e < 2; ++e) {
      const u64 *src = (e ? a.packed2 : a.packed1) + r * 4 * a.W;
      for (u32 k = lane; k < 4 * a.W; k += 64) lds.qpk[e * 4 * a.W + k] = src[k];
    }
    __syncthreads();
    for (u32 e8 = 0; e8 < 8; ++e8)
      for (u32 wb = 0; wb < a.WB; ++wb) {
        const u32 j = wb * 64 + lane, Le = w.L[e8 >> 2];
        const bool b = j < Le ? bit2(q_nibble(lds.qpk + e8 * a.W, j)) : true;
        const u64 word = __ballot(b);
        if (lane == 0) lds.qbits[e8 * a.WB + wb] = word;
      }
    __syncthreads();

    PairBest best;
    best.f1 = best.f2 = 0;
    best.clear(w.L[0] >= kMinReadLen ? w.L[0] : 0u, w.L[1] >= kMinReadLen ? w.L[1] : 0u);
    w.se[0].begin_read(w.L[0] >= kMinReadLen ? w.L[0] : 0u);
    w.se[1].begin_read(w.L[1] >= kMinReadLen ? w.L[1] : 0u);
    w.need_big = false;
    w.n_ops[0] = w.n_ops[1] = 0;
    w.ref_len[0] = w.ref_len[1] = 0;

    bool any = false;
    // orientation(endA, alphabet): src/abismal.cpp:1963-1979, :2106-2133
    if (a.mode == 2) {
      any |= w.template orientation<TIMED>(0, false, r, best);
      if (!(w.need_big && !BIG)) any |= w.template orientation<TIMED>(1, true, r, best);
      if (!(w.need_big && !BIG)) any |= w.template orientation<TIMED>(0, true, r, best);
      if (!(w.need_big && !BIG)) any |= w.template orientation<TIMED>(1, false, r, best);
    }
    else {
      const bool ar = a.mode == 1;
      any |= w.template orientation<TIMED>(0, ar, r, best);
      if (!(w.need_big && !BIG)) any |= w.template orientation<TIMED>(1, !ar, r, best);
    }
    if (w.need_big && !BIG) {
      if (lane == 0) a.need_big[r] = 1;
      continue;
    }
    if (!any) {  // :1981-1985
      best.clear();
      w.se[0].begin_read(0); w.se[0].hk = 32767 * 256; w.se[0].cutoff = 32767; w.se[0].best_d = 32767;
      w.se[1].begin_read(0); w.se[1].hk = 32767 * 256; w.se[1].cutoff = 32767; w.se[1].best_d = 32767;
    }
    {  // valid_pair, :624-631, on whatever CIGARs the mating left behind
      const u32 a1 = w.ref_len[0], a2 = w.ref_len[1];
      const bool ok = long_enough(a1, w.L[0]) && long_enough(a2, w.L[1]) &&
                      static_cast<i16>(best.d1 + best.d2) <= static_cast<i16>(a.valid_frac * (a1 + a2));
      if (!ok) best.clear();
    }
    Hit h1, h2;
    h1.diffs = static_cast<i16>(0.4 * w.L[0]); h1.flags = 0; h1.pos = 0;
    h2.diffs = static_cast<i16>(0.4 * w.L[1]); h2.flags = 0; h2.pos = 0;
    if (!best.should_report(a.allow_ambig != 0)) {  // single-end fallback at half the error budget
      for (int e = 0; e < 2; ++e) {
        const WaveLds we = w.lds_of(e);
        u32 nops = 0;
        Hit &h = e ? h2 : h1;
        choose_se(a.ix, we, w.L[e], a.valid_frac / 2, w.se[e], h, w.cig_of(e, r), a.cig_stride, nops,
                  w.overflow, w.n_aln);
        if (h.pos != 0 || w.se[e].best_p != 0) w.n_ops[e] = nops;
      }
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
      if (!BIG) a.need_big[r] = 0;
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
    }
  }
  if (lane == 0 && (w.overflow || too_long))
    atomicOr(a.status, (w.overflow ? 1u : 0u) | (too_long ? 2u : 0u));
}

// compact the pairs flagged by tier 1 into a list for tier 2
__global__ __launch_bounds__(256) void collect_big_kernel(const u8 *__restrict__ need_big, u64 n,
                                                          u32 *__restrict__ subset, u32 *__restrict__ count) {
  const u64 r = static_cast<u64>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r < n && need_big[r]) subset[atomicAdd(count, 1u)] = static_cast<u32>(r);
}

size_t pe_lds_bytes(u32 W, u32 WB, u32 GW, u32 cig_stride, u32 max_len, double valid_frac, u32 cap, bool big) {
  const int md = static_cast<i16>(valid_frac * max_len);
  int bw = 2 * md + 1;
  if (bw > static_cast<int>(kMaxBand) || bw < 1) bw = kMaxBand;
  size_t b = static_cast<size_t>(8) * W * 8 + static_cast<size_t>(8) * WB * 8 + (static_cast<size_t>(8) << kPosCacheBits) + static_cast<size_t>(kMaxJobs) * GW * 8 +
             static_cast<size_t>(cig_stride) * 4 + 3 * kSeCap * 4 + 64 * 4 + 64 * 2;
  if (!big) b += static_cast<size_t>(cap) * (4 + 2 * 4 + 4 * 2);
  b += static_cast<size_t>(max_len + bw) * bw;
  return (b + 15) & ~static_cast<size_t>(15);
}

int pe_resident_waves(size_t lds, bool big) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  const hipError_t e = big ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<true, false>, 64, lds)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_pe_kernel<false, false>, 64, lds);
  if (e != hipSuccess) return 0;
  return per_cu * prop.multiProcessorCount;
}

hipError_t launch_map_pe(const PeArgs &a, size_t lds, u32 grid, bool big, hipStream_t st) {
  if (grid == 0) return hipSuccess;
  if (big)
    hipLaunchKernelGGL((map_pe_kernel<true, false>), dim3(grid), dim3(64), lds, st, a);
  else
    hipLaunchKernelGGL((map_pe_kernel<false, false>), dim3(grid), dim3(64), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_collect_big(const u8 *need_big, u64 n, u32 *subset, u32 *count, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(count, 0, sizeof(u32), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(collect_big_kernel, dim3(static_cast<u32>((n + 255) / 256)), dim3(256), 0, st, need_big, n, subset, count);
  return hipGetLastError();
}

}  // namespace abm
