// ORACLE — test infrastructure only (see abo_common.hpp).
#include "abo_map.hpp"

#include <algorithm>
#include <cstring>

namespace abo {

u32 cigar_ref_len(const Cigar &c) {
  // ops that consume the reference: M D N = X (htslib bam_cigar_type & 2)
  u32 n = 0;
  for (u32 x : c) {
    const u32 op = x & 15u;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8)
      n += x >> 4;
  }
  return n;
}

// src/abismal.cpp:570-587
bool PairHit::offer(i16 scr, const Hit &s1, const Hit &s2) {
  const int have = r1.diffs + r2.diffs, got = s1.diffs + s2.diffs;
  if (scr > aln_score || (scr == aln_score && got < have)) {
    r1 = s1; r2 = s2; aln_score = scr;
    return true;
  }
  if (scr == aln_score && got == have)
    r1.flags |= kFlagAmbig;
  return false;
}

namespace {

constexpr u32 kSeCap = 50;           // src/abismal.cpp:448
constexpr u32 kPeCapSmall = 32;      // src/abismal.cpp:861
constexpr u32 kPeCapLarge = 32u << 10;
constexpr u32 kMaxBand = 61;         // 2*30+1, src/AbismalAlign.hpp:108,133
constexpr u32 kBufSlack = 64;

inline bool by_diffs(const Hit &a, const Hit &b) { return a.diffs < b.diffs; }

// An encoded query: one nibble per byte (hashing, DP) and 16 per u64 (filter).
// The byte buffer outlives each read, like the reference's reused `Read`
// vectors, so offsets a short read hashes past its end see the previous
// occupant's bytes (src/abismal.cpp:1377-1386).
struct Query {
  std::vector<u8> nib = std::vector<u8>(kPadding + kBufSlack, 0);
  u32 len = 0;
  std::vector<u64> packed;
  void encode(const std::string &s, bool a_alphabet) {
    const u32 n = static_cast<u32>(s.size());
    if (n > len)
      std::fill(nib.begin() + len, nib.begin() + n, 0);  // vector::resize growth
    for (u32 i = 0; i < n; ++i)
      nib[i] = read_nibble(s[i], a_alphabet);
    len = n;
    // src/abismal.cpp:1393-1426: tail nibbles of the last word are 0xF
    packed.assign((n + 15) / 16, 0);
    for (u32 i = 0; i < n; ++i)
      packed[i >> 4] |= static_cast<u64>(nib[i]) << ((i & 15) << 2);
    if (n & 15)
      for (u32 i = n; i < ((n + 15) & ~15u); ++i)
        packed[i >> 4] |= 0xFull << ((i & 15) << 2);
  }
};

// se_candidates: src/abismal.cpp:334-449
struct SeSet {
  Hit best;
  std::vector<Hit> v = std::vector<Hit>(kSeCap);
  u32 sz = 1;
  i16 cutoff = 0, good_cutoff = 0;
  bool sure_ambig = false;

  bool full() const { return sz == kSeCap; }
  void begin_read(u32 readlen) {
    best.clear(readlen);
    v[0].clear(readlen);
    cutoff = v[0].diffs;
    good_cutoff = static_cast<i16>(readlen / 10u);
    sure_ambig = false;
    sz = 1;
  }
  void wipe() {  // the no-argument reset(), src/abismal.cpp:406-415
    best.clear();
    v[0].clear();
    cutoff = v[0].diffs;
    sure_ambig = false;
    sz = 1;
  }
  void enter_specific() { cutoff = good_cutoff; }
  void enter_sensitive() { cutoff = v[0].diffs; }
  bool wants_sensitive() const { return !full() || cutoff > good_cutoff; }
  void admit(bool specific, i16 d, u16 flags, u32 pos) {
    if (d == 0) {  // exact matches never enter the heap
      const Hit h{0, flags, pos};
      if (best.empty())
        best = h;
      else if (!h.same_site(best))
        best.flags |= kFlagAmbig;
    }
    else {
      if (full()) {
        std::pop_heap(v.begin(), v.begin() + sz, by_diffs);
        v[sz - 1] = Hit{d, flags, pos};
      }
      else
        v[sz++] = Hit{d, flags, pos};
      std::push_heap(v.begin(), v.begin() + sz, by_diffs);
    }
    sure_ambig = best.ambig() && best.diffs == 0;
    cutoff = specific ? std::min(cutoff, v[0].diffs) : v[0].diffs;
  }
  void sort_unique() {  // prepare_for_alignments
    std::sort(v.begin(), v.begin() + sz, [](const Hit &a, const Hit &b) {
      return a.pos < b.pos || (a.pos == b.pos && a.flags < b.flags);
    });
    sz = static_cast<u32>(std::unique(v.begin(), v.begin() + sz,
                                      [](const Hit &a, const Hit &b) { return a.same_site(b); }) -
                          v.begin());
  }
};

// pe_candidates: src/abismal.cpp:775-863
struct PeSet {
  std::vector<Hit> v = std::vector<Hit>(kPeCapLarge);
  u32 sz = 1, capacity = kPeCapSmall;
  i16 cutoff = 0, good_cutoff = 0;
  bool sure_ambig = false;

  bool full() const { return sz == capacity; }
  void begin_read(u32 readlen) {
    v[0].clear(readlen);
    sure_ambig = false;
    cutoff = v[0].diffs;
    good_cutoff = static_cast<i16>(readlen / 10);
    sz = 1;
    capacity = kPeCapSmall;
  }
  void enter_specific() { cutoff = good_cutoff; }
  void enter_sensitive() { cutoff = v[0].diffs; }
  bool wants_sensitive() const { return capacity == kPeCapSmall || cutoff > good_cutoff; }
  bool worth_aligning() const { return sz != kPeCapLarge || cutoff != 0; }
  void admit(bool specific, i16 d, u16 flags, u32 pos) {
    if (full()) {
      if (specific && capacity != kPeCapLarge && d <= good_cutoff)
        ++capacity;  // grow instead of evicting while hits are good
      else {
        std::pop_heap(v.begin(), v.begin() + sz, by_diffs);
        --sz;
      }
    }
    v[sz++] = Hit{d, flags, pos};
    std::push_heap(v.begin(), v.begin() + sz, by_diffs);
    cutoff = specific ? std::min(cutoff, v[0].diffs) : v[0].diffs;
    sure_ambig = full() && cutoff == 0;
  }
  void sort_unique() {  // prepare_for_mating
    std::sort(v.begin(), v.begin() + sz, [](const Hit &a, const Hit &b) { return a.pos < b.pos; });
    sz = static_cast<u32>(std::unique(v.begin(), v.begin() + sz,
                                      [](const Hit &a, const Hit &b) { return a.same_site(b); }) -
                          v.begin());
  }
};

// Banded local alignment, +2 / -3 / -4 (src/AbismalAlign.hpp, all of it).
struct Aligner {
  const u64 *genome;
  std::vector<i16> tab;
  std::vector<signed char> arrow;
  u32 q_len = 0;
  Work *work = nullptr;
  enum : signed char { DIAG = 0, LEFT = 1, ABOVE = 2, NONE = -1 };  // M, I, D

  static u32 band_for(i16 diffs, i16 max_diffs) {
    const int v = 2 * std::min<int>(diffs, max_diffs) + 1;
    return v < 0 ? kMaxBand : std::min<u32>(kMaxBand, static_cast<u32>(v));
  }

  template <bool TB> i16 run(i16 diffs, i16 max_diffs, const u8 *q, u32 qn, u32 t_pos) {
    q_len = qn;
    if (diffs == 0)
      return static_cast<i16>(2 * qn);
    const u32 bw = band_for(diffs, max_diffs);
    const u32 rows = qn + bw;
    const std::size_t cells = static_cast<std::size_t>(rows) * bw;
    if (tab.size() < cells) { tab.resize(cells); arrow.resize(cells); }
    std::fill_n(tab.begin(), cells, 0);
    if (TB)
      std::fill_n(arrow.begin(), cells, NONE);
    if (work) { (TB ? work->aligns_tb : work->aligns)++; work->dp_cells += cells; }

    const u64 t0 = static_cast<u64>(t_pos) - (bw - 1) / 2;
    for (u32 i = 1; i < rows; ++i) {
      const u32 lo = i < bw ? bw - i : 0, hi = std::min(bw, rows - i);
      i16 *cur = &tab[static_cast<std::size_t>(i) * bw];
      const i16 *prev = cur - bw;
      signed char *ar = TB ? &arrow[static_cast<std::size_t>(i) * bw] : nullptr;
      const u8 t = gnib(genome, t0 + i - 1);
      const u8 *qq = q + (i > bw ? i - bw : 0);
      for (u32 j = lo; j < hi; ++j) {  // substitution from the row above
        const i16 s = static_cast<i16>(((qq[j - lo] & t) ? 2 : -3) + prev[j]);
        if (s > cur[j]) cur[j] = s;
        if (TB && cur[j] == s) ar[j] = DIAG;
      }
      for (u32 j = lo; j + 1 < hi; ++j) {  // gap in the read: skips a target base
        const i16 s = static_cast<i16>(prev[j + 1] - 4);
        if (s > cur[j]) cur[j] = s;
        if (TB && cur[j] == s) ar[j] = ABOVE;
      }
      for (u32 j = lo + 1; j < hi; ++j) {  // gap in the target: sequential chain
        const i16 s = static_cast<i16>(cur[j - 1] - 4);
        if (s > cur[j]) cur[j] = s;
        if (TB && cur[j] == s) ar[j] = LEFT;
      }
    }
    return *std::max_element(tab.begin(), tab.begin() + cells);
  }

  // src/AbismalAlign.hpp:388-440 (+ get_traceback :166-193)
  void cigar_from_last(i16 diffs, i16 max_diffs, Cigar &cig, u32 &aln_len, u32 &t_pos) const {
    const u32 bw = band_for(diffs, max_diffs);
    const std::size_t cells = static_cast<std::size_t>(q_len + bw) * bw;
    if (diffs == 0) {  // align() returned before touching the table; the reference reads stale cells
      cig.assign(1, q_len << 4);  // here and then ignores them (default CIGAR either way)
      aln_len = q_len;
      return;
    }
    const auto top = std::max_element(tab.begin(), tab.begin() + cells);  // first maximum
    if (*top == 0) {
      cig.assign(1, q_len << 4);
      aln_len = q_len;
      return;
    }
    std::size_t r = static_cast<std::size_t>(top - tab.begin()) / bw;
    std::size_t c = static_cast<std::size_t>(top - tab.begin()) % bw;
    const std::size_t clip_tail = (q_len + (bw - 1)) - (r + c);
    cig.clear();
    auto step = [&](signed char a) {
      if (a != LEFT) --r;
      if (a == LEFT) --c;
      if (a == ABOVE) ++c;
    };
    signed char run_op = arrow[r * bw + c];
    step(run_op);
    u32 run = 1;
    while (tab[r * bw + c] > 0) {
      const signed char a = arrow[r * bw + c];
      step(a);
      if (a != run_op) {
        cig.push_back((run << 4) | static_cast<u32>(static_cast<int>(run_op)));
        run = 0;
      }
      ++run;
      run_op = a;
    }
    cig.push_back((run << 4) | static_cast<u32>(static_cast<int>(run_op)));
    const std::size_t clip_head = (r + c) - (bw - 1);
    if (clip_head > 0)
      cig.push_back((static_cast<u32>(clip_head) << 4) | 4u);
    std::reverse(cig.begin(), cig.end());
    if (clip_tail > 0)
      cig.push_back((static_cast<u32>(clip_tail) << 4) | 4u);
    aln_len = static_cast<u32>(q_len - clip_tail - clip_head);
    t_pos = static_cast<u32>(t_pos - (bw - 1) / 2 + r);
  }
};

// src/AbismalAlign.hpp:73-89 with its integer types kept
i16 edit_distance(i16 scr, u32 len, const Cigar &cig) {
  if (scr == 0)
    return static_cast<i16>(len);
  int ins = 0, del = 0;
  for (u32 x : cig) {
    const u8 oplen = static_cast<u8>(x >> 4);  // the reference's oplen() returns uint8_t
    if ((x & 15u) == 1) ins = static_cast<i16>(ins + oplen);
    if ((x & 15u) == 2) del = static_cast<i16>(del + oplen);
  }
  const i16 A = static_cast<i16>(scr + 4 * (ins + del));
  const u32 num = 2u * (len - static_cast<u32>(ins)) - static_cast<u32>(static_cast<int>(A));
  const i16 mism = static_cast<i16>(num / 5u);
  return static_cast<i16>(mism + ins + del);
}

inline i16 max_diffs_for(u32 readlen, double frac) { return static_cast<i16>(frac * readlen); }
inline bool long_enough(u32 aln_len, u32 readlen, u32 kMinReadLen) {
  static const double min_frac = 1.0 - 0.4;
  return aln_len >= std::max(kMinReadLen, static_cast<u32>(min_frac * readlen));
}

}  // namespace

int probe_align(const u64 *genome, const u8 *q, u32 qlen, i16 diffs, i16 max_diffs, u32 t_pos, bool tb,
                Cigar *cig, u32 *aln_len, u32 *new_pos, int *nm) {
  Aligner a;
  a.genome = genome;
  if (!tb)
    return a.run<false>(diffs, max_diffs, q, qlen, t_pos);
  const i16 scr = a.run<true>(diffs, max_diffs, q, qlen, t_pos);
  u32 len = 0, pos = t_pos;
  a.cigar_from_last(diffs, max_diffs, *cig, len, pos);
  *aln_len = len;
  *new_pos = pos;
  *nm = edit_distance(scr, len, *cig);
  return scr;
}

struct Mapper::Impl {
  const Index &ix;
  MapParams par;
  Work &work;
  Aligner aln;
  // encodings keyed [end][rc][alphabet] (alphabet 1 = A-rich letters)
  Query enc[2][2][2];
  SeSet se[2];
  PeSet pe[2];
  std::vector<i16> memo = std::vector<i16>(kPeCapLarge);

  Impl(const Index &i, const MapParams &p, Work &w) : ix(i), par(p), work(w) {
    aln.genome = ix.genome.data();
    aln.work = &w;
  }

  // src/abismal.cpp:1105-1122
  i16 hamming(i16 cutoff, const std::vector<u64> &pk, u32 pos) {
    const u64 *g = ix.genome.data() + (pos >> 4);
    const u32 sh = (pos & 15u) << 2;
    i16 d = 0;
    for (std::size_t w = 0; d <= cutoff && w < pk.size(); ++w) {
      const u64 window = (g[w] >> sh) | ((g[w + 1] << (63 - sh)) << 1);
      d = static_cast<i16>(d + 16 - __builtin_popcountll(pk[w] & window));
      ++work.words;
    }
    return d;
  }

  // src/abismal.cpp:1124-1150
  template <class Set>
  void scan_bucket(Set &S, const Query &q, u16 flags, u32 offset, const u32 *lo, const u32 *hi) {
    for (; lo != hi && !S.sure_ambig; ++lo) {
      const u32 pos = *lo - offset;
      ++work.candidates;
      const i16 d = hamming(S.cutoff, q.packed, pos);
      if (d <= S.cutoff) {
        S.admit(true, d, flags, pos);
        ++work.set_updates;
      }
    }
  }

  // std::lower_bound's halving, spelled out (positions need not be sorted
  // beyond kSortDepth letters, so the exact probe sequence matters)
  template <class Pred> const u32 *first_not(const u32 *lo, const u32 *hi, Pred below) {
    std::ptrdiff_t n = hi - lo;
    while (n > 0) {
      const std::ptrdiff_t half = n >> 1;
      ++work.search_probes;
      if (below(lo[half])) { lo += half + 1; n -= half + 1; }
      else n = half;
    }
    return lo;
  }

  // src/abismal.cpp:1163-1194: follow the read's 2-letter symbols past the
  // hashed prefix while the bucket is too big
  u32 narrow2(const u8 *q, u32 limit, const u32 *&lo, const u32 *&hi) {
    const u64 *g = ix.genome.data();
    u32 p = kKeyWeight;
    const u32 *plo = lo, *phi = hi;
    for (; p != limit && (hi - lo) > static_cast<std::ptrdiff_t>(par.max_candidates); ++p) {
      plo = lo; phi = hi;
      const u32 *ones = first_not(lo, hi, [&](u32 gp) { return bit2(gnib(g, static_cast<u64>(gp) + p)) < 1u; });
      if (bit2(q[p])) lo = ones; else hi = ones;
    }
    if (lo == hi) { --p; lo = plo; hi = phi; }
    return p;
  }

  // src/abismal.cpp:1214-1259
  u32 narrow3(Conv cv, const u8 *q, u32 limit, const u32 *&lo, const u32 *&hi) {
    const u64 *g = ix.genome.data();
    const u32 mid_sym = cv == C_TO_T ? 1u : 2u, top_sym = cv == C_TO_T ? 4u : 8u;
    u32 p = kKeyWeight3;
    const u32 *plo = lo, *phi = hi;
    for (; p != limit && (hi - lo) > static_cast<std::ptrdiff_t>(par.max_candidates); ++p) {
      plo = lo; phi = hi;
      const u32 *b1 = first_not(lo, hi, [&](u32 gp) { return sortsym3(gnib(g, static_cast<u64>(gp) + p), cv) < mid_sym; });
      const u32 *b2 = first_not(lo, hi, [&](u32 gp) { return sortsym3(gnib(g, static_cast<u64>(gp) + p), cv) < top_sym; });
      const u32 sym = sortsym3(q[p], cv);
      if (sym == 0) hi = b1;
      else if (sym == mid_sym) { lo = b1; hi = b2; }
      else lo = b2;
    }
    if (lo == hi) { --p; lo = plo; hi = phi; }
    return p;
  }

  // src/abismal.cpp:1269-1375
  template <class Set> void seed_passes(Set &S, const Query &q, bool rc, bool a_rich) {
    const u16 flags = static_cast<u16>((rc ? kFlagRC : 0) | (a_rich ? kFlagARich : 0));
    const Conv cv = (rc != a_rich) ? G_TO_A : C_TO_T;
    const u32 *cnt2 = ix.counter.data(), *idx2 = ix.index.data();
    const u32 *cnt3 = cv == C_TO_T ? ix.counter_t.data() : ix.counter_a.data();
    const u32 *idx3 = cv == C_TO_T ? ix.index_t.data() : ix.index_a.data();
    const u8 *nb = q.nib.data();
    const u32 L = q.len, maxc = par.max_candidates;

    auto keys_at_0 = [&](u32 &k2, u32 &k3) {
      k2 = k3 = 0;
      for (u32 j = 0; j < kKeyWeight; ++j) k2 = (k2 << 1) | bit2(nb[j]);
      for (u32 j = 0; j < kKeyWeight3; ++j) roll3(nb[j], cv, k3);
    };
    u32 k2, k3;
    keys_at_0(k2, k3);

    const u32 spec_len = std::min(L - ix.window, L >> 1);
    const u32 spec_lim = std::max(ix.window, L >> 1);
    S.enter_specific();
    for (u32 i = 0; i < spec_lim && !S.sure_ambig; ++i) {
      ++work.seed_iters;
      const u32 *lo2 = idx2 + cnt2[k2], *hi2 = idx2 + cnt2[k2 + 1];
      const u32 len2 = narrow2(nb + i, L - i, lo2, hi2);
      const u32 *lo3 = idx3 + cnt3[k3], *hi3 = idx3 + cnt3[k3 + 1];
      const u32 len3 = narrow3(cv, nb + i, L - i, lo3, hi3);
      if (static_cast<u32>(hi2 - lo2) <= maxc || len2 >= spec_len)
        scan_bucket(S, q, flags, i, lo2, hi2);
      if (static_cast<u32>(hi3 - lo3) <= maxc || len3 >= spec_len)
        scan_bucket(S, q, flags, i, lo3, hi3);
      roll2(nb[i + kKeyWeight], k2);
      roll3(nb[i + kKeyWeight3], cv, k3);
    }
    if (!S.wants_sensitive())
      return;

    keys_at_0(k2, k3);
    S.enter_sensitive();
    const u32 n_off = L - kKeyWeight + 1;
    for (u32 i = 0; i < n_off && !S.sure_ambig; ++i) {
      ++work.seed_iters;
      const u32 *lo2 = idx2 + cnt2[k2], *hi2 = idx2 + cnt2[k2 + 1];
      const u32 *lo3 = idx3 + cnt3[k3], *hi3 = idx3 + cnt3[k3 + 1];
      const u32 d2 = static_cast<u32>(hi2 - lo2), d3 = static_cast<u32>(hi3 - lo3);
      if (d2 != 0 && d2 <= maxc && (d3 == 0 || d2 <= 10 * d3))
        scan_bucket(S, q, flags, i, lo2, hi2);
      if (d3 != 0 && d3 <= maxc)
        scan_bucket(S, q, flags, i, lo3, hi3);
      roll2(nb[i + kKeyWeight], k2);
      roll3(nb[i + kKeyWeight3], cv, k3);
    }
  }

  const Query &query_for(int end, const Hit &h) const {
    const bool rc = h.rc();
    return enc[end][rc][rc != h.a_rich()];
  }

  // src/abismal.cpp:1435-1497
  void choose_se(int end, u32 readlen, double frac, SeSet &S, Hit &best, Cigar &cig) {
    const i16 L = static_cast<i16>(readlen);
    const i16 md = max_diffs_for(static_cast<u32>(L), frac);
    const i16 perfect = static_cast<i16>(2 * readlen);
    if (!S.best.empty()) {
      best = S.best;
      cig.assign(1, static_cast<u32>(L) << 4);
      return;
    }
    i16 top = 0;
    u32 top_pos = 0;
    S.sort_unique();
    u32 k = 0;
    while (k < S.sz && S.v[k].empty()) ++k;
    for (; k < S.sz; ++k) {
      const Hit &h = S.v[k];
      if (!(h.diffs < static_cast<i16>(0.4 * L)))
        continue;
      const Query &q = query_for(end, h);
      const i16 s = aln.run<false>(h.diffs, md, q.nib.data(), q.len, h.pos);
      if (s > top) { best = h; top = s; top_pos = h.pos; }
      else if (s == top) {
        const u32 gap = h.pos > top_pos ? h.pos - top_pos : top_pos - h.pos;
        if (s == perfect ? h.pos != top_pos : gap > 3)
          best.flags |= kFlagAmbig;
      }
    }
    if (best.pos == 0) { best.clear(); return; }
    const Query &q = query_for(end, best);
    aln.run<true>(best.diffs, md, q.nib.data(), q.len, best.pos);
    u32 alen = 0;
    aln.cigar_from_last(best.diffs, md, cig, alen, best.pos);
    best.diffs = edit_distance(top, alen, cig);
    if (!(long_enough(alen, static_cast<u32>(L), ix.min_read_len()) && best.diffs <= max_diffs_for(static_cast<u32>(L), frac)))
      best.clear();
  }

  // src/abismal.cpp:1722-1831.  A/B are the sets of the two ends in the order
  // this orientation call received them; `swapped` says A is read 2.
  void mate(bool swapped, const PeSet &A, const PeSet &B, const Query &qa, const Query &qb,
            Cigar &ciga, Cigar &cigb, PairHit &best) {
    const i16 mda = max_diffs_for(qa.len, par.valid_frac), mdb = max_diffs_for(qb.len, par.valid_frac);
    std::fill_n(memo.begin(), A.sz, 0);
    std::ptrdiff_t ia = 0, ib = 0;
    const std::ptrdiff_t na = A.sz, nb = B.sz;
    while (ia != na && A.v[ia].empty()) ++ia;
    while (ib != nb && B.v[ib].empty()) ++ib;
    i16 last_sa = 0, keep_sa = 0, keep_sb = 0;
    u32 keep_pa = 0, keep_pb = 0;
    for (; ib != nb && !best.sure_ambig(); ++ib) {
      const Hit hb = B.v[ib];
      i16 sb = 0;
      const u32 frag_end = hb.pos + qb.len;
      while (ia == na || (ia != 0 && A.v[ia].pos + par.max_frag >= frag_end)) --ia;
      while (ia != na && A.v[ia].pos + par.max_frag < frag_end) ++ia;
      for (; ia != na && A.v[ia].pos + par.min_frag <= frag_end && !best.sure_ambig(); ++ia) {
        const Hit ha = A.v[ia];
        if (sb == 0)
          sb = aln.run<false>(hb.diffs, mdb, qb.nib.data(), qb.len, hb.pos);
        if (memo[ia] == 0) {
          last_sa = aln.run<false>(ha.diffs, mda, qa.nib.data(), qa.len, ha.pos);
          memo[ia] = last_sa;
        }
        const i16 pair = static_cast<i16>(sb + memo[ia]);
        if (swapped ? best.offer(pair, hb, ha) : best.offer(pair, ha, hb)) {
          keep_sa = last_sa;  // sic: the last *computed* score, not memo[ia]
          keep_sb = sb;
          keep_pa = ha.pos;
          keep_pb = hb.pos;
        }
      }
    }
    if (keep_pa == 0)
      return;
    Hit ha = swapped ? best.r2 : best.r1, hb = swapped ? best.r1 : best.r2;
    u32 la = 0, lb = 0;
    aln.run<true>(ha.diffs, mda, qa.nib.data(), qa.len, keep_pa);
    aln.cigar_from_last(ha.diffs, mda, ciga, la, keep_pa);
    ha.pos = keep_pa;
    ha.diffs = edit_distance(keep_sa, la, ciga);
    aln.run<true>(hb.diffs, mdb, qb.nib.data(), qb.len, keep_pb);
    aln.cigar_from_last(hb.diffs, mdb, cigb, lb, keep_pb);
    hb.pos = keep_pb;
    hb.diffs = edit_distance(keep_sb, lb, cigb);
    const u32 fe = keep_pb + lb;
    if (fe >= keep_pa + par.min_frag && fe <= keep_pa + par.max_frag) {
      best.r1 = swapped ? hb : ha;
      best.r2 = swapped ? ha : hb;
    }
    else
      best.clear();
  }

  // map_fragments + select_maps + best_single: src/abismal.cpp:1715-1720,1833-1885.
  // endA is mapped forward with (rc=0, a_rich=ar); endB reverse-complemented
  // with (rc=1, a_rich=!ar).  Both use the alphabet `ar`.
  bool orientation(int endA, const std::string &ra, const std::string &rb, bool ar, Cigar &ciga,
                   Cigar &cigb, PairHit &best) {
    const int endB = 1 - endA;
    PeSet &A = pe[endA], &B = pe[endB];
    A.begin_read(static_cast<u32>(ra.size()));
    B.begin_read(static_cast<u32>(rb.size()));
    if (ra.empty() && rb.empty())
      return false;
    Query &qa = enc[endA][0][ar], &qb = enc[endB][1][ar];
    if (!ra.empty()) {
      qa.encode(ra, ar);
      seed_passes(A, qa, false, ar);
    }
    if (!rb.empty()) {
      qb.encode(revcomp_read(rb), ar);
      seed_passes(B, qb, true, !ar);
    }
    if (A.worth_aligning() && B.worth_aligning()) {
      A.sort_unique();
      B.sort_unique();
      mate(endA == 1, A, B, qa, qb, ciga, cigb, best);
    }
    for (int e : {endA, endB}) {
      const PeSet &P = pe[e];
      SeSet &S = se[e];
      for (u32 k = 0; k < P.sz && !S.sure_ambig; ++k) {
        S.admit(false, P.v[k].diffs, P.v[k].flags, P.v[k].pos);
        ++work.set_updates;
      }
    }
    return true;
  }
};

Mapper::Mapper(const Index &ix, const MapParams &p) : m(new Impl(ix, p, work)) {}
Mapper::~Mapper() { delete m; }

void Mapper::map_se(const std::string &read, SeMode mode, Hit &best, Cigar &cig) {
  SeSet &S = m->se[0];
  S.begin_read(static_cast<u32>(read.size()));
  best.clear();
  ++work.reads;
  if (read.empty())
    return;
  const std::string rc = revcomp_read(read);
  // (rc, a_rich) calls in the reference's order; alphabet = rc xor a_rich
  static const bool kCalls[3][4][2] = {
    {{0, 0}, {1, 0}, {0, 0}, {0, 0}},   // T-rich: src/abismal.cpp:1556-1572
    {{0, 1}, {1, 1}, {0, 0}, {0, 0}},   // A-rich / PBAT single-end
    {{0, 0}, {0, 1}, {1, 1}, {1, 0}}};  // random PBAT: src/abismal.cpp:1649-1676
  const int n_calls = mode == SE_RANDOM ? 4 : 2;
  for (int c = 0; c < n_calls; ++c) {
    const bool r = kCalls[mode][c][0], ar = kCalls[mode][c][1];
    Query &q = m->enc[0][r][r != ar];
    q.encode(r ? rc : read, r != ar);
    m->seed_passes(S, q, r, ar);
  }
  m->choose_se(0, static_cast<u32>(read.size()), m->par.valid_frac, S, best, cig);
}

void Mapper::touch_se(const std::string &read, SeMode mode) {
  if (read.empty())
    return;
  const std::string rc = revcomp_read(read);
  for (int r = 0; r < 2; ++r)
    for (int alpha = 0; alpha < 2; ++alpha) {
      const bool ar = (r != 0) != (alpha != 0);  // alphabet = rc xor a_rich
      const bool used = mode == SE_RANDOM || (mode == SE_A_RICH) == ar;
      if (used) m->enc[0][r][alpha].encode(r ? rc : read, alpha != 0);
    }
}

void Mapper::touch_pe(const std::string &r1, const std::string &r2, PeMode mode) {
  if (r1.empty() && r2.empty())
    return;
  const std::string *rd[2] = {&r1, &r2};
  for (int end = 0; end < 2; ++end) {
    if (rd[end]->empty())
      continue;
    const std::string rc = revcomp_read(*rd[end]);
    for (int r = 0; r < 2; ++r)
      for (int alpha = 0; alpha < 2; ++alpha) {
        // orientation(endA, ar): endA forward and endB reverse-complemented, both in alphabet ar;
        // normal = {(0, T), (1, A)}, PBAT = {(0, A), (1, T)}, random = all four
        const bool as_a = r == 0;  // this end is endA in the call that uses [end][r][alpha]
        const int end_a = as_a ? end : 1 - end;
        const bool used = mode == PE_RANDOM || ((mode == PE_PBAT) != (end_a == 1)) == (alpha != 0);
        if (used) m->enc[end][r][alpha].encode(r ? rc : *rd[end], alpha != 0);
      }
  }
}

void Mapper::map_pe(const std::string &r1, const std::string &r2, PeMode mode, PairHit &best,
                    Hit &se1, Hit &se2, Cigar &cig1, Cigar &cig2) {
  const u32 l1 = static_cast<u32>(r1.size()), l2 = static_cast<u32>(r2.size());
  m->pe[0].begin_read(l1);
  m->pe[1].begin_read(l2);
  m->se[0].begin_read(l1);
  m->se[1].begin_read(l2);
  best.clear(l1, l2);
  se1.clear(l1);
  se2.clear(l2);
  work.reads += 2;

  bool any = false;
  // orientation(endA, readA, readB, alphabet): src/abismal.cpp:1963-1979, :2106-2133
  if (mode == PE_RANDOM) {
    any |= m->orientation(0, r1, r2, false, cig1, cig2, best);
    any |= m->orientation(1, r2, r1, true, cig2, cig1, best);
    any |= m->orientation(0, r1, r2, true, cig1, cig2, best);
    any |= m->orientation(1, r2, r1, false, cig2, cig1, best);
  }
  else {
    const bool ar = mode == PE_PBAT;
    any |= m->orientation(0, r1, r2, ar, cig1, cig2, best);
    any |= m->orientation(1, r2, r1, !ar, cig2, cig1, best);
  }
  if (!any) {
    best.clear();
    m->se[0].wipe();
    m->se[1].wipe();
  }
  // valid_pair: src/abismal.cpp:624-631
  {
    const u32 a1 = cigar_ref_len(cig1), a2 = cigar_ref_len(cig2);
    const bool ok = long_enough(a1, l1, m->ix.min_read_len()) && long_enough(a2, l2, m->ix.min_read_len()) &&
                    best.diffs() <= static_cast<i16>(m->par.valid_frac * (a1 + a2));
    if (!ok)
      best.clear();
  }
  if (!best.should_report(m->par.allow_ambig)) {
    m->choose_se(0, l1, m->par.valid_frac / 2, m->se[0], se1, cig1);
    m->choose_se(1, l2, m->par.valid_frac / 2, m->se[1], se2, cig2);
  }
}

}  // namespace abo
