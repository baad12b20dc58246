// abismal_amd HIP kernels for gfx950: read packing and the single-end mapping
// kernel (seed probe -> bucket narrowing -> Hamming filter -> ordered replay
// into the candidate set -> banded alignment -> CIGAR).  One wavefront maps one
// read; see DESIGN.md for the data layout and the reasoning.
#include "abm_kernels.hpp"

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
// Wave-resident single-end candidate set (se_candidates, src/abismal.cpp:334-449).
// Heap slot k lives in lane k (k < 50); everything else is wave-uniform.
// =============================================================================
struct SeSet {
  int hd;   // per lane: diffs of slot
  u32 hf;   // per lane: flags of slot
  u32 hp;   // per lane: pos of slot
  int sz, cutoff, good_cutoff;
  int best_d;
  u32 best_f, best_p;
  bool sure_ambig;

  __device__ __forceinline__ void begin_read(u32 readlen) {
    const int worst = static_cast<i16>(0.4 * readlen);  // se_element::reset(readlen), :292-296
    hd = worst; hf = 0; hp = 0;
    sz = 1;
    cutoff = worst;
    good_cutoff = static_cast<i16>(readlen / 10u);
    best_d = worst; best_f = 0; best_p = 0;
    sure_ambig = false;
  }
  __device__ __forceinline__ void move_slot(int dst, int src) {
    const int d = rdlane(hd, src);
    const u32 f = rdlane(hf, src), p = rdlane(hp, src);
    wrlane(hd, dst, d); wrlane(hf, dst, f); wrlane(hp, dst, p);
  }
  __device__ __forceinline__ void put_slot(int dst, int d, u32 f, u32 p) {
    wrlane(hd, dst, d); wrlane(hf, dst, f); wrlane(hp, dst, p);
  }
  // libstdc++ __push_heap with value (d,f,p) entering at `hole`, comparator diffs<
  __device__ __forceinline__ void sift_up(int hole, int d, u32 f, u32 p) {
    int parent = (hole - 1) / 2;
    while (hole > 0 && rdlane(hd, parent) < d) {
      move_slot(hole, parent);
      hole = parent;
      parent = (hole - 1) / 2;
    }
    put_slot(hole, d, f, p);
  }
  // libstdc++ pop_heap on [0,n) followed by overwriting slot n-1 and push_heap:
  // only the __adjust_heap of the displaced last element matters here
  __device__ __forceinline__ void pop_max(int n) {
    const int len = n - 1;
    const int vd = rdlane(hd, len);
    const u32 vf = rdlane(hf, len), vp = rdlane(hp, len);
    int hole = 0, second = 0;
    while (second < (len - 1) / 2) {
      second = 2 * (second + 1);
      if (rdlane(hd, second) < rdlane(hd, second - 1))
        --second;
      move_slot(hole, second);
      hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
      second = 2 * (second + 1);
      move_slot(hole, second - 1);
      hole = second - 1;
    }
    sift_up(hole, vd, vf, vp);
  }
  // se_candidates::update, :394-404
  __device__ __forceinline__ void admit(bool specific, int d, u32 f, u32 p) {
    if (d == 0) {
      if (best_p == 0) { best_d = 0; best_f = f; best_p = p; }
      else if (p != best_p || f != best_f) best_f |= kFlagAmbig;
    }
    else {
      if (sz == static_cast<int>(kSeCap)) pop_max(sz);
      else ++sz;
      sift_up(sz - 1, d, f, p);
    }
    sure_ambig = (best_f & kFlagAmbig) && best_d == 0;
    const int top = rdlane(hd, 0);
    cutoff = specific ? min(cutoff, top) : top;
  }
};

// Per-wave LDS carve-up
struct WaveLds {
  u64 *qpk;    // [4][W] packed encodings
  u64 *qbits;  // [4][WB] 2-letter bit strings, bit j = bit2(nibble j), 1 past the end
  u16 *mark;   // [64]
  u32 *ctmp;   // [cig_stride] reversed CIGAR scratch
  u8 *tb;      // traceback bytes
  u32 W, WB;
};

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

// std::lower_bound's exact probe sequence (see oracle first_not)
template <class Below>
__device__ __forceinline__ u32 first_not(u32 lo, u32 hi, u32 &probes, Below below) {
  int n = static_cast<int>(hi - lo);
  while (n > 0) {
    const int half = n >> 1;
    ++probes;
    if (below(lo + half)) { lo += half + 1; n -= half + 1; }
    else n = half;
  }
  return lo;
}

// find_candidates, src/abismal.cpp:1163-1194 (range as indices into tbl[])
__device__ __forceinline__ u32 narrow2(const u64 *__restrict__ genome, const u32 *__restrict__ tbl,
                                       const u64 *qpk, u32 qbase, u32 limit, u32 maxc, u32 &lo,
                                       u32 &hi, u32 &probes) {
  u32 p = kKeyWeight, plo = lo, phi = hi;
  for (; p != limit && (hi - lo) > maxc; ++p) {
    plo = lo; phi = hi;
    const u32 ones = first_not(lo, hi, probes, [&](u32 k) {
      return bit2(gnib(genome, static_cast<u64>(tbl[k]) + p)) < 1u;
    });
    if (bit2(q_nibble(qpk, qbase + p))) lo = ones; else hi = ones;
  }
  if (lo == hi) { --p; lo = plo; hi = phi; }
  return p;
}

// find_candidates_three, src/abismal.cpp:1214-1259
__device__ __forceinline__ u32 narrow3(const u64 *__restrict__ genome, const u32 *__restrict__ tbl,
                                       bool g_to_a, const u64 *qpk, u32 qbase, u32 limit, u32 maxc,
                                       u32 &lo, u32 &hi, u32 &probes) {
  const u32 mid_sym = g_to_a ? 2u : 1u, top_sym = g_to_a ? 8u : 4u;
  u32 p = kKeyWeight3, plo = lo, phi = hi;
  for (; p != limit && (hi - lo) > maxc; ++p) {
    plo = lo; phi = hi;
    const u32 b1 = first_not(lo, hi, probes, [&](u32 k) {
      return sortsym3(gnib(genome, static_cast<u64>(tbl[k]) + p), g_to_a) < mid_sym;
    });
    const u32 b2 = first_not(lo, hi, probes, [&](u32 k) {
      return sortsym3(gnib(genome, static_cast<u64>(tbl[k]) + p), g_to_a) < top_sym;
    });
    const u32 sym = sortsym3(q_nibble(qpk, qbase + p), g_to_a);
    if (sym == 0) hi = b1;
    else if (sym == mid_sym) { lo = b1; hi = b2; }
    else lo = b2;
  }
  if (lo == hi) { --p; lo = plo; hi = phi; }
  return p;
}

// full_compare without the early exit (src/abismal.cpp:1105-1122): the reference
// stops once d exceeds the cutoff, which changes d only when the hit is rejected
// anyway, so the complete distance gives identical admit/reject decisions.
__device__ __forceinline__ int hamming(const u64 *__restrict__ genome, const u64 *qpk, u32 nwords,
                                       u32 pos) {
  const u64 *g = genome + (pos >> 4);
  const u32 sh = (pos & 15u) << 2;
  int d = 0;
  u64 g0 = g[0];
  for (u32 w = 0; w < nwords; ++w) {
    const u64 g1 = g[w + 1];
    const u64 win = (g0 >> sh) | ((g1 << (63 - sh)) << 1);
    d += 16 - __popcll(qpk[w] & win);
    g0 = g1;
  }
  return static_cast<i16>(d);
}

// One (strand, alphabet) call of process_seeds (src/abismal.cpp:1269-1375) for
// the whole wave.  Lanes are seed offsets while probing/narrowing, then become
// candidates (all checked buckets of 64 offsets flattened in reference order)
// for the Hamming filter; survivors are replayed in order into the set.
struct WorkTally { u32 seed_iters, probes, cands, words, updates; };

template <bool SPECIFIC>
__device__ __forceinline__ void seed_pass(const DevIndex &ix, const WaveLds &lds, u32 enc, bool g_to_a,
                                          u32 flags, u32 L, SeSet &S, WorkTally &wt) {
  const int lane = lane_id();
  const u64 *qpk = lds.qpk + enc * lds.W;
  const u64 *qb = lds.qbits + enc * lds.WB;
  const u32 *cnt3 = g_to_a ? ix.counter_a : ix.counter_t;
  const u32 *idx3 = g_to_a ? ix.index_a : ix.index_t;
  const u32 maxc = ix.max_candidates;
  const u32 nwords = (L + 15) >> 4;
  const u32 spec_len = min(L - kWindow, L >> 1);
  const u32 n_off = SPECIFIC ? max(kWindow, L >> 1) : L - kKeyWeight + 1;

  for (u32 g0 = 0; g0 < n_off && !S.sure_ambig; g0 += 64) {
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
      lo2 = ix.counter[k2]; hi2 = ix.counter[k2 + 1];
      lo3 = cnt3[k3];       hi3 = cnt3[k3 + 1];
      if (SPECIFIC) {
        u32 probes = 0;
        const u32 len2 = narrow2(ix.genome, ix.index, qpk, i, L - i, maxc, lo2, hi2, probes);
        const u32 len3 = narrow3(ix.genome, idx3, g_to_a, qpk, i, L - i, maxc, lo3, hi3, probes);
        chk2 = (hi2 - lo2) <= maxc || len2 >= spec_len;
        chk3 = (hi3 - lo3) <= maxc || len3 >= spec_len;
        wt.probes += probes;
      }
      else {
        const u32 d2 = hi2 - lo2, d3 = hi3 - lo3;
        chk2 = d2 != 0 && d2 <= maxc && (d3 == 0 || d2 <= 10u * d3);
        chk3 = d3 != 0 && d3 <= maxc;
      }
      ++wt.seed_iters;
    }
    const u32 na = chk2 ? hi2 - lo2 : 0u, nb = chk3 ? hi3 - lo3 : 0u;
    u32 total;
    const u32 start_a = wave_excl_sum(na + nb, total), start_b = start_a + na;

    int carry = 0;
    for (u32 c0 = 0; c0 < total && !S.sure_ambig; c0 += 64) {
      // which (offset, table) segment does each of these 64 candidates belong to
      lds.mark[lane] = 0;
      __syncthreads();
      if (na && start_a - c0 < 64u) lds.mark[start_a - c0] = static_cast<u16>(2 * lane + 1);
      if (nb && start_b - c0 < 64u) lds.mark[start_b - c0] = static_cast<u16>(2 * lane + 2);
      __syncthreads();
      const int m = wave_incl_max(static_cast<int>(lds.mark[lane]));
      const int seg = m ? m - 1 : carry;
      carry = rdlane(seg, 63);
      const int owner = seg >> 1;
      const bool three = seg & 1;
      const u32 sa = __shfl(start_a, owner), sb = __shfl(start_b, owner);
      const u32 ba = __shfl(lo2, owner), bb = __shfl(lo3, owner);
      const u32 c = c0 + lane;
      const bool valid = c < total;
      u32 pos = 0;
      int h = 0x7fff;
      if (valid) {
        const u32 entry = three ? idx3[bb + (c - sb)] : ix.index[ba + (c - sa)];
        pos = entry - (g0 + static_cast<u32>(owner));
        h = hamming(ix.genome, qpk, nwords, pos);
        ++wt.cands;
        wt.words += nwords;
      }
      // ordered replay (check_hits + se_candidates::update, :1133-1149, :394-404)
      u64 todo = __ballot(valid && h <= S.cutoff);
      while (todo && !S.sure_ambig) {
        const int l = __builtin_ctzll(todo);
        const int before = S.cutoff;
        S.admit(true, rdlane(h, l), flags, rdlane(pos, l));
        ++wt.updates;
        todo &= ~(((1ull << l) << 1) - 1);
        if (S.cutoff < before) todo &= __ballot(valid && h <= S.cutoff);
      }
    }
  }
}

// =============================================================================
// Banded local alignment on a wave (AbismalAlign::align, src/AbismalAlign.hpp:320-386).
// Lane j is band column j; rows advance serially; the in-row insertion chain
// (from_left) is a max-plus prefix scan.  With TB the arrows go to LDS and the
// first maximum in row-major order is returned in (best_r, best_c).
// =============================================================================
__device__ __forceinline__ int band_for(int diffs, int max_diffs) {
  const int v = 2 * min(diffs, max_diffs) + 1;
  return v < 0 ? static_cast<int>(kMaxBand) : min(static_cast<int>(kMaxBand), v);
}

template <bool TB>
__device__ __forceinline__ int wave_align(const u64 *__restrict__ genome, const u64 *qpk, u32 W, int L,
                                          int diffs, int max_diffs, u32 t_pos, u8 *tb, int &best_r,
                                          int &best_c) {
  const int lane = lane_id();
  best_r = best_c = 0;
  if (diffs == 0)
    return static_cast<i16>(2 * L);
  const int bw = band_for(diffs, max_diffs);
  const int rows = L + bw;
  const u64 t0 = static_cast<u64>(t_pos) - static_cast<u64>((bw - 1) / 2);
  const u64 gw0 = t0 >> 4;
  const int ngw = static_cast<int>(((t0 + rows - 2) >> 4) - gw0) + 1;  // <= 64 for L <= kMaxReadLen
  const u64 gw = lane < ngw ? genome[gw0 + lane] : 0ull;
  const u64 qw = lane < static_cast<int>(W) ? qpk[lane] : 0ull;
  constexpr int NEG = -(1 << 20);

  if (TB && lane < bw) tb[lane] = 3;  // row 0: no arrow, score 0
  int prev = 0, qv = 0, bestv = 0, bestrow = 0;
  for (int i = 1; i < rows; ++i) {
    const u64 tk = t0 + static_cast<u64>(i - 1);
    const int t = static_cast<int>(rdlane(gw, static_cast<int>((tk >> 4) - gw0)) >> ((tk & 15u) << 2)) & 15;
    const int qi = i - 1;  // read base entering the band at its last column
    const int qn = qi < L ? static_cast<int>(rdlane(qw, qi >> 4) >> ((qi & 15) << 2)) & 15 : 0;
    qv = __shfl_down(qv, 1);
    if (lane == bw - 1) qv = qn;
    const int left = i < bw ? bw - i : 0, right = min(bw, rows - i);
    const bool valid = lane >= left && lane < right;
    const int up = __shfl_down(prev, 1);
    const int sdiag = prev + ((qv & t) ? 2 : -3);
    int c = max(sdiag, 0);
    int arrow = (c == sdiag) ? 0 : 3;
    const int sabove = up - 4;
    if (lane + 1 < right) {
      c = max(c, sabove);
      if (c == sabove) arrow = 2;
    }
    // from_left: cur[j] = max(cur[j], cur[j-1]-4) sequentially == prefix max of c[k]+4k
    int u = valid ? c + 4 * lane : NEG;
    u = wave_incl_max(u);
    const int f = u - 4 * lane;
    const int cur = valid ? f : 0;
    if (TB) {
      const int lf = __shfl_up(cur, 1) - 4;
      if (lane > left && cur == lf) arrow = 1;
      if (lane < bw) tb[i * bw + lane] = static_cast<u8>(valid ? (arrow | (cur > 0 ? 4 : 0)) : 3);
    }
    if (cur > bestv) { bestv = cur; bestrow = i; }
    prev = cur;
  }
  // first maximum in row-major order: max value, then smallest row, then smallest column
  const u64 key = (static_cast<u64>(static_cast<u32>(bestv)) << 32) |
                  (static_cast<u64>(0xFFFFu - static_cast<u32>(bestrow)) << 8) |
                  static_cast<u64>(0xFFu - static_cast<u32>(lane));
  const u64 top = wave_max_u64(lane < bw ? key : 0ull);
  best_r = static_cast<int>(0xFFFFu - static_cast<u32>((top >> 8) & 0xFFFFu));
  best_c = static_cast<int>(0xFFu - static_cast<u32>(top & 0xFFu));
  return static_cast<i16>(static_cast<int>(top >> 32));
}

// build_cigar_len_and_pos + get_traceback (src/AbismalAlign.hpp:166-193, :388-440).
// Runs uniformly on the wave; ops are collected reversed in LDS then emitted.
__device__ __forceinline__ void wave_cigar(const u8 *tb, u32 *ctmp, int L, int diffs, int max_diffs,
                                           int score, int best_r, int best_c, u32 *cig_out,
                                           u32 cig_stride, u32 &n_ops, u32 &n_body, u32 &aln_len,
                                           u32 &t_pos, bool &overflow) {
  const int lane = lane_id();
  n_body = 0;  // I/D/M runs left in ctmp[] (reversed), for the NM computation
  if (score == 0 || diffs == 0) {
    if (lane == 0) cig_out[0] = static_cast<u32>(L) << 4;
    n_ops = 1;
    aln_len = static_cast<u32>(L);
    return;
  }
  const int bw = band_for(diffs, max_diffs);
  int r = best_r, c = best_c;
  const int clip_tail = (L + (bw - 1)) - (r + c);
  u32 n = 0;
  auto emit = [&](u32 run, int op) {
    if (n < cig_stride) { if (lane == 0) ctmp[n] = (run << 4) | static_cast<u32>(op); }
    else overflow = true;
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
  __syncthreads();
  // final order: [head clip] reversed(ops) [tail clip]
  const u32 body = min(n, cig_stride);
  u32 total = body + (clip_head > 0) + (clip_tail > 0);
  if (n > cig_stride || total > cig_stride) { overflow = true; total = min(total, cig_stride); }
  for (u32 k = lane; k < total; k += 64) {
    u32 v;
    const u32 kk = k - (clip_head > 0 ? 1u : 0u);
    if (clip_head > 0 && k == 0) v = (static_cast<u32>(clip_head) << 4) | 4u;
    else if (kk < body) v = ctmp[body - 1 - kk];
    else v = (static_cast<u32>(clip_tail) << 4) | 4u;
    cig_out[k] = v;
  }
  n_ops = total;
  n_body = body;
  aln_len = static_cast<u32>(L - clip_tail - clip_head);
  t_pos = t_pos - static_cast<u32>((bw - 1) / 2) + static_cast<u32>(r);
}

// simple_aln::edit_distance with the reference's integer types
// (src/AbismalAlign.hpp:73-89; oplen() narrows to uint8_t, abismal_cigar_utils.hpp:50-53)
__device__ __forceinline__ int edit_distance(int scr, u32 len, const u32 *cig, u32 n_ops) {
  if (scr == 0)
    return static_cast<i16>(len);
  int ins = 0, del = 0;
  for (u32 k = 0; k < n_ops; ++k) {
    const u32 x = cig[k];
    const int oplen = static_cast<u8>(x >> 4);
    if ((x & 15u) == 1u) ins = static_cast<i16>(ins + oplen);
    if ((x & 15u) == 2u) del = static_cast<i16>(del + oplen);
  }
  const int A = static_cast<i16>(scr + 4 * (ins + del));
  const u32 num = 2u * (len - static_cast<u32>(ins)) - static_cast<u32>(A);
  const int mism = static_cast<i16>(num / 5u);
  return static_cast<i16>(mism + ins + del);
}

__device__ __forceinline__ bool long_enough(u32 aln_len, u32 readlen) {
  const double min_frac = 1.0 - 0.4;  // src/abismal.cpp:307-314
  return aln_len >= max(kMinReadLen, static_cast<u32>(min_frac * readlen));
}

// encoding index of the query a hit was found with (src/abismal.cpp:1463-1464)
__device__ __forceinline__ u32 enc_of(u32 flags) {
  const u32 rc = (flags & kFlagRC) ? 1u : 0u, ar = (flags & kFlagARich) ? 1u : 0u;
  return rc * 2u + (rc ^ ar);
}

// align_se_candidates (src/abismal.cpp:1435-1497) on the wave-resident set
__device__ __forceinline__ void choose_se(const DevIndex &ix, const WaveLds &lds, u32 L, double frac,
                                          SeSet &S, Hit &best, u32 *cig_out, u32 cig_stride,
                                          u32 &n_ops, bool &overflow, u32 &n_aln) {
  const int lane = lane_id();
  const int Ls = static_cast<i16>(L);
  const int md = static_cast<i16>(frac * static_cast<u32>(Ls));  // valid_diffs_cutoff
  const int perfect = static_cast<i16>(2 * L);
  n_ops = 0;
  if (S.best_p != 0) {  // exact match: no alignment needed
    best.diffs = static_cast<i16>(S.best_d); best.flags = static_cast<u16>(S.best_f); best.pos = S.best_p;
    if (lane == 0) cig_out[0] = L << 4;
    n_ops = 1;
    return;
  }
  // prepare_for_alignments: order by (pos, flags), drop duplicates
  const bool mine = lane < S.sz;
  const u64 key = (static_cast<u64>(S.hp) << 16) | S.hf;
  bool dup = false;
  for (int k = 0; k < S.sz; ++k) {
    const u64 kk = rdlane(key, k);
    dup |= (mine && k < lane && kk == key);
  }
  const u64 uniq = __ballot(mine && !dup);
  int slot = 0;
  for (int k = 0; k < S.sz; ++k) {
    const u64 kk = rdlane(key, k);
    slot += ((uniq >> k) & 1) && kk < key;
  }
  const int n_uniq = __popcll(uniq);
  const int invalid_at = static_cast<i16>(0.4 * Ls);  // valid_hit, :323-326

  int top = 0;
  u32 top_pos = 0, b_pos = 0, b_flags = 0;
  int b_diffs = 0x7fff;
  int dummy_r, dummy_c;
  for (int s = 0; s < n_uniq; ++s) {
    const u64 who = __ballot(mine && !dup && slot == s);
    const int l = __builtin_ctzll(who);
    const u32 pos = rdlane(S.hp, l), flags = rdlane(S.hf, l);
    const int d = rdlane(S.hd, l);
    if (pos == 0 || !(d < invalid_at))
      continue;
    const int sc = wave_align<false>(ix.genome, lds.qpk + enc_of(flags) * lds.W, lds.W, static_cast<int>(L),
                                     d, md, pos, lds.tb, dummy_r, dummy_c);
    ++n_aln;
    if (sc > top) { b_diffs = d; b_flags = flags; b_pos = pos; top = sc; top_pos = pos; }
    else if (sc == top) {
      const u32 gap = pos > top_pos ? pos - top_pos : top_pos - pos;
      if (sc == perfect ? pos != top_pos : gap > 3u) b_flags |= kFlagAmbig;
    }
  }
  best.diffs = 0x7fff; best.flags = static_cast<u16>(b_flags); best.pos = 0;
  if (b_pos == 0)
    return;
  int br, bc;
  const int sc = wave_align<true>(ix.genome, lds.qpk + enc_of(b_flags) * lds.W, lds.W, static_cast<int>(L),
                                  b_diffs, md, b_pos, lds.tb, br, bc);
  __syncthreads();
  u32 alen = 0, pos = b_pos, n_body = 0;
  wave_cigar(lds.tb, lds.ctmp, static_cast<int>(L), b_diffs, md, sc, br, bc, cig_out, cig_stride, n_ops,
             n_body, alen, pos, overflow);
  __syncthreads();
  // NM from the score found by the scoring pass (best_scr), as the reference does
  const int nm = edit_distance(top, alen, lds.ctmp, n_body);
  if (long_enough(alen, static_cast<u32>(Ls)) && nm <= md) {
    best.diffs = static_cast<i16>(nm);
    best.pos = pos;
  }
  else
    n_ops = 0;
}

// =============================================================================
// Kernel 2: single-end mapping, one wave per read (persistent, strided).
// =============================================================================
__global__ __launch_bounds__(64) void map_se_kernel(SeArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = lane_id();
  WaveLds lds;
  lds.W = a.W;
  lds.WB = a.WB;
  lds.qpk = reinterpret_cast<u64 *>(smem);
  lds.qbits = lds.qpk + 4 * a.W;
  lds.ctmp = reinterpret_cast<u32 *>(lds.qbits + 4 * a.WB);
  lds.mark = reinterpret_cast<u16 *>(lds.ctmp + a.cig_stride);
  lds.tb = reinterpret_cast<u8 *>(lds.mark + 64);

  // (rc, a_rich) calls per mode, in the reference's order
  // T-rich :1556-1572 | A-rich (-A/-P) | random PBAT :1649-1676
  const u32 n_calls = a.mode == 2 ? 4u : 2u;
  const u32 call_rc = a.mode == 2 ? 0xCu /*0,0,1,1*/ : 0x2u /*0,1*/;
  const u32 call_ar = a.mode == 2 ? 0x6u /*0,1,1,0*/ : (a.mode == 1 ? 0x3u : 0x0u);

  WorkTally wt = {0, 0, 0, 0, 0};
  u32 n_aln = 0;
  bool overflow = false, too_long = false;

  for (u64 r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
    const u32 L = a.lens[r];
    Hit best;
    best.diffs = 0x7fff; best.flags = 0; best.pos = 0;
    u32 n_ops = 0;
    u32 *cig_out = a.cig + r * a.cig_stride;
    if (L > kMaxReadLen) too_long = true;
    if (L >= kMinReadLen && L <= kMaxReadLen) {
      // stage the four encodings and derive their 2-letter bit strings
      const u64 *src = a.packed + r * 4 * a.W;
      for (u32 k = lane; k < 4 * a.W; k += 64) lds.qpk[k] = src[k];
      __syncthreads();
      for (u32 e = 0; e < 4; ++e)
        for (u32 wb = 0; wb < a.WB; ++wb) {
          const u32 j = wb * 64 + lane;
          const bool b = j < L ? bit2(q_nibble(lds.qpk + e * a.W, j)) : true;
          const u64 word = __ballot(b);
          if (lane == 0) lds.qbits[e * a.WB + wb] = word;
        }
      __syncthreads();

      SeSet S;
      S.begin_read(L);
      for (u32 cidx = 0; cidx < n_calls; ++cidx) {
        const bool rc = (call_rc >> cidx) & 1u, ar = (call_ar >> cidx) & 1u;
        const bool g_to_a = rc != ar;  // get_conv_type, src/abismal.cpp:1261-1267
        const u32 enc = (rc ? 2u : 0u) + (g_to_a ? 1u : 0u);
        const u32 flags = (rc ? kFlagRC : 0u) | (ar ? kFlagARich : 0u);
        S.cutoff = S.good_cutoff;  // set_specific
        seed_pass<true>(a.ix, lds, enc, g_to_a, flags, L, S, wt);
        // should_do_sensitive, :367-370
        if (S.sz != static_cast<int>(kSeCap) || S.cutoff > S.good_cutoff) {
          S.cutoff = rdlane(S.hd, 0);  // set_sensitive
          seed_pass<false>(a.ix, lds, enc, g_to_a, flags, L, S, wt);
        }
      }
      choose_se(a.ix, lds, L, a.valid_frac, S, best, cig_out, a.cig_stride, n_ops, overflow, n_aln);
    }
    if (lane == 0) {
      a.res[r] = best;
      a.cig_n[r] = best.pos != 0 ? n_ops : 0u;
    }
  }
  if (a.work) {  // exact per-launch work tallies for the roofline model
    auto wsum = [&](u32 v) { u32 t; (void)wave_excl_sum(v, t); return t; };
    const u32 s0 = wsum(wt.seed_iters), s1 = wsum(wt.probes), s2 = wsum(wt.cands), s3 = wsum(wt.words);
    if (lane == 0) {
      atomicAdd(&a.work[0], static_cast<unsigned long long>(s0));
      atomicAdd(&a.work[1], static_cast<unsigned long long>(s1));
      atomicAdd(&a.work[2], static_cast<unsigned long long>(s2));
      atomicAdd(&a.work[3], static_cast<unsigned long long>(s3));
      atomicAdd(&a.work[4], static_cast<unsigned long long>(wt.updates));
      atomicAdd(&a.work[5], static_cast<unsigned long long>(n_aln));
    }
  }
  if (lane == 0 && (overflow || too_long))
    atomicOr(a.status, (overflow ? 1u : 0u) | (too_long ? 2u : 0u));
}

// ---- launchers ----------------------------------------------------------------
size_t se_lds_bytes(u32 W, u32 WB, u32 cig_stride, u32 max_len, double valid_frac) {
  const int md = static_cast<i16>(valid_frac * max_len);
  int bw = 2 * md + 1;
  if (bw > static_cast<int>(kMaxBand) || bw < 0) bw = kMaxBand;
  if (bw < 1) bw = 1;
  size_t b = static_cast<size_t>(4) * W * 8 + static_cast<size_t>(4) * WB * 8 + static_cast<size_t>(cig_stride) * 4 + 64 * 2;
  b += static_cast<size_t>(max_len + bw) * bw;
  return (b + 15) & ~static_cast<size_t>(15);
}

int se_resident_waves(u32 W, u32 WB, u32 cig_stride, u32 max_len, double valid_frac) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_se_kernel, 64,
                                                   se_lds_bytes(W, WB, cig_stride, max_len, valid_frac)) != hipSuccess)
    return 0;
  return per_cu * prop.multiProcessorCount;
}

hipError_t launch_pack_reads(const char *d_blob, const u64 *d_off, u64 n, u32 W, u64 *d_packed,
                             u32 *d_lens, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const u64 threads = n * W;
  const u32 blocks = static_cast<u32>((threads + 255) / 256);
  hipLaunchKernelGGL(pack_reads_kernel, dim3(blocks), dim3(256), 0, st, d_blob, d_off, n, W, d_packed, d_lens);
  return hipGetLastError();
}

hipError_t launch_map_se(const SeArgs &a, u32 max_len, u32 n_waves, hipStream_t st) {
  if (a.n_reads == 0) return hipSuccess;
  const size_t lds = se_lds_bytes(a.W, a.WB, a.cig_stride, max_len, a.valid_frac);
  const u32 blocks = static_cast<u32>(a.n_reads < n_waves ? a.n_reads : n_waves);
  hipLaunchKernelGGL(map_se_kernel, dim3(blocks), dim3(64), lds, st, a);
  return hipGetLastError();
}

}  // namespace abm
