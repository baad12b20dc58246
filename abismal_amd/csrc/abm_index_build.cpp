// abismal_amd host side: multi-threaded AbismalIndex builder (`abismal idx`,
// src/abismalidx.cpp:35-115 -> AbismalIndex::create_index, src/AbismalIndex.cpp:281-331).
// Produces files byte-identical to the reference's (pinned by the tRex1.idx md5).
// Parallel decomposition is ours: bucket occupancy by relaxed atomic increments
// over position chunks, per-block selection DP on a thread pool, scatter fill in
// any order followed by a total-order bucket sort (key, then descending
// position) that reproduces the reference's descending fill + stable_sort.
#include "abm_index_build.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <thread>

namespace abm {

namespace {

constexpr uint32_t K2 = 25, K3 = 16, DEPTH = 256, PAD = 32767, MAXN = 256;
constexpr uint32_t MASK2 = (1u << K2) - 1, MOD3 = 43046721u;
constexpr uint64_t BLOCK = 1000000;

using Span = std::pair<uint64_t, uint64_t>;

inline uint32_t nib(const uint64_t *g, uint64_t k) { return static_cast<uint32_t>(g[k >> 4] >> ((k & 15) << 2)) & 15u; }
inline uint32_t sym2(uint32_t n) { return (n & 5u) == 0u; }
inline uint32_t dig_t(uint32_t n) { return (((n & 4u) != 0u) << 1) | ((n & 1u) != 0u); }
inline uint32_t dig_a(uint32_t n) { return (((n & 8u) != 0u) << 1) | ((n & 2u) != 0u); }

// dna_four_bit_encoding as compiled (src/dna_four_bit_bisulfite.hpp:156-165): N -> 0
uint8_t genome_code(unsigned char c) {
  switch (c & 0xDF) {
  case 'A': return 1;  case 'B': return 14; case 'C': return 2;  case 'D': return 13;
  case 'G': return 4;  case 'H': return 11; case 'K': return 12; case 'M': return 3;
  case 'R': return 5;  case 'S': return 6;  case 'T': return 8;  case 'V': return 7;
  case 'W': return 9;  case 'Y': return 10;
  default: return 0;
  }
}

template <class F> void parallel_for(unsigned nt, uint64_t n_items, F &&body) {
  nt = std::max(1u, nt);
  if (nt == 1 || n_items <= 1) { for (uint64_t k = 0; k < n_items; ++k) body(k); return; }
  std::atomic<uint64_t> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&] { for (uint64_t k; (k = next.fetch_add(1)) < n_items;) body(k); });
  for (auto &x : th) x.join();
}

// three rolling keys positioned so that after step() they describe position p
struct Keys {
  const uint64_t *g;
  uint64_t n2, n3;
  uint32_t h2 = 0, ht = 0, ha = 0;
  Keys(const uint64_t *g_, uint64_t p, uint32_t spool2 = K2 - 1) : g(g_), n2(p), n3(p) {
    for (uint32_t k = 0; k < spool2; ++k) h2 = ((h2 << 1) | sym2(nib(g, n2++))) & MASK2;
    for (uint32_t k = 0; k + 1 < K3; ++k) {
      const uint32_t x = nib(g, n3++);
      ht = (ht * 3 + dig_t(x)) % MOD3;
      ha = (ha * 3 + dig_a(x)) % MOD3;
    }
  }
  void step() {
    h2 = ((h2 << 1) | sym2(nib(g, n2++))) & MASK2;
    const uint32_t x = nib(g, n3++);
    ht = (ht * 3 + dig_t(x)) % MOD3;
    ha = (ha * 3 + dig_a(x)) % MOD3;
  }
};

inline void bump(uint32_t *tab, uint32_t k) {
  __atomic_fetch_add(&tab[k], 1u, __ATOMIC_RELAXED);
}

}  // namespace

void load_fasta(const std::string &path, std::string &text, std::vector<std::string> &names,
                std::vector<uint32_t> &starts) {
  std::ifstream in(path, std::ios::binary);
  if (!in)
    throw std::runtime_error("failed to open genome file: " + path);
  text.assign(PAD, 'N');
  names.assign(1, "pad_start");
  starts.assign(1, 0u);
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') {
      names.push_back(line.substr(1, line.find_first_of(" \t") - 1));
      starts.push_back(static_cast<uint32_t>(text.size()));
    }
    else
      text += line;
  }
  if (names.size() < 2)
    throw std::runtime_error("no names found in genome file");
  names.push_back("pad_end");
  starts.push_back(static_cast<uint32_t>(text.size()));
  text.append(PAD, 'N');
  starts.push_back(static_cast<uint32_t>(text.size()));
}

// `abismal idx -A targets` (src/AbismalIndex.cpp:83-123, :188-205, :206-241): only the listed regions are
// indexed -- everything outside them is turned into N before the usual N handling, so it lands in the
// long excluded runs.  The file has one "chrom start end" per line (0-based, end exclusive); regions
// must be sorted within a chromosome and are taken in the genome's chromosome order; names the genome
// does not have are dropped.
void mask_outside_targets(const std::string &targets_path, std::string &text, const std::vector<std::string> &names,
                          const std::vector<uint32_t> &starts) {
  std::ifstream in(targets_path);
  if (!in)
    throw std::runtime_error("failed reading target file");
  struct Region { std::string chrom; uint64_t a, b; };
  std::vector<Region> listed;
  for (std::string line; std::getline(in, line);) {
    std::istringstream f(line);
    Region r;
    if (!(f >> r.chrom) || !(f >> r.a) || !(f >> r.b))
      throw std::runtime_error("failed parsing target region");
    listed.push_back(r);
  }
  // per chromosome of the genome, in its order: the FIRST run of consecutive lines naming it
  std::vector<std::pair<uint32_t, uint32_t>> spans;
  for (size_t c = 0; c < names.size(); ++c) {
    size_t u = 0;
    while (u < listed.size() && listed[u].chrom != names[c]) ++u;
    size_t v = u;
    while (v < listed.size() && listed[v].chrom == names[c]) ++v;
    for (size_t k = u; k + 1 < v; ++k)
      if (listed[k + 1].a < listed[k].a || (listed[k + 1].a == listed[k].a && listed[k + 1].b < listed[k].b))
        throw std::runtime_error("target regions not sorted");
    for (size_t k = u; k < v; ++k)  // ChromLookup::get_pos: 32-bit arithmetic (:1296-1303)
      spans.emplace_back(starts[c] + static_cast<uint32_t>(listed[k].a), starts[c] + static_cast<uint32_t>(listed[k].b));
  }
  // mask_non_target (:108-123): a position is blanked while it lies before the current region's start; once
  // inside or past it, it is kept and the regions ending at or before it are retired -- so the base just
  // past a region's end survives too
  const uint64_t G = text.size();
  uint64_t i = 0;
  size_t t = 0;
  while (i < G) {
    if (t == spans.size()) { std::fill(text.begin() + static_cast<std::ptrdiff_t>(i), text.end(), 'N'); break; }
    if (i < spans[t].first) {
      const uint64_t to = std::min<uint64_t>(spans[t].first, G);
      std::fill(text.begin() + static_cast<std::ptrdiff_t>(i), text.begin() + static_cast<std::ptrdiff_t>(to), 'N');
      i = to;
      continue;
    }
    while (t < spans.size() && spans[t].second <= i) ++t;
    ++i;
  }
}

void build_index(std::string &text, const std::vector<std::string> &names,
                 const std::vector<uint32_t> &starts, unsigned nt, HostIndex &out, uint32_t window) {
  if (window != 20 && window != 12)
    throw std::runtime_error("window size must be 20 or 12 (short reads)");
  const uint32_t WIN = window;
  out.window = window;
  nt = std::max(1u, nt);
  const uint64_t G = text.size();
  if (G >= (1ull << 32))
    throw std::runtime_error("genome too large for 32-bit positions");

  // long N runs are excluded from the index; shorter ones become LCG bases
  // (src/AbismalIndex.cpp:125-175, :295-304; generator src/AbismalIndex.hpp:39-61)
  std::vector<Span> runs;
  for (uint64_t i = 0; i < G;) {
    if (text[i] != 'N') { ++i; continue; }
    uint64_t j = i;
    while (j < G && text[j] == 'N') ++j;
    if (j - i > MAXN) runs.emplace_back(i, j);
    else {
      // defer: filled below in genome order with one LCG stream
    }
    i = j;
  }
  {
    uint32_t x = 1;
    size_t r = 0;
    for (uint64_t i = 0; i < G; ++i) {
      if (text[i] == 'N' && i < runs[r].first) {
        x = (1103515245u * x + 12345u) & 0x7fffffffu;
        text[i] = "ACGT"[x & 3];
      }
      if (runs[r].second <= i) ++r;
    }
  }

  out.chrom_names = names;
  out.chrom_starts = starts;
  out.max_candidates = 100;
  const uint64_t gwords = (G + 15) / 16;
  out.genome.assign(gwords + 2, 0);
  parallel_for(nt, (gwords + 65535) / 65536, [&](uint64_t c) {
    const uint64_t w0 = c * 65536, w1 = std::min(gwords, w0 + 65536);
    for (uint64_t w = w0; w < w1; ++w) {
      uint64_t v = 0;
      const uint64_t b = w * 16, e = std::min(G, b + 16);
      for (uint64_t k = b; k < e; ++k)
        v |= static_cast<uint64_t>(genome_code(static_cast<unsigned char>(text[k]))) << ((k - b) << 2);
      out.genome[w] = v;
    }
  });
  std::string().swap(text);
  const uint64_t *g = out.genome.data();

  // indexable stretches: strictly between one long run's end and the next one's
  // start -- the reference's cursor logic also skips the first base after a run
  // (src/AbismalIndex.cpp:355-364)
  std::vector<Span> open;
  for (size_t r = 0; r + 1 < runs.size(); ++r)
    if (runs[r].second + 1 < runs[r + 1].first)
      open.emplace_back(runs[r].second + 1, runs[r + 1].first);
  if (!runs.empty() && runs.front().first > 0)
    open.insert(open.begin(), Span(0, runs.front().first));
  const uint64_t lim2 = G - K2 + 1, lim3 = G - K3 + 1;

  // split the open stretches into chunks for the counting/fill passes
  std::vector<Span> chunks;
  for (const Span &s : open)
    for (uint64_t a = s.first; a < s.second; a += BLOCK)
      chunks.emplace_back(a, std::min(s.second, a + BLOCK));

  // DP / selection blocks (src/AbismalIndex.cpp:438-469): these start AT the run end
  std::vector<Span> blocks;
  {
    uint64_t cur = 0;
    size_t r = 0;
    while (cur < lim2 && r < runs.size()) {
      if (cur < runs[r].first) {
        blocks.emplace_back(cur, std::min({runs[r].first, cur + BLOCK, lim2}));
        cur += BLOCK;
        if (cur >= runs[r].second) cur = runs[r++].second;
      }
      else
        cur = runs[r++].second;
    }
    for (; cur < lim2; cur += BLOCK) blocks.emplace_back(cur, std::min(cur + BLOCK, lim2));
  }

  out.counter_size = 1ull << K2;
  out.counter_size3 = MOD3;
  std::vector<uint8_t> keep(G, 1), two(G, 0);

  auto count = [&](bool masked) {
    out.counter.assign(out.counter_size + 1, 0);
    out.counter_t.assign(out.counter_size3 + 1, 0);
    out.counter_a.assign(out.counter_size3 + 1, 0);
    uint32_t *c2 = out.counter.data(), *ct = out.counter_t.data(), *ca = out.counter_a.data();
    parallel_for(nt, chunks.size(), [&](uint64_t k) {
      const Span s = chunks[k];
      Keys h(g, s.first);
      for (uint64_t p = s.first; p < s.second; ++p) {
        h.step();
        if (!keep[p]) continue;
        if (p < lim2 && (!masked || two[p])) bump(c2, h.h2);
        if (p < lim3 && (!masked || !two[p])) { bump(ct, h.ht); bump(ca, h.ha); }
      }
    });
  };
  count(false);

  // alphabet choice (src/AbismalIndex.cpp:471-543) fused with the windowed
  // selection DP (:643-855): per block, cost[i] = occupancy of the bucket the
  // position would join; choose a min-cost subset leaving no 20-gap
  std::fill(keep.begin(), keep.end(), 0);
  parallel_for(nt, blocks.size(), [&](uint64_t bk) {
    const Span b = blocks[bk];
    const uint64_t n = b.second - b.first;
    {
      Keys h(g, b.first);
      for (uint64_t p = b.first; p < b.second; ++p) {
        h.step();
        two[p] = out.counter[h.h2] <= ((out.counter_t[h.ht] + out.counter_a[h.ha]) >> 1);
      }
    }
    if (n < WIN) return;
    constexpr uint64_t NONE = ~0ull;
    std::vector<uint64_t> cost(n);
    std::vector<uint32_t> from(n);
    Keys h(g, b.first, static_cast<uint32_t>(std::min<uint64_t>(n, K2 - 1)));
    uint64_t qc[32];
    uint32_t qp[32], qf = 0, qb = 0;
    for (uint64_t i = 0; i < n; ++i) {
      h.step();
      const uint64_t c = two[b.first + i] ? out.counter[h.h2]
                                          : ((out.counter_t[h.ht] + out.counter_a[h.ha]) >> 1);
      if (i < WIN) { cost[i] = c; from[i] = 0xffffffffu; }
      else { cost[i] = qc[qf] + c; from[i] = qp[qf]; }
      while (qf != qb && qc[(qb - 1) & 31] > cost[i]) qb = (qb - 1) & 31;
      qc[qb] = cost[i]; qp[qb] = static_cast<uint32_t>(i); qb = (qb + 1) & 31;
      while (static_cast<uint64_t>(qp[qf]) + WIN <= i) qf = (qf + 1) & 31;
    }
    uint64_t best = NONE;
    uint32_t last = 0xffffffffu;
    for (uint64_t k = 0; k < WIN; ++k) {
      const uint64_t i = n - 1 - k;
      if (cost[i] < best) { best = cost[i]; last = static_cast<uint32_t>(i); }
    }
    for (uint32_t p = last; p != 0xffffffffu; p = from[p]) keep[b.first + p] = 1;
  });

  count(true);

  // exclusive bucket offsets (the reference ends with counter[k] = bucket start)
  auto to_offsets = [](std::vector<uint32_t> &v) {
    uint32_t s = 0;
    for (auto &x : v) { const uint32_t c = x; x = s; s += c; }
    return s;
  };
  out.index_size = to_offsets(out.counter);
  out.index_size3 = to_offsets(out.counter_t);
  to_offsets(out.counter_a);
  out.index.assign(out.index_size, 0);
  out.index_t.assign(out.index_size3, 0);
  out.index_a.assign(out.index_size3, 0);

  // scatter fill (any order) with per-bucket cursors
  {
    std::vector<uint32_t> cur2(out.counter.begin(), out.counter.end() - 1),
      cur_t(out.counter_t.begin(), out.counter_t.end() - 1),
      cur_a(out.counter_a.begin(), out.counter_a.end() - 1);
    parallel_for(nt, chunks.size(), [&](uint64_t k) {
      const Span s = chunks[k];
      Keys h(g, s.first);
      for (uint64_t p = s.first; p < s.second && p < lim2; ++p) {
        h.step();
        if (!keep[p]) continue;
        if (two[p])
          out.index[__atomic_fetch_add(&cur2[h.h2], 1u, __ATOMIC_RELAXED)] = static_cast<uint32_t>(p);
        else {
          out.index_t[__atomic_fetch_add(&cur_t[h.ht], 1u, __ATOMIC_RELAXED)] = static_cast<uint32_t>(p);
          out.index_a[__atomic_fetch_add(&cur_a[h.ha], 1u, __ATOMIC_RELAXED)] = static_cast<uint32_t>(p);
        }
      }
    });
  }
  std::vector<uint8_t>().swap(keep);
  std::vector<uint8_t>().swap(two);

  // bucket order (src/AbismalIndex.cpp:857-978): by the symbols that follow the
  // hashed prefix up to 256 letters; ties keep the fill order, i.e. descending
  // position.  A total order, so the parallel scatter above is harmless.
  auto sort_table = [&](const std::vector<uint32_t> &cnt, uint64_t n_buckets, std::vector<uint32_t> &idx,
                        int alphabet) {
    const uint32_t skip = alphabet == 2 ? K2 : K3;
    auto symbol = [&](uint32_t x) { return alphabet == 2 ? sym2(x) : alphabet == 0 ? (x & 5u) : (x & 10u); };
    auto less = [&](uint32_t a, uint32_t b) {
      for (uint32_t k = skip; k < DEPTH; ++k) {
        const uint32_t sa = symbol(nib(g, static_cast<uint64_t>(a) + k)),
                       sb = symbol(nib(g, static_cast<uint64_t>(b) + k));
        if (sa != sb) return sa < sb;
      }
      return a > b;
    };
    constexpr uint64_t GRAIN = 1 << 16;
    parallel_for(nt, (n_buckets + GRAIN - 1) / GRAIN, [&](uint64_t c) {
      const uint64_t k0 = c * GRAIN, k1 = std::min(n_buckets, k0 + GRAIN);
      for (uint64_t k = k0; k < k1; ++k)
        if (cnt[k + 1] > cnt[k] + 1)
          std::sort(idx.begin() + cnt[k], idx.begin() + cnt[k + 1], less);
    });
  };
  sort_table(out.counter, out.counter_size, out.index, 2);
  sort_table(out.counter_t, out.counter_size3, out.index_t, 0);
  sort_table(out.counter_a, out.counter_size3, out.index_a, 1);
}

void write_index(const HostIndex &h, const std::string &path) {
  struct Closer { void operator()(FILE *f) const { if (f) std::fclose(f); } };
  std::unique_ptr<FILE, Closer> fp(std::fopen(path.c_str(), "wb"));
  if (!fp)
    throw std::runtime_error("cannot open output file " + path);
  FILE *f = fp.get();
  auto put = [&](const void *p, size_t bytes) {
    if (bytes && std::fwrite(p, 1, bytes, f) != bytes)
      throw std::runtime_error("failed writing index");
  };
  put("AbismalIndex", 12);
  const uint32_t seed[3] = {K2, h.window, DEPTH};
  put(seed, sizeof(seed));
  const uint32_t n_chroms = static_cast<uint32_t>(h.chrom_names.size());
  put(&n_chroms, 4);
  for (const auto &nm : h.chrom_names) {
    const uint32_t len = static_cast<uint32_t>(nm.size());
    put(&len, 4);
    put(nm.data(), len);
  }
  put(h.chrom_starts.data(), h.chrom_starts.size() * 4);
  const uint64_t gwords = (static_cast<uint64_t>(h.chrom_starts.back()) + 15) / 16;
  put(h.genome.data(), gwords * 8);
  put(&h.max_candidates, 4);
  put(&h.counter_size, 8);
  put(&h.counter_size3, 8);
  put(&h.index_size, 8);
  put(&h.index_size3, 8);
  put(h.counter.data(), h.counter.size() * 4);
  put(h.counter_t.data(), h.counter_t.size() * 4);
  put(h.counter_a.data(), h.counter_a.size() * 4);
  put(h.index.data(), h.index.size() * 4);
  put(h.index_t.data(), h.index_t.size() * 4);
  put(h.index_a.data(), h.index_a.size() * 4);
  FILE *raw = fp.release();
  if (std::fclose(raw) != 0)
    throw std::runtime_error("problem closing file: " + path);
}

}  // namespace abm
