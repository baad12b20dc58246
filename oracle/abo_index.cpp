// ORACLE — test infrastructure only (see abo_common.hpp).
// Restatement of the AbismalIndex builder / reader / writer.
#include "abo_index.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace abo {

// ---------------------------------------------------------------- chrom table
void ChromTable::locate(u32 pos, i32 &chrom, u32 &off) const {
  auto it = std::upper_bound(starts.begin(), starts.end(), pos);
  --it;
  chrom = static_cast<i32>(it - starts.begin());
  off = pos - starts[chrom];
}

bool ChromTable::locate(u32 pos, u32 reflen, i32 &chrom, u32 &off) const {
  auto it = std::upper_bound(starts.begin(), starts.end(), pos);
  if (it == starts.begin())
    return false;
  --it;
  chrom = static_cast<i32>(it - starts.begin());
  off = pos - starts[chrom];
  return pos + reflen <= starts[chrom + 1];
}

void load_fasta_padded(const std::string &path, std::string &genome, ChromTable &ct) {
  std::ifstream in(path);
  if (!in)
    throw std::runtime_error("failed to open genome file: " + path);
  genome.clear();
  ct.names.clear();
  ct.starts.clear();
  ct.names.push_back("pad_start");
  ct.starts.push_back(0);
  genome.append(kPadding, 'N');
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') {
      // name = first whitespace-delimited token after '>'
      ct.names.push_back(line.substr(1, line.find_first_of(" \t") - 1));
      ct.starts.push_back(static_cast<u32>(genome.size()));
    }
    else
      genome += line;
  }
  if (ct.names.size() < 2)
    throw std::runtime_error("no names found in genome file");
  ct.names.push_back("pad_end");
  ct.starts.push_back(static_cast<u32>(genome.size()));
  genome.append(kPadding, 'N');
  ct.starts.push_back(static_cast<u32>(genome.size()));
}

// ------------------------------------------------------------------- file I/O
namespace {
struct File {
  FILE *f;
  File(const std::string &p, const char *mode) : f(std::fopen(p.c_str(), mode)) {
    if (!f)
      throw std::runtime_error(std::string("cannot open ") + p);
  }
  ~File() { if (f) std::fclose(f); }
  template <class T> void put(const T &x) {
    if (std::fwrite(&x, sizeof(T), 1, f) != 1)
      throw std::runtime_error("failed writing index");
  }
  template <class T> void put(const T *p, std::size_t n) {
    if (std::fwrite(p, sizeof(T), n, f) != n)
      throw std::runtime_error("failed writing index");
  }
  template <class T> void get(T &x) {
    if (std::fread(&x, sizeof(T), 1, f) != 1)
      throw std::runtime_error("failed loading index file");
  }
  template <class T> void get(T *p, std::size_t n) {
    if (std::fread(p, sizeof(T), n, f) != n)
      throw std::runtime_error("failed loading index file");
  }
};
const char kMagic[] = "AbismalIndex";  // 12 bytes, no NUL on disk
}  // namespace

void Index::write(const std::string &path) const {
  File o(path, "wb");
  o.put(kMagic, 12);
  o.put(kKeyWeight);
  o.put(window);
  o.put(kSortDepth);
  const u32 n_chroms = static_cast<u32>(chroms.names.size());
  o.put(n_chroms);
  for (const auto &nm : chroms.names) {
    const u32 len = static_cast<u32>(nm.size());
    o.put(len);
    o.put(nm.data(), len);
  }
  o.put(chroms.starts.data(), n_chroms + 1);
  o.put(genome.data(), genome.size());
  o.put(max_candidates);
  o.put(counter_size);
  o.put(counter_size3);
  o.put(index_size);
  o.put(index_size3);
  o.put(counter.data(), counter_size + 1);
  o.put(counter_t.data(), counter_size3 + 1);
  o.put(counter_a.data(), counter_size3 + 1);
  o.put(index.data(), index_size);
  o.put(index_t.data(), index_size3);
  o.put(index_a.data(), index_size3);
}

void Index::read(const std::string &path) {
  File in(path, "rb");
  char magic[12];
  in.get(magic, 12);
  if (std::memcmp(magic, kMagic, 12) != 0)
    throw std::runtime_error("index file format problem: " + path);
  u32 kw = 0, ws = 0, sd = 0;
  in.get(kw);
  if (kw != kKeyWeight)
    throw std::runtime_error("inconsistent k-mer size. Expected: 25, got: " + std::to_string(kw));
  in.get(ws);
  if (ws != 20 && ws != 12)  // (a reference binary accepts only the value it was compiled with)
    throw std::runtime_error("inconsistent window size size. Expected: 20, got: " + std::to_string(ws));
  window = ws;
  in.get(sd);
  if (sd != kSortDepth)
    throw std::runtime_error("inconsistent sorting size size. Expected: 256, got: " + std::to_string(sd));
  u32 n_chroms = 0;
  in.get(n_chroms);
  chroms.names.assign(n_chroms, std::string());
  for (auto &nm : chroms.names) {
    u32 len = 0;
    in.get(len);
    nm.resize(len);
    in.get(&nm[0], len);
  }
  chroms.starts.assign(n_chroms + 1, 0);
  in.get(chroms.starts.data(), n_chroms + 1);
  genome.assign((static_cast<u64>(chroms.genome_size()) + 15) / 16, 0);
  in.get(genome.data(), genome.size());
  in.get(max_candidates);
  in.get(counter_size);
  in.get(counter_size3);
  in.get(index_size);
  in.get(index_size3);
  counter.assign(counter_size + 1, 0);
  counter_t.assign(counter_size3 + 1, 0);
  counter_a.assign(counter_size3 + 1, 0);
  in.get(counter.data(), counter.size());
  in.get(counter_t.data(), counter_t.size());
  in.get(counter_a.data(), counter_a.size());
  index.assign(index_size, 0);
  index_t.assign(index_size3, 0);
  index_a.assign(index_size3, 0);
  in.get(index.data(), index.size());
  in.get(index_t.data(), index_t.size());
  in.get(index_a.data(), index_a.size());
}

// -------------------------------------------------------------------- builder
namespace {

using Span = std::pair<u64, u64>;  // [first, second)

// maximal runs of 'N' strictly longer than kMaxNRun
// (src/AbismalIndex.cpp:125-145 + :297-302)
std::vector<Span> long_n_runs(const std::string &g) {
  std::vector<Span> runs;
  u64 i = 0;
  const u64 n = g.size();
  while (i < n) {
    if (g[i] != 'N') { ++i; continue; }
    u64 j = i;
    while (j < n && g[j] == 'N') ++j;
    if (j - i > kMaxNRun)
      runs.emplace_back(i, j);
    i = j;
  }
  return runs;
}

// Positions the reference's counting loops regard as indexable:
// "i < current_run.first", with the run cursor advanced only once
// run.second <= i -- so the first base after a run is NOT eligible
// (src/AbismalIndex.cpp:355-364, :396-409, :585-594).
std::vector<u8> eligibility(u64 G, const std::vector<Span> &runs) {
  std::vector<u8> e(G, 0);
  std::size_t r = 0;
  for (u64 i = 0; i < G; ++i) {
    e[i] = (i < runs[r].first);
    if (runs[r].second <= i)
      ++r;
  }
  return e;
}

// Work blocks of <= 1e6 positions between long N runs
// (src/AbismalIndex.cpp:438-469, incl. its cursor arithmetic).
std::vector<Span> work_blocks(u64 step, u64 end, const std::vector<Span> &runs) {
  std::vector<Span> blocks;
  u64 cur = 0;
  std::size_t r = 0;
  while (cur < end && r < runs.size()) {
    if (cur < runs[r].first) {
      blocks.emplace_back(cur, std::min({runs[r].first, cur + step, end}));
      cur += step;
      if (cur >= runs[r].second)
        cur = runs[r++].second;
    }
    else
      cur = runs[r++].second;
  }
  for (; cur < end; cur += step)
    blocks.emplace_back(cur, std::min(cur + step, end));
  return blocks;
}

struct Hasher {  // rolling 2-letter / 3-letter keys over the packed genome
  const u64 *g;
  u64 next2 = 0, next3 = 0;  // next nibble to shift in
  u32 h2 = 0, ht = 0, ha = 0;
  Hasher(const u64 *g_, u64 start, u32 spool2, u32 spool3) : g(g_) {
    next2 = next3 = start;
    for (u32 k = 0; k < spool2; ++k) roll2(gnib(g, next2++), h2);
    for (u32 k = 0; k < spool3; ++k) {
      const u8 nt = gnib(g, next3++);
      roll3(nt, C_TO_T, ht);
      roll3(nt, G_TO_A, ha);
    }
  }
  void step() {
    roll2(gnib(g, next2++), h2);
    const u8 nt = gnib(g, next3++);
    roll3(nt, C_TO_T, ht);
    roll3(nt, G_TO_A, ha);
  }
};

}  // namespace

// create_index(targets_file, genome_file), src/AbismalIndex.cpp:206-241: load_target_regions (:83-106),
// sort_by_chrom (:188-205), ChromLookup::get_pos (:1296-1303), mask_non_target (:108-123)
static void keep_only_targets(const std::string &path, const ChromTable &ct, std::string &genome) {
  std::ifstream in(path);
  if (!in)
    throw std::runtime_error("failed reading target file");
  struct Iv { std::string chrom; std::size_t s = 0, e = 0; };
  std::vector<Iv> u;
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream iss(line);
    Iv iv;
    if (!(iss >> iv.chrom) || !(iss >> iv.s) || !(iss >> iv.e))
      throw std::runtime_error("failed parsing target region");
    u.push_back(iv);
  }
  auto less = [](const Iv &a, const Iv &b) {
    const int x = a.chrom.compare(b.chrom);
    return x < 0 || (x == 0 && (a.s < b.s || (a.s == b.s && a.e < b.e)));
  };
  std::vector<std::pair<u32, u32>> targets;
  for (std::size_t i = 0; i < ct.names.size(); ++i) {
    const auto first = std::find_if(u.begin(), u.end(), [&](const Iv &x) { return x.chrom == ct.names[i]; });
    const auto last = std::find_if(first, u.end(), [&](const Iv &x) { return x.chrom != ct.names[i]; });
    if (!std::is_sorted(first, last, less))
      throw std::runtime_error("target regions not sorted");
    for (auto it = first; it != last; ++it)
      targets.emplace_back(ct.starts[i] + static_cast<u32>(it->s), ct.starts[i] + static_cast<u32>(it->e));
  }
  auto t = targets.begin();
  for (std::size_t i = 0; i < genome.size(); ++i) {
    if (t == targets.end() || i < t->first)
      genome[i] = 'N';
    else
      while (t != targets.end() && t->second <= i)
        ++t;
  }
}

void Index::build_from_fasta(const std::string &fasta, unsigned n_threads, const std::string &targets, u32 window_size) {
  if (window_size != 20 && window_size != 12)
    throw std::runtime_error("window size must be 20 or 12 (--enable-short)");
  window = window_size;
  const u64 kWindow = window;  // (shadows the default: everything below follows this index's window)
  std::string text;
  load_fasta_padded(fasta, text, chroms);
  if (!targets.empty())
    keep_only_targets(targets, chroms, text);
  const u64 G = text.size();
  const std::vector<Span> runs = long_n_runs(text);

  // short N runs become pseudo-random bases, in genome order
  // (src/AbismalIndex.cpp:164-175; LCG of src/AbismalIndex.hpp:39-61)
  {
    BaseLCG lcg;
    std::size_t r = 0;
    for (u64 i = 0; i < G; ++i) {
      if (i < runs[r].first && text[i] == 'N')
        text[i] = lcg.next();
      if (runs[r].second <= i)
        ++r;
    }
  }

  genome.assign((G + 15) / 16, 0);
  for (u64 k = 0; k < G; ++k)
    genome[k >> 4] |= static_cast<u64>(genome_nibble(static_cast<unsigned char>(text[k])))
                      << ((k & 15) << 2);
  std::string().swap(text);
  const u64 *g = genome.data();

  const std::vector<u8> elig = eligibility(G, runs);
  std::vector<u8> keep(G, 1), two(G, 0);

  counter_size = 1ull << kKeyWeight;
  counter_size3 = kHashMod3;
  const u64 lim2 = G - kKeyWeight + 1, lim3 = G - kKeyWeight3 + 1;

  // bucket occupancy; masked=false counts every eligible kept position in all
  // three tables, masked=true only where the position was assigned to that
  // alphabet (src/AbismalIndex.cpp:333-436)
  auto count_all = [&](bool masked) {
    counter.assign(counter_size + 1, 0);
    counter_t.assign(counter_size3 + 1, 0);
    counter_a.assign(counter_size3 + 1, 0);
    auto c2 = [&] {
      u32 h = 0;
      u64 nx = 0;
      for (u32 k = 0; k + 1 < kKeyWeight; ++k) roll2(gnib(g, nx++), h);
      for (u64 i = 0; i < lim2; ++i) {
        roll2(gnib(g, nx++), h);
        if (elig[i] && keep[i])
          counter[h] += (!masked || two[i]);
      }
    };
    auto c3 = [&](Conv cv, std::vector<u32> &tab) {
      u32 h = 0;
      u64 nx = 0;
      for (u32 k = 0; k + 1 < kKeyWeight3; ++k) roll3(gnib(g, nx++), cv, h);
      for (u64 i = 0; i < lim3; ++i) {
        roll3(gnib(g, nx++), cv, h);
        if (elig[i] && keep[i])
          tab[h] += (!masked || !two[i]);
      }
    };
    if (n_threads > 1) {
      std::thread a(c2), b(c3, C_TO_T, std::ref(counter_t)), c(c3, G_TO_A, std::ref(counter_a));
      a.join(); b.join(); c.join();
    }
    else {
      c2(); c3(C_TO_T, counter_t); c3(G_TO_A, counter_a);
    }
  };
  count_all(false);

  const std::vector<Span> blocks = work_blocks(1000000, lim2, runs);

  auto parallel_blocks = [&](auto &&body) {
    const std::size_t nb = blocks.size();
    const std::size_t nt = std::max<std::size_t>(1, std::min<std::size_t>(n_threads, nb));
    if (nt == 1) { for (const auto &b : blocks) body(b); return; }
    std::vector<std::thread> th;
    for (std::size_t t = 0; t < nt; ++t)
      th.emplace_back([&, t] { for (std::size_t k = t; k < nb; k += nt) body(blocks[k]); });
    for (auto &x : th) x.join();
  };

  // a position goes to the 2-letter table when its 2-letter bucket is no
  // fuller than the mean of its two 3-letter buckets
  // (src/AbismalIndex.cpp:471-543, costs :412-420)
  parallel_blocks([&](const Span &b) {
    Hasher h(g, b.first, kKeyWeight - 1, kKeyWeight3 - 1);
    for (u64 p = b.first; p < b.second; ++p) {
      h.step();
      two[p] = counter[h.h2] <= ((counter_t[h.ht] + counter_a[h.ha]) >> 1);
    }
  });

  // windowed DP: choose a min-cost subset with no 20 consecutive unchosen
  // positions inside each block (src/AbismalIndex.cpp:643-855)
  std::fill(keep.begin(), keep.end(), 0);
  parallel_blocks([&](const Span &b) {
    const u64 n = b.second - b.first;
    if (n < kWindow)
      return;
    constexpr u64 NONE = ~0ull;
    std::vector<u64> cost(n + 1), from(n + 1, NONE);
    Hasher h(g, b.first, static_cast<u32>(std::min<u64>(n, kKeyWeight - 1)), kKeyWeight3 - 1);
    // monotone queue of (cost,pos): front = cheapest in window, earliest on ties
    u64 qc[32], qp[32];
    u32 qf = 0, qb = 0;
    auto push = [&](u64 pos, u64 c) {
      while (qf != qb && qc[(qb - 1) & 31] > c) qb = (qb - 1) & 31;
      qc[qb] = c; qp[qb] = pos; qb = (qb + 1) & 31;
      while (qp[qf] + kWindow <= pos) qf = (qf + 1) & 31;
    };
    for (u64 i = 0; i < n; ++i) {
      h.step();
      const u64 c = two[b.first + i] ? counter[h.h2]
                                     : ((counter_t[h.ht] + counter_a[h.ha]) >> 1);
      if (i < kWindow) { cost[i] = c; from[i] = NONE; }
      else { cost[i] = qc[qf] + c; from[i] = qp[qf]; }
      push(i, cost[i]);
    }
    // cheapest of the last window, scanning right-to-left with strict <
    u64 best = NONE, last = NONE;
    for (u64 k = 0; k < kWindow; ++k) {
      const u64 i = n - 1 - k;
      if (cost[i] < best) { best = cost[i]; last = i; }
    }
    for (u64 p = last; p != NONE; p = from[p])
      keep[b.first + p] = 1;
  });
  max_candidates = 100;

  count_all(true);

  // bucket boundaries + fill (src/AbismalIndex.cpp:545-641): inclusive scan,
  // then ascending genome scan writing from each bucket's end downwards, which
  // leaves counter[] holding bucket starts and buckets in descending position
  auto scan = [](std::vector<u32> &v) { u32 s = 0; for (auto &x : v) { s += x; x = s; } };
  scan(counter); scan(counter_t); scan(counter_a);
  index_size = counter[counter_size];
  index_size3 = counter_t[counter_size3];
  index.assign(index_size, 0);
  index_t.assign(index_size3, 0);
  index_a.assign(index_size3, 0);
  {
    auto f2 = [&] {
      u32 h = 0; u64 nx = 0;
      for (u32 k = 0; k + 1 < kKeyWeight; ++k) roll2(gnib(g, nx++), h);
      for (u64 i = 0; i < lim2; ++i) {
        roll2(gnib(g, nx++), h);
        if (elig[i] && keep[i] && two[i]) index[--counter[h]] = static_cast<u32>(i);
      }
    };
    auto f3 = [&](Conv cv, std::vector<u32> &cnt, std::vector<u32> &idx) {
      u32 h = 0; u64 nx = 0;
      for (u32 k = 0; k + 1 < kKeyWeight3; ++k) roll3(gnib(g, nx++), cv, h);
      for (u64 i = 0; i < lim2; ++i) {
        roll3(gnib(g, nx++), cv, h);
        if (elig[i] && keep[i] && !two[i]) idx[--cnt[h]] = static_cast<u32>(i);
      }
    };
    if (n_threads > 1) {
      std::thread a(f2), b(f3, C_TO_T, std::ref(counter_t), std::ref(index_t)),
        c(f3, G_TO_A, std::ref(counter_a), std::ref(index_a));
      a.join(); b.join(); c.join();
    }
    else { f2(); f3(C_TO_T, counter_t, index_t); f3(G_TO_A, counter_a, index_a); }
  }

  // order each bucket by the letters that follow the hashed prefix, stable
  // w.r.t. the descending fill order (src/AbismalIndex.cpp:857-978)
  auto sort_table = [&](std::vector<u32> &cnt, u64 nb, std::vector<u32> &idx, int mode) {
    const u32 skip = (mode == 2) ? kKeyWeight : kKeyWeight3;
    auto less = [&](u32 a, u32 b) {
      for (u32 k = skip; k < kSortDepth; ++k) {
        const u8 x = gnib(g, static_cast<u64>(a) + k), y = gnib(g, static_cast<u64>(b) + k);
        const u32 sx = mode == 2 ? bit2(x) : sortsym3(x, mode == 0 ? C_TO_T : G_TO_A);
        const u32 sy = mode == 2 ? bit2(y) : sortsym3(y, mode == 0 ? C_TO_T : G_TO_A);
        if (sx != sy) return sx < sy;
      }
      return false;
    };
    const unsigned nt = std::max(1u, n_threads);
    std::vector<std::thread> th;
    const u64 per = (nb + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        const u64 lo = t * per, hi = std::min(nb, (t + 1) * per);
        for (u64 k = lo; k < hi; ++k)
          if (cnt[k + 1] > cnt[k] + 1)
            std::stable_sort(idx.begin() + cnt[k], idx.begin() + cnt[k + 1], less);
      });
    for (auto &x : th) x.join();
  };
  sort_table(counter, counter_size, index, 2);
  sort_table(counter_t, counter_size3, index_t, 0);
  sort_table(counter_a, counter_size3, index_a, 1);
}

}  // namespace abo
