// ORACLE — test infrastructure only (see abo_common.hpp).
// Flat C entry points so tests/ and bench.py's cpu_baseline leg can drive the
// restatement through ctypes.  Nothing in the shipped library links this.
#include "abo_map.hpp"
#include "abo_sim.hpp"

#include <atomic>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>

using namespace abo;

namespace {
thread_local std::string g_err;
template <class F> int guarded(F &&f) {
  try { f(); return 0; }
  catch (const std::exception &e) { g_err = e.what(); return -1; }
}
struct MapperBox {
  const Index *ix;
  MapParams par;
};
}  // namespace

extern "C" {

struct abo_hit { int16_t diffs; uint16_t flags; uint32_t pos; };
struct abo_pair { int16_t aln_score; int16_t pad; abo_hit r1, r2; };

const char *abo_last_error() { return g_err.c_str(); }

int abo_index_build_targets(const char *fasta, const char *targets, const char *out, unsigned threads) {
  return guarded([&] { Index ix; ix.build_from_fasta(fasta, threads, targets ? targets : ""); ix.write(out); });
}
int abo_index_build_opts(const char *fasta, const char *targets, unsigned window, const char *out, unsigned threads) {
  return guarded([&] { Index ix; ix.build_from_fasta(fasta, threads, targets ? targets : "", window); ix.write(out); });
}
int abo_index_build(const char *fasta, const char *out, unsigned threads) {
  return guarded([&] { Index ix; ix.build_from_fasta(fasta, threads); ix.write(out); });
}

void *abo_index_load(const char *path) {
  Index *ix = new Index;
  if (guarded([&] { ix->read(path); }) != 0) { delete ix; return nullptr; }
  return ix;
}
void abo_index_free(void *p) { delete static_cast<Index *>(p); }
uint32_t abo_index_max_candidates(void *p) { return static_cast<Index *>(p)->max_candidates; }
uint64_t abo_index_genome_size(void *p) { return static_cast<Index *>(p)->chroms.genome_size(); }

int abo_simulate(const char *fasta, const char *prefix, int single_end, int pbat, int random_pbat,
                 uint64_t read_len, uint64_t min_frag, uint64_t max_frag, uint64_t n_reads,
                 uint64_t seed, double mut, double bis) {
  return guarded([&] {
    SimParams p;
    p.fasta = fasta; p.out_prefix = prefix; p.single_end = single_end; p.pbat = pbat;
    p.random_pbat = random_pbat; p.read_len = read_len; p.min_frag = min_frag; p.max_frag = max_frag;
    p.n_reads = n_reads; p.seed = seed; p.mut_rate = mut; p.bs_conv = bis;
    simulate_reads(p);
  });
}

int abo_read_code(int c, int a_rich) { return read_nibble(static_cast<char>(c), a_rich != 0); }
int abo_genome_code(int c) { return genome_nibble(static_cast<unsigned char>(c)); }
int abo_get_bit(int nt) { return static_cast<int>(bit2(static_cast<u8>(nt))); }
int abo_trit(int nt, int g_to_a_conv) { return static_cast<int>(trit(static_cast<u8>(nt), g_to_a_conv ? G_TO_A : C_TO_T)); }
uint32_t abo_roll2(uint32_t k, int nt) { roll2(static_cast<u8>(nt), k); return k; }
uint32_t abo_roll3(uint32_t k, int nt, int g_to_a_conv) { roll3(static_cast<u8>(nt), g_to_a_conv ? G_TO_A : C_TO_T, k); return k; }

int abo_align(const uint64_t *genome, uint64_t /*n_words*/, const uint8_t *q, uint32_t qlen, int diffs,
              int max_diffs, uint32_t t_pos, int do_tb, uint32_t *cig_out, uint32_t cig_cap,
              uint32_t *n_cig, uint32_t *aln_len, uint32_t *new_pos, int *nm) {
  Cigar c;
  const int scr = probe_align(reinterpret_cast<const u64 *>(genome), q, qlen, static_cast<i16>(diffs),
                              static_cast<i16>(max_diffs), t_pos, do_tb != 0, &c, aln_len, new_pos, nm);
  if (do_tb) {
    *n_cig = static_cast<uint32_t>(c.size());
    for (std::size_t i = 0; i < c.size() && i < cig_cap; ++i) cig_out[i] = c[i];
  }
  return scr;
}

void *abo_mapper_new(void *index, uint32_t max_candidates, double valid_frac, uint32_t min_frag,
                     uint32_t max_frag, int allow_ambig) {
  MapperBox *b = new MapperBox;
  b->ix = static_cast<Index *>(index);
  b->par.max_candidates = max_candidates ? max_candidates : b->ix->max_candidates;
  b->par.valid_frac = valid_frac;
  b->par.min_frag = min_frag;
  b->par.max_frag = max_frag;
  b->par.allow_ambig = allow_ambig != 0;
  return b;
}
void abo_mapper_free(void *p) { delete static_cast<MapperBox *>(p); }

// Reads are trimmed ASCII, concatenated in `blob` with n+1 offsets.  Results:
// out[i] as the reference's bests[i] just before format_se, CIGARs as BAM u32
// ops in fixed slots of `cig_stride` per read with their count in cig_n[i].
// work9 (optional) accumulates the Work tallies.  threads>1 splits the batch
// into contiguous shards, one Mapper per thread.
int abo_map_se(void *mapper, int mode, uint64_t n, const char *blob, const uint64_t *off,
               abo_hit *out, uint32_t *cig, uint32_t cig_stride, uint32_t *cig_n,
               unsigned threads, uint64_t *work9) {
  MapperBox *b = static_cast<MapperBox *>(mapper);
  std::atomic<int> bad{0};
  std::vector<Work> works(std::max(1u, threads));
  auto shard = [&](unsigned t, uint64_t lo, uint64_t hi) {
    try {
      Mapper mp(*b->ix, b->par);
      Cigar c;
      {  // the reads before this shard, back to one longer than 46 bases, leave their trace in the buffers
        uint64_t from = lo;
        while (from > 0) { --from; if (off[from + 1] - off[from] > 46) break; }  // (no cap: the reference's buffers never forget)
        for (uint64_t i = from; i < lo; ++i) mp.touch_se(std::string(blob + off[i], blob + off[i + 1]), static_cast<SeMode>(mode));
      }
      for (uint64_t i = lo; i < hi; ++i) {
        Hit h;
        c.clear();
        mp.map_se(std::string(blob + off[i], blob + off[i + 1]), static_cast<SeMode>(mode), h, c);
        out[i] = abo_hit{h.diffs, h.flags, h.pos};
        if (c.size() > cig_stride) throw std::runtime_error("cigar slot overflow");
        cig_n[i] = static_cast<uint32_t>(c.size());
        std::memcpy(cig + i * cig_stride, c.data(), c.size() * 4);
      }
      works[t] = mp.work;
    }
    catch (const std::exception &e) { bad = 1; }
  };
  const unsigned nt = std::max(1u, threads);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back(shard, t, n * t / nt, n * (t + 1) / nt);
  for (auto &x : th) x.join();
  if (work9) {
    for (const Work &w : works) {
      const uint64_t v[9] = {w.reads, w.seed_iters, w.search_probes, w.candidates, w.words,
                             w.set_updates, w.aligns, w.aligns_tb, w.dp_cells};
      for (int k = 0; k < 9; ++k) work9[k] += v[k];
    }
  }
  if (bad) { g_err = "oracle shard failed"; return -1; }
  return 0;
}

int abo_map_pe(void *mapper, int mode, uint64_t n, const char *blob1, const uint64_t *off1,
               const char *blob2, const uint64_t *off2, abo_pair *out_pair, abo_hit *out_se1,
               abo_hit *out_se2, uint32_t *cig1, uint32_t *cig2, uint32_t cig_stride,
               uint32_t *cig_n1, uint32_t *cig_n2, unsigned threads, uint64_t *work9) {
  MapperBox *b = static_cast<MapperBox *>(mapper);
  std::atomic<int> bad{0};
  std::vector<Work> works(std::max(1u, threads));
  auto shard = [&](unsigned t, uint64_t lo, uint64_t hi) {
    try {
      Mapper mp(*b->ix, b->par);
      Cigar c1, c2;
      {  // (see abo_map_se)
        uint64_t from = lo;
        while (from > 0) {
          --from;
          if (off1[from + 1] - off1[from] > 46 && off2[from + 1] - off2[from] > 46) break;
        }
        for (uint64_t i = from; i < lo; ++i)
          mp.touch_pe(std::string(blob1 + off1[i], blob1 + off1[i + 1]), std::string(blob2 + off2[i], blob2 + off2[i + 1]),
                      static_cast<PeMode>(mode));
      }
      for (uint64_t i = lo; i < hi; ++i) {
        PairHit p;
        Hit h1, h2;
        c1.clear();
        c2.clear();
        mp.map_pe(std::string(blob1 + off1[i], blob1 + off1[i + 1]),
                  std::string(blob2 + off2[i], blob2 + off2[i + 1]), static_cast<PeMode>(mode), p,
                  h1, h2, c1, c2);
        out_pair[i] = abo_pair{p.aln_score, 0, {p.r1.diffs, p.r1.flags, p.r1.pos},
                               {p.r2.diffs, p.r2.flags, p.r2.pos}};
        out_se1[i] = abo_hit{h1.diffs, h1.flags, h1.pos};
        out_se2[i] = abo_hit{h2.diffs, h2.flags, h2.pos};
        if (c1.size() > cig_stride || c2.size() > cig_stride) throw std::runtime_error("cigar slot overflow");
        cig_n1[i] = static_cast<uint32_t>(c1.size());
        cig_n2[i] = static_cast<uint32_t>(c2.size());
        std::memcpy(cig1 + i * cig_stride, c1.data(), c1.size() * 4);
        std::memcpy(cig2 + i * cig_stride, c2.data(), c2.size() * 4);
      }
      works[t] = mp.work;
    }
    catch (const std::exception &e) { bad = 1; }
  };
  const unsigned nt = std::max(1u, threads);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back(shard, t, n * t / nt, n * (t + 1) / nt);
  for (auto &x : th) x.join();
  if (work9) {
    for (const Work &w : works) {
      const uint64_t v[9] = {w.reads, w.seed_iters, w.search_probes, w.candidates, w.words,
                             w.set_updates, w.aligns, w.aligns_tb, w.dp_cells};
      for (int k = 0; k < 9; ++k) work9[k] += v[k];
    }
  }
  if (bad) { g_err = "oracle shard failed"; return -1; }
  return 0;
}

}  // extern "C"
