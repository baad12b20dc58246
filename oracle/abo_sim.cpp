// ORACLE — test infrastructure only (see abo_common.hpp).
// Restatement of `abismal sim` (src/simreads.cpp) far enough to regenerate the
// FASTQ inputs the reference's regression tests map (data/md5sum.txt:1-7).
// libstdc++'s mt19937 + uniform_{int,real}_distribution are used directly, as
// the reference does (src/simreads.cpp:54-78); the goldens bake those in.
#include "abo_sim.hpp"

#include <algorithm>
#include <fstream>
#include <random>
#include <stdexcept>

namespace abo {

namespace {

struct Rng {
  std::mt19937 eng;
  std::uniform_real_distribution<double> real;        // [0,1)
  std::uniform_int_distribution<std::uint64_t> whole;  // [0, 2^64-1]
  explicit Rng(std::size_t seed) : eng(seed) {}
  u64 draw() { return whole(eng); }
  double unit() { return real(eng); }
};

std::string revcomp_frag(const std::string &s) {
  std::string t(s.rbegin(), s.rend());
  for (char &c : t)
    c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
  return t;
}

// the reference grows `read` inside a loop whose bound shrinks as it grows
// (src/simreads.cpp:116-118, :128-130): only about half the gap is filled
void pad_like_reference(std::string &read, std::size_t want, BaseLCG &lcg) {
  for (std::size_t i = 0; i < want - read.size(); ++i)
    read += lcg.next();
}

}  // namespace

void simulate_reads(const SimParams &p) {
  Rng rng(p.seed);
  BaseLCG lcg;  // shared by mutation and padding, as one process-global stream

  std::string genome;
  ChromTable ct;
  load_fasta_padded(p.fasta, genome, ct);
  for (char &c : genome)
    c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));

  std::ofstream out1(p.out_prefix + "_1.fq");
  if (!out1)
    throw std::runtime_error("bad output file: " + p.out_prefix + "_1.fq");
  std::ofstream out2;
  if (!p.single_end) {
    out2.open(p.out_prefix + "_2.fq");
    if (!out2)
      throw std::runtime_error("bad output file: " + p.out_prefix + "_2.fq");
  }

  // change-type thresholds: src/simreads.cpp:351-362
  double sub = p.sub_rate, ins = p.ins_rate, del = p.del_rate;
  const double tot = std::max(sub + ins + del, std::numeric_limits<double>::min());
  sub /= tot; ins /= tot; del /= tot;
  ins += sub;

  auto emit = [&](std::ofstream &o, const std::string &name, const std::string &read) {
    o << '@' << name << '\n' << read << "\n+\n" << std::string(read.size(), 'B') << '\n';
  };

  for (std::size_t n = 0; n < p.n_reads; ++n) {
    // fragment length, position, strand: src/simreads.cpp:273-341
    std::size_t flen = p.min_frag;
    if (p.max_frag != p.min_frag)
      flen = p.min_frag + rng.draw() % (p.max_frag - p.min_frag);
    const std::size_t pos = rng.draw() % (genome.size() - flen + 1);
    std::string frag = genome.substr(pos, flen);
    const bool forward = p.strand == 'f' ? true : p.strand == 'r' ? false : (rng.draw() & 1);
    if (!forward)
      frag = revcomp_frag(frag);

    // point mutations and indels: src/simreads.cpp:363-411
    std::string seq;
    for (std::size_t i = 0; i < frag.size();) {
      if (rng.unit() > p.mut_rate) { seq += frag[i++]; continue; }
      const double y = rng.unit();
      if (y < sub) { seq += lcg.next(); ++i; }
      else if (y < ins) { seq += lcg.next(); }
      else { ++i; }
    }

    // bisulfite conversion: src/simreads.cpp:160-175
    const bool ga = p.pbat || (p.random_pbat && rng.unit() < 0.5);
    const char from = ga ? 'G' : 'C', to = ga ? 'A' : 'T';
    for (char &c : seq)
      if (c == from && rng.unit() < p.bs_conv)
        c = to;

    const std::string name = "read" + std::to_string(n);
    std::string r1 = seq.substr(0, p.read_len);
    pad_like_reference(r1, p.read_len, lcg);
    emit(out1, name + ".1", r1);
    if (!p.single_end) {
      std::string r2 = revcomp_frag(seq).substr(0, p.read_len);
      pad_like_reference(r2, p.read_len, lcg);
      emit(out2, name + ".2", r2);
    }
  }
}

}  // namespace abo
