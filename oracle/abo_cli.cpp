// ORACLE — test infrastructure only (see abo_common.hpp).
// Command-line front end `abismal_oracle {idx,sim,map}`: the host-side I/O the
// reference wraps around the mapping path (FASTQ reader, SAM text, stats),
// restated so that whole-file md5s can be compared with data/md5sum.txt.
#include "abo_map.hpp"
#include "abo_sim.hpp"

#include <algorithm>
#include <chrono>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

using namespace abo;

namespace {

// ---- FASTQ: src/abismal.cpp:150-209 ---------------------------------------
struct FastqReader {
  std::ifstream in;
  std::string path;
  u64 line_no = 0;
  bool alive = true;
  u32 min_len = kMinReadLen;  // fewer informative bases: the read is skipped (src/abismal.cpp:187-195, :212-213)
  explicit FastqReader(const std::string &p, u32 min_read_len = kMinReadLen) : in(p), path(p), min_len(min_read_len) {
    if (!in)
      throw std::runtime_error("cannot open reads file: " + p);
  }
  // appends up to `batch` records; returns false once the file is exhausted
  void load(std::size_t batch, std::vector<std::string> &names, std::vector<std::string> &reads) {
    names.clear();
    reads.clear();
    std::string line, name;
    std::size_t k = 0;
    while (k < 4 * batch) {
      if (!std::getline(in, line)) { alive = false; break; }
      if (k % 4 == 0) {
        if (line.empty())
          throw std::runtime_error("file " + path + " contains an empty read name at line " +
                                   std::to_string(line_no));
        name = line.substr(1, line.find_first_of(" \t") - 1);
      }
      else if (k % 4 == 1) {
        if (line.size() >= kPadding)
          throw std::runtime_error("found a read of size " + std::to_string(line.size()) +
                                   ", which is too long. Maximum allowed read size = " +
                                   std::to_string(kPadding));
        const auto informative =
          std::count_if(line.begin(), line.end(), [](char c) { return c != 'N'; });
        if (informative < static_cast<std::ptrdiff_t>(min_len))
          line.clear();
        else {
          while (line.back() == 'N') line.pop_back();
          line = line.substr(line.find_first_of("ACGT"));
        }
        names.push_back(name);
        reads.push_back(line);
      }
      ++k;
      ++line_no;
    }
  }
};

// ---- SAM text: src/abismal.cpp:481-545, :648-773 + SURVEY App. C -----------
const char kCigarOps[] = "MIDNSHP=XB";

void cigar_text(const Cigar &c, std::string &out) {
  for (u32 x : c) {
    out += std::to_string(x >> 4);
    out += kCigarOps[std::min<u32>(x & 15u, 9)];
  }
}

// SEQ as htslib prints it after its 4-bit round trip: IUPAC upper-cased, rest N
char seq_char(char c) {
  static const char ok[] = "=ACMGRSVTWYHKDBN";
  const char u = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
  return (u != '\0' && std::strchr(ok, u)) ? u : 'N';
}

struct SamRecord {
  const std::string *qname;
  u16 flag;
  i32 tid, mtid;
  u32 pos, mpos;
  int tlen;
  const Cigar *cig;
  std::string seq;
  i16 nm;
  char cv;
};

void put_record(std::string &o, const ChromTable &ct, const SamRecord &r) {
  o += *r.qname; o += '\t';
  o += std::to_string(r.flag); o += '\t';
  o += ct.names[r.tid + 1]; o += '\t';
  o += std::to_string(r.pos + 1); o += "\t255\t";
  cigar_text(*r.cig, o); o += '\t';
  if (r.mtid < 0) o += "*\t0\t";
  else {
    o += (r.mtid == r.tid) ? std::string("=") : ct.names[r.mtid + 1];
    o += '\t'; o += std::to_string(r.mpos + 1); o += '\t';
  }
  o += std::to_string(r.tlen); o += '\t';
  for (char c : r.seq) o += seq_char(c);
  o += "\t*\tNM:i:"; o += std::to_string(r.nm);
  o += "\tCV:A:"; o += r.cv; o += '\n';
}

enum Outcome { UNMAPPED, UNIQUE, AMBIG };

// format_se: src/abismal.cpp:481-545
Outcome emit_se(std::string &o, bool allow_ambig, const Hit &h, const ChromTable &ct,
                const std::string &name, const std::string &read, const Cigar &cig) {
  const bool ambig = h.ambig();
  if (!allow_ambig && ambig)
    return AMBIG;
  u32 off = 0;
  i32 chrom = 0;
  if (h.empty() || !ct.locate(h.pos, cigar_ref_len(cig), chrom, off))
    return UNMAPPED;
  SamRecord r{&name, 0, chrom - 1, -1, off, 0, 0, &cig, h.rc() ? revcomp_read(read) : read,
              h.diffs, h.a_rich() ? 'A' : 'T'};
  if (h.rc()) r.flag |= 0x10;
  if (allow_ambig && ambig) r.flag |= 0x100;
  put_record(o, ct, r);
  return ambig ? AMBIG : UNIQUE;
}

// format_pe: src/abismal.cpp:648-773
Outcome emit_pe(std::string &o, bool allow_ambig, const PairHit &p, const ChromTable &ct,
                const std::string &n1, const std::string &n2, const std::string &s1,
                const std::string &s2, const Cigar &c1, const Cigar &c2) {
  if (p.empty())
    return UNMAPPED;
  const bool ambig = p.ambig();
  if (!allow_ambig && ambig)
    return AMBIG;
  i32 ch1 = 0, ch2 = 0;
  u32 b1 = 0, b2 = 0;
  const u32 rl1 = cigar_ref_len(c1), rl2 = cigar_ref_len(c2);
  if (!ct.locate(p.r1.pos, rl1, ch1, b1) || !ct.locate(p.r2.pos, rl2, ch2, b2) || ch1 != ch2)
    return UNMAPPED;
  const u32 e2 = b2 + rl2;
  const bool rc = p.r1.rc();
  const int isize = rc ? static_cast<int>(b1) - static_cast<int>(e2)
                       : static_cast<int>(e2) - static_cast<int>(b1);
  u16 f1 = 0x1 | 0x2 | 0x40, f2 = 0x1 | 0x2 | 0x80;
  if (p.r1.rc()) { f1 |= 0x10; f2 |= 0x20; }
  if (p.r2.rc()) { f2 |= 0x10; f1 |= 0x20; }
  if (allow_ambig && ambig) { f1 |= 0x100; f2 |= 0x100; }
  SamRecord a{&n1, f1, ch1 - 1, ch2 - 1, b1, b2, isize, &c1,
              p.r1.rc() ? revcomp_read(s1) : s1, p.r1.diffs, p.r1.a_rich() ? 'A' : 'T'};
  SamRecord b{&n2, f2, ch2 - 1, ch1 - 1, b2, b1, -isize, &c2,
              p.r2.rc() ? revcomp_read(s2) : s2, p.r2.diffs, p.r2.a_rich() ? 'A' : 'T'};
  put_record(o, ct, a);
  put_record(o, ct, b);
  return ambig ? AMBIG : UNIQUE;
}

// ---- statistics: src/abismal.cpp:865-1071 ----------------------------------
struct Stats {
  u32 total = 0, unique = 0, ambiguous = 0, skipped = 0;  // u32 like the reference
  u64 edits = 0, bases = 0;
  void tally(bool empty_read, const Hit &h, bool count_ambig_error, const Cigar &c) {
    ++total;
    const bool valid = !h.empty(), amb = h.ambig();
    unique += valid && !amb;
    ambiguous += valid && amb;
    skipped += empty_read;
    if (valid && (!amb || count_ambig_error)) { edits += h.diffs; bases += cigar_ref_len(c); }
  }
  std::string yaml(const std::string &label) const {
    auto frac = [&](double x) { return total > 0 ? x / total : 0.0; };
    const u32 mapped = unique + ambiguous;
    const u32 unmapped = total - mapped;
    std::ostringstream s;
    const char *t = "    ";
    s << label << ":\n"
      << t << "total_reads: " << total << '\n'
      << t << "mapped:\n"
      << t << "    num_mapped: " << mapped << '\n'
      << t << "    num_unique: " << unique << '\n'
      << t << "    num_ambiguous: " << ambiguous << '\n'
      << t << "    percent_mapped: " << frac(mapped) * 100.0 << '\n'
      << t << "    percent_unique: " << frac(unique) * 100.0 << '\n'
      << t << "    percent_ambiguous: " << frac(ambiguous) * 100.0 << '\n'
      << t << "    unique_error:\n"
      << t << "        edits: " << edits << '\n'
      << t << "        total_bases: " << bases << '\n'
      << t << "        error_rate: " << (bases > 0 ? static_cast<double>(edits) / bases : 0.0) << '\n'
      << t << "num_unmapped: " << unmapped << '\n'
      << t << "num_skipped: " << skipped << '\n'
      << t << "percent_unmapped: " << frac(unmapped) * 100.0 << '\n'
      << t << "percent_skipped: " << frac(skipped) * 100.0 << '\n';
    return s.str();
  }
  std::string json() const {
    std::ostringstream s;
    s << "{\"edit_distance\":" << edits << ",\"reads_mapped_ambiguous\":" << ambiguous
      << ",\"reads_mapped_unique\":" << unique << ",\"reads_skipped\":" << skipped
      << ",\"total_bases\":" << bases << ",\"total_reads\":" << total << "}";
    return s.str();
  }
};

// header: src/abismal.cpp:2265-2293
std::string sam_header(const ChromTable &ct, int argc, char **argv) {
  std::ostringstream h;
  h << "@HD\tVN:1.0\n";
  for (std::size_t i = 1; i + 1 < ct.names.size(); ++i)
    h << "@SQ\tSN:" << ct.names[i] << "\tLN:" << (ct.starts[i + 1] - ct.starts[i]) << '\n';
  h << "@PG\tID:ABISMAL\tVN:3.3.0\tCL:\"";
  for (int i = 0; i < argc; ++i) h << argv[i] << ' ';
  h << "\"\n";
  return h.str();
}

struct Args {  // minimal getopt for the reference's short flags
  std::vector<std::string> pos;
  std::vector<std::pair<std::string, std::string>> kv;
  bool has(const std::string &k) const {
    for (auto &p : kv) if (p.first == k) return true;
    return false;
  }
  std::string get(const std::string &k, const std::string &d = "") const {
    for (auto &p : kv) if (p.first == k) return p.second;
    return d;
  }
};

Args parse(int argc, char **argv, const std::string &with_value) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s.size() >= 2 && s[0] == '-') {
      std::string key = s.substr(s.find_first_not_of('-'));
      const bool takes = (with_value.find("," + key + ",") != std::string::npos);
      a.kv.emplace_back(key, takes && i + 1 < argc ? argv[++i] : "1");
    }
    else
      a.pos.push_back(s);
  }
  return a;
}

int cmd_idx(int argc, char **argv) {
  Args a = parse(argc, argv, ",t,threads,A,targets,w,window,");
  if (a.pos.size() != 2) { std::cerr << "usage: idx [-t n] <genome.fa> <out.idx>\n"; return 1; }
  Index ix;
  ix.build_from_fasta(a.pos[0], static_cast<unsigned>(std::stoul(a.get("t", a.get("threads", "1")))), a.get("A", a.get("targets")),
                      static_cast<u32>(std::stoul(a.get("w", a.get("window", "20")))));
  ix.write(a.pos[1]);
  return 0;
}

int cmd_sim(int argc, char **argv) {
  Args a = parse(argc, argv, ",o,out,l,read-len,min-fraglen,max-fraglen,n,n-reads,m,mut,b,bis,seed,s,strand,");
  if (a.pos.size() != 1 || !(a.has("o") || a.has("out"))) { std::cerr << "usage: sim -o prefix [opts] <genome.fa>\n"; return 1; }
  SimParams p;
  p.fasta = a.pos[0];
  p.out_prefix = a.get("o", a.get("out"));
  p.single_end = a.has("single");
  p.pbat = a.has("a") || a.has("pbat");
  p.random_pbat = a.has("R") || a.has("random-pbat");
  p.read_len = std::stoul(a.get("l", a.get("read-len", "100")));
  p.min_frag = std::stoul(a.get("min-fraglen", "100"));
  p.max_frag = std::stoul(a.get("max-fraglen", "250"));
  p.n_reads = std::stoul(a.get("n", a.get("n-reads", "100")));
  p.mut_rate = std::stod(a.get("m", a.get("mut", "0")));
  p.bs_conv = std::stod(a.get("b", a.get("bis", "1")));
  p.seed = std::stoul(a.get("seed", "1"));
  p.strand = a.get("s", a.get("strand", "b"))[0];
  simulate_reads(p);
  return 0;
}

// the driver around the per-read bodies: src/abismal.cpp:1511-1600, :1887-2029, :2295-2504
int cmd_map(int argc, char **argv) {
  Args a = parse(argc, argv, ",i,index,o,outfile,s,stats,c,max-candidates,l,min-frag,L,max-frag,m,max-distance,t,threads,w,work,");
  const std::string index_file = a.get("i", a.get("index"));
  const std::string outfile = a.get("o", a.get("outfile"));
  if (index_file.empty() || outfile.empty() || a.pos.empty() || a.pos.size() > 2) {
    std::cerr << "usage: map -i idx -o out.sam [-s stats] [-a -P -R -A -j -c n -l n -L n -m f] r1.fq [r2.fq]\n";
    return 1;
  }
  const bool allow_ambig = a.has("a") || a.has("ambig");
  const bool pbat = a.has("P") || a.has("pbat");
  const bool rpbat = a.has("R") || a.has("random-pbat");
  const bool arich = a.has("A") || a.has("a-rich");

  Index ix;
  ix.read(index_file);
  MapParams par;
  par.max_candidates = ix.max_candidates;
  if (a.has("c") && std::stoul(a.get("c")) != 0) par.max_candidates = static_cast<u32>(std::stoul(a.get("c")));
  par.valid_frac = std::stod(a.get("m", a.get("max-distance", "0.1")));
  par.min_frag = static_cast<u32>(std::stoul(a.get("l", a.get("min-frag", "32"))));
  par.max_frag = static_cast<u32>(std::stoul(a.get("L", a.get("max-frag", "3000"))));
  par.allow_ambig = allow_ambig;

  std::ofstream out(outfile, std::ios::binary);
  if (!out)
    throw std::runtime_error("failed to open output file: " + outfile);
  out << sam_header(ix.chroms, argc, argv);

  Mapper mapper(ix, par);
  const auto t_start = std::chrono::steady_clock::now();
  std::vector<std::string> n1, s1, n2, s2;
  std::string buf;
  Stats se_stats, pair_stats, end1_stats, end2_stats;
  const bool paired = a.pos.size() == 2;
  constexpr std::size_t kBatch = 1000;  // src/abismal.cpp:207

  // -t n (single-end): the batch is cut into n contiguous shares mapped by n mappers; a share's mapper
  // first replays, for their side effect on its reused buffers only, the reads before the share
  // (Mapper::touch_se), so the output equals the reference at -t 1 whatever n is.
  const unsigned n_threads = std::max(1u, static_cast<unsigned>(std::stoul(a.get("t", a.get("threads", "1")))));
  if (!paired) {
    const SeMode mode = rpbat ? SE_RANDOM : ((arich || pbat) ? SE_A_RICH : SE_T_RICH);
    FastqReader rd(a.pos[0], ix.min_read_len());
    const std::size_t batch = n_threads > 1 ? 4096u * n_threads : kBatch;
    std::vector<std::unique_ptr<Mapper>> mappers;
    for (unsigned t = 0; t < n_threads; ++t) mappers.emplace_back(new Mapper(ix, par));
    std::vector<std::string> tail;  // the reads just before the batch, back to one longer than 46 bases
    std::vector<Hit> bests;
    std::vector<Cigar> cigs;
    while (rd.alive) {
      rd.load(batch, n1, s1);
      const std::size_t n = s1.size();
      bests.assign(n, Hit());
      cigs.assign(n, Cigar());
      auto share = [&](unsigned t) {
        const std::size_t lo = n * t / n_threads, hi = n * (t + 1) / n_threads;
        Mapper &mp = *mappers[t];
        if (n_threads > 1) {
          std::size_t from = lo;
          bool closed = false;
          while (from > 0 && !closed) { --from; closed = s1[from].size() > 46; }
          if (!closed) for (const std::string &r : tail) mp.touch_se(r, mode);
          for (std::size_t i = from; i < lo; ++i) mp.touch_se(s1[i], mode);
        }
        for (std::size_t i = lo; i < hi; ++i) mp.map_se(s1[i], mode, bests[i], cigs[i]);
      };
      if (n_threads == 1) share(0);
      else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_threads; ++t) th.emplace_back(share, t);
        for (auto &x : th) x.join();
      }
      buf.clear();
      for (std::size_t i = 0; i < n; ++i) {
        Hit &best = bests[i];
        if (!s1[i].empty() &&
            emit_se(buf, allow_ambig, best, ix.chroms, n1[i], s1[i], cigs[i]) == UNMAPPED)
          best.clear();
        se_stats.tally(s1[i].empty(), best, allow_ambig, cigs[i]);
      }
      out << buf;
      if (n_threads > 1) {  // what the next batch's first share has to replay
        std::vector<std::string> next;
        bool closed = false;
        for (std::size_t i = n; i-- > 0 && !closed;) { next.push_back(s1[i]); closed = s1[i].size() > 46; }
        if (!closed) for (std::size_t i = tail.size(); i-- > 0;) next.push_back(tail[i]);
        std::reverse(next.begin(), next.end());
        tail.swap(next);
      }
    }
    for (auto &mp : mappers) {
      const Work &w = mp->work;
      mapper.work.reads += w.reads; mapper.work.seed_iters += w.seed_iters; mapper.work.search_probes += w.search_probes;
      mapper.work.candidates += w.candidates; mapper.work.words += w.words; mapper.work.set_updates += w.set_updates;
      mapper.work.aligns += w.aligns; mapper.work.aligns_tb += w.aligns_tb; mapper.work.dp_cells += w.dp_cells;
    }
  }
  else {
    const PeMode mode = rpbat ? PE_RANDOM : (pbat ? PE_PBAT : PE_NORMAL);
    FastqReader rd1(a.pos[0], ix.min_read_len()), rd2(a.pos[1], ix.min_read_len());
    Cigar c1, c2;
    while (rd1.alive && rd2.alive) {
      rd1.load(kBatch, n1, s1);
      rd2.load(kBatch, n2, s2);
      if (s1.size() != s2.size())
        throw std::runtime_error("paired-end batch sizes differ. Batch 1: " + std::to_string(s1.size()) +
                                 ", batch 2: " + std::to_string(s2.size()) +
                                 ". Are you sure your paired-end inputs have the same number of reads?");
      buf.clear();
      for (std::size_t i = 0; i < s1.size(); ++i) {
        PairHit best;
        Hit h1, h2;
        c1.clear();
        c2.clear();
        mapper.map_pe(s1[i], s2[i], mode, best, h1, h2, c1, c2);
        // select_output: src/abismal.cpp:1073-1088
        const Outcome po = emit_pe(buf, allow_ambig, best, ix.chroms, n1[i], n2[i], s1[i], s2[i], c1, c2);
        if (!best.should_report(allow_ambig) || po == UNMAPPED) {
          if (po == UNMAPPED) best.clear();
          if (emit_se(buf, allow_ambig, h1, ix.chroms, n1[i], s1[i], c1) == UNMAPPED) h1.clear();
          if (emit_se(buf, allow_ambig, h2, ix.chroms, n2[i], s2[i], c2) == UNMAPPED) h2.clear();
        }
        // paired_end_mapping_statistics::update: src/abismal.cpp:1039-1057
        ++pair_stats.total;
        const bool valid = !best.empty(), amb = best.ambig();
        pair_stats.unique += valid && !amb;
        pair_stats.ambiguous += valid && amb;
        pair_stats.skipped += (s1[i].empty() || s2[i].empty());
        if (best.should_report(allow_ambig)) {
          pair_stats.edits += best.r1.diffs + best.r2.diffs;
          pair_stats.bases += cigar_ref_len(c1) + cigar_ref_len(c2);
        }
        else {
          end1_stats.tally(s1[i].empty(), h1, false, c1);
          end2_stats.tally(s2[i].empty(), h2, false, c2);
        }
      }
      out << buf;
    }
  }
  out.close();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

  const std::string stats_file = a.get("s", a.get("stats"));
  if (!stats_file.empty()) {
    std::ofstream so(stats_file);
    if (!so) std::cerr << "failed to open stats out file: " << stats_file << '\n';
    else if (a.has("j") || a.has("json")) {
      if (!paired) so << se_stats.json();
      else so << "{\"end1_stats\":" << end1_stats.json() << ",\"end2_stats\":" << end2_stats.json()
              << ",\"read_pair_stats\":" << pair_stats.json() << "}";
    }
    else if (!paired) so << se_stats.yaml("read1");
    else {
      so << pair_stats.yaml("pairs");
      if (!allow_ambig) so << end1_stats.yaml("read1") << end2_stats.yaml("read2");
    }
  }
  if (a.has("w") || a.has("work")) {  // per-read work tallies (oracle-only extension)
    const Work &w = mapper.work;
    std::ofstream wo(a.get("w", a.get("work")));
    wo << "{\"reads\":" << w.reads << ",\"seconds\":" << secs << ",\"seed_iters\":" << w.seed_iters
       << ",\"search_probes\":" << w.search_probes << ",\"candidates\":" << w.candidates
       << ",\"words\":" << w.words << ",\"set_updates\":" << w.set_updates << ",\"aligns\":" << w.aligns
       << ",\"aligns_tb\":" << w.aligns_tb << ",\"dp_cells\":" << w.dp_cells << "}\n";
  }
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  try {
    if (argc < 2) { std::cerr << "usage: abismal_oracle {idx|sim|map} ...\n"; return 1; }
    const std::string cmd = argv[1];
    if (cmd == "idx") return cmd_idx(argc - 1, argv + 1);
    if (cmd == "sim") return cmd_sim(argc - 1, argv + 1);
    if (cmd == "map") return cmd_map(argc - 1, argv + 1);
    std::cerr << "ERROR: invalid command " << cmd << '\n';
    return 1;
  }
  catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return 1;
  }
}
