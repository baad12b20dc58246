// abismal-amd: command-line host program around the C ABI.
//   abismal-amd map [flags of `abismal map`] reads_1.fq [reads_2.fq]
//   abismal-amd idx [-t n] genome.fa out.idx
// Mirrors the reference driver (src/abismal.cpp:2295-2504): same flags, same SAM
// text and statistics files, output in input order (= the reference at -t 1).
// Batches of reads go round-robin to one worker thread per GPU (index replicated
// in each GPU's HBM); results are written in batch order; the per-GPU mapping
// statistics are summed with one RCCL all-reduce at the end.
#include "../../include/abismal_amd.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr const char *kVersion = "3.3.0";  // the SAM @PG line carries the reference's version
constexpr uint32_t kMinReadLen = 44, kPadding = 32767;

[[noreturn]] void die_abm(const char *what) { throw std::runtime_error(std::string(what) + ": " + abm_last_error()); }

// ---- FASTQ, with ReadLoader's rules (src/abismal.cpp:164-201) -----------------
struct Batch {
  uint64_t seq = 0;  // batch number, defines output order
  std::vector<std::string> names[2];
  std::string blob[2];
  std::vector<uint64_t> off[2];
  // results
  std::vector<abm_hit> se[2];
  std::vector<abm_pair> pairs;
  std::vector<uint32_t> cig[2];
  std::vector<uint64_t> cig_off[2];
  std::string sam;
  size_t n() const { return names[0].size(); }
};

struct FastqReader {
  std::ifstream in;
  std::string path;
  uint64_t line_no = 0;
  bool alive = true;
  explicit FastqReader(const std::string &p) : in(p, std::ios::binary), path(p) {
    if (!in) throw std::runtime_error("cannot open reads file: " + p);
  }
  void load(size_t want, std::vector<std::string> &names, std::string &blob, std::vector<uint64_t> &off) {
    names.clear(); blob.clear(); off.assign(1, 0);
    std::string line, name;
    for (size_t k = 0; k < 4 * want; ++k, ++line_no) {
      if (!std::getline(in, line)) { alive = false; break; }
      if (k % 4 == 0) {
        if (line.empty())
          throw std::runtime_error("file " + path + " contains an empty read name at line " + std::to_string(line_no));
        name = line.substr(1, line.find_first_of(" \t") - 1);
      }
      else if (k % 4 == 1) {
        if (line.size() >= kPadding)
          throw std::runtime_error("found a read of size " + std::to_string(line.size()) +
                                   ", which is too long. Maximum allowed read size = " + std::to_string(kPadding));
        const auto informative = std::count_if(line.begin(), line.end(), [](char c) { return c != 'N'; });
        if (informative < static_cast<std::ptrdiff_t>(kMinReadLen)) line.clear();
        else {
          while (line.back() == 'N') line.pop_back();
          line = line.substr(line.find_first_of("ACGT"));
        }
        names.push_back(name);
        blob += line;
        off.push_back(blob.size());
      }
    }
  }
};

// ---- SAM text (format_se / format_pe, src/abismal.cpp:481-545, :648-773) -------
struct Chroms {
  std::vector<std::string> names;
  std::vector<uint32_t> starts;
  bool locate(uint32_t pos, uint32_t reflen, int32_t &chrom, uint32_t &off) const {
    auto it = std::upper_bound(starts.begin(), starts.end(), pos);
    if (it == starts.begin()) return false;
    --it;
    chrom = static_cast<int32_t>(it - starts.begin());
    off = pos - starts[chrom];
    return pos + reflen <= starts[chrom + 1];
  }
};

uint32_t ref_len(const uint32_t *c, size_t n) {
  uint32_t r = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t op = c[i] & 15u;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) r += c[i] >> 4;
  }
  return r;
}

void append_revcomp(std::string &o, const char *s, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const char c = s[n - 1 - i];
    o += c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
  }
}
void append_seq(std::string &o, const char *s, size_t n) {  // htslib's 4-bit round trip
  static const char ok[] = "=ACMGRSVTWYHKDBN";
  for (size_t i = 0; i < n; ++i) {
    const char u = static_cast<char>(std::toupper(static_cast<unsigned char>(s[i])));
    o += (u && std::strchr(ok, u)) ? u : 'N';
  }
}

struct Record {
  const std::string *name;
  uint16_t flag;
  int32_t tid, mtid;
  uint32_t pos, mpos;
  int tlen;
  const uint32_t *cig;
  size_t n_cig;
  const char *seq;
  size_t n_seq;
  bool rc;
  int nm;
  char cv;
};

void put_record(std::string &o, const Chroms &ch, const Record &r) {
  o += *r.name; o += '\t'; o += std::to_string(r.flag); o += '\t';
  o += ch.names[r.tid + 1]; o += '\t'; o += std::to_string(r.pos + 1); o += "\t255\t";
  for (size_t i = 0; i < r.n_cig; ++i) { o += std::to_string(r.cig[i] >> 4); o += "MIDNSHP=XB"[std::min<uint32_t>(r.cig[i] & 15u, 9)]; }
  o += '\t';
  if (r.mtid < 0) o += "*\t0\t";
  else { o += (r.mtid == r.tid) ? std::string("=") : ch.names[r.mtid + 1]; o += '\t'; o += std::to_string(r.mpos + 1); o += '\t'; }
  o += std::to_string(r.tlen); o += '\t';
  if (r.rc) { std::string t; append_revcomp(t, r.seq, r.n_seq); append_seq(o, t.data(), t.size()); }
  else append_seq(o, r.seq, r.n_seq);
  o += "\t*\tNM:i:"; o += std::to_string(r.nm); o += "\tCV:A:"; o += r.cv; o += '\n';
}

enum Outcome { UNMAPPED, UNIQUE, AMBIG };

Outcome emit_se(std::string &o, bool allow_ambig, const abm_hit &h, const Chroms &ch, const std::string &name,
                const char *seq, size_t n_seq, const uint32_t *cig, size_t n_cig) {
  const bool ambig = h.flags & 0x100;
  if (!allow_ambig && ambig) return AMBIG;
  uint32_t off = 0; int32_t chrom = 0;
  if (h.pos == 0 || !ch.locate(h.pos, ref_len(cig, n_cig), chrom, off)) return UNMAPPED;
  Record r{&name, 0, chrom - 1, -1, off, 0, 0, cig, n_cig, seq, n_seq, (h.flags & 0x10) != 0, h.diffs,
           (h.flags & 0x1000) ? 'A' : 'T'};
  if (h.flags & 0x10) r.flag |= 0x10;
  if (allow_ambig && ambig) r.flag |= 0x100;
  put_record(o, ch, r);
  return ambig ? AMBIG : UNIQUE;
}

Outcome emit_pe(std::string &o, bool allow_ambig, const abm_pair &p, const Chroms &ch, const std::string &n1,
                const std::string &n2, const char *s1, size_t l1, const char *s2, size_t l2, const uint32_t *c1,
                size_t nc1, const uint32_t *c2, size_t nc2) {
  if (p.r1.pos == 0) return UNMAPPED;
  const bool ambig = p.r1.flags & 0x100;
  if (!allow_ambig && ambig) return AMBIG;
  int32_t ch1 = 0, ch2 = 0; uint32_t b1 = 0, b2 = 0;
  const uint32_t rl1 = ref_len(c1, nc1), rl2 = ref_len(c2, nc2);
  if (!ch.locate(p.r1.pos, rl1, ch1, b1) || !ch.locate(p.r2.pos, rl2, ch2, b2) || ch1 != ch2) return UNMAPPED;
  const uint32_t e2 = b2 + rl2;
  const bool rc1 = p.r1.flags & 0x10, rc2 = p.r2.flags & 0x10;
  const int isize = rc1 ? static_cast<int>(b1) - static_cast<int>(e2) : static_cast<int>(e2) - static_cast<int>(b1);
  uint16_t f1 = 0x1 | 0x2 | 0x40, f2 = 0x1 | 0x2 | 0x80;
  if (rc1) { f1 |= 0x10; f2 |= 0x20; }
  if (rc2) { f2 |= 0x10; f1 |= 0x20; }
  if (allow_ambig && ambig) { f1 |= 0x100; f2 |= 0x100; }
  put_record(o, ch, Record{&n1, f1, ch1 - 1, ch2 - 1, b1, b2, isize, c1, nc1, s1, l1, rc1, p.r1.diffs, (p.r1.flags & 0x1000) ? 'A' : 'T'});
  put_record(o, ch, Record{&n2, f2, ch2 - 1, ch1 - 1, b2, b1, -isize, c2, nc2, s2, l2, rc2, p.r2.diffs, (p.r2.flags & 0x1000) ? 'A' : 'T'});
  return ambig ? AMBIG : UNIQUE;
}

// ---- statistics (src/abismal.cpp:865-1071); 6 counters x {pairs|se, read1, read2} ----
struct Stats {
  uint64_t v[6] = {0, 0, 0, 0, 0, 0};  // total, unique, ambiguous, skipped, edits, bases
  void tally(bool empty_read, const abm_hit &h, bool count_ambig_error, uint32_t bases) {
    ++v[0];
    const bool valid = h.pos != 0, amb = h.flags & 0x100;
    v[1] += valid && !amb; v[2] += valid && amb; v[3] += empty_read;
    if (valid && (!amb || count_ambig_error)) { v[4] += static_cast<uint64_t>(static_cast<int64_t>(h.diffs)); v[5] += bases; }
  }
  std::string yaml(const std::string &label) const {
    // the reference keeps the first four in 32-bit counters (they wrap there)
    const uint32_t total = static_cast<uint32_t>(v[0]), unique = static_cast<uint32_t>(v[1]),
                   ambiguous = static_cast<uint32_t>(v[2]), skipped = static_cast<uint32_t>(v[3]);
    auto frac = [&](double x) { return total > 0 ? x / total : 0.0; };
    const uint32_t mapped = unique + ambiguous, unmapped = total - mapped;
    std::ostringstream s; const char *t = "    ";
    s << label << ":\n" << t << "total_reads: " << total << '\n' << t << "mapped:\n"
      << t << "    num_mapped: " << mapped << '\n' << t << "    num_unique: " << unique << '\n'
      << t << "    num_ambiguous: " << ambiguous << '\n' << t << "    percent_mapped: " << frac(mapped) * 100.0 << '\n'
      << t << "    percent_unique: " << frac(unique) * 100.0 << '\n' << t << "    percent_ambiguous: " << frac(ambiguous) * 100.0 << '\n'
      << t << "    unique_error:\n" << t << "        edits: " << v[4] << '\n' << t << "        total_bases: " << v[5] << '\n'
      << t << "        error_rate: " << (v[5] > 0 ? static_cast<double>(v[4]) / v[5] : 0.0) << '\n'
      << t << "num_unmapped: " << unmapped << '\n' << t << "num_skipped: " << skipped << '\n'
      << t << "percent_unmapped: " << frac(unmapped) * 100.0 << '\n' << t << "percent_skipped: " << frac(skipped) * 100.0 << '\n';
    return s.str();
  }
  std::string json() const {
    std::ostringstream s;
    s << "{\"edit_distance\":" << v[4] << ",\"reads_mapped_ambiguous\":" << static_cast<uint32_t>(v[2])
      << ",\"reads_mapped_unique\":" << static_cast<uint32_t>(v[1]) << ",\"reads_skipped\":" << static_cast<uint32_t>(v[3])
      << ",\"total_bases\":" << v[5] << ",\"total_reads\":" << static_cast<uint32_t>(v[0]) << "}";
    return s.str();
  }
};
struct Stats3 { Stats s[3]; };  // SE: s[0]; PE: pairs, read1, read2

struct Options {
  std::string index, genome, out, stats;
  bool bam = false, json = false, ambig = false, pbat = false, rpbat = false, arich = false, verbose = false;
  uint32_t max_candidates = 0, min_frag = 32, max_frag = 3000, threads = 1;
  int gpus = 0;
  size_t batch = 1u << 20;
  double max_distance = 0.1;
  std::vector<std::string> reads;
};

Options parse_map(int argc, char **argv) {
  Options o;
  auto need = [&](int &i) -> std::string { if (i + 1 >= argc) throw std::runtime_error(std::string("missing value for ") + argv[i]); return argv[++i]; };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') { o.reads.push_back(a); continue; }
    const std::string k = a.substr(a.find_first_not_of('-'));
    if (k == "i" || k == "index") o.index = need(i);
    else if (k == "g" || k == "genome") o.genome = need(i);
    else if (k == "o" || k == "outfile") o.out = need(i);
    else if (k == "B" || k == "bam") o.bam = true;
    else if (k == "s" || k == "stats") o.stats = need(i);
    else if (k == "j" || k == "json") o.json = true;
    else if (k == "c" || k == "max-candidates") o.max_candidates = static_cast<uint32_t>(std::stoul(need(i)));
    else if (k == "l" || k == "min-frag") o.min_frag = static_cast<uint32_t>(std::stoul(need(i)));
    else if (k == "L" || k == "max-frag") o.max_frag = static_cast<uint32_t>(std::stoul(need(i)));
    else if (k == "m" || k == "max-distance") o.max_distance = std::stod(need(i));
    else if (k == "a" || k == "ambig") o.ambig = true;
    else if (k == "P" || k == "pbat") o.pbat = true;
    else if (k == "R" || k == "random-pbat") o.rpbat = true;
    else if (k == "A" || k == "a-rich") o.arich = true;
    else if (k == "t" || k == "threads") o.threads = static_cast<uint32_t>(std::stoul(need(i)));
    else if (k == "v" || k == "verbose") o.verbose = true;
    else if (k == "gpus") o.gpus = std::stoi(need(i));
    else if (k == "batch") o.batch = std::stoul(need(i));
    else throw std::runtime_error("unknown option " + a);
  }
  return o;
}

int cmd_idx(int argc, char **argv) {
  unsigned threads = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if ((a == "-t" || a == "-threads" || a == "--threads") && i + 1 < argc) threads = static_cast<unsigned>(std::stoul(argv[++i]));
    else if (a == "-v" || a == "-verbose") {}
    else pos.push_back(a);
  }
  if (pos.size() != 2) { std::cerr << "usage: abismal-amd idx [-t n] <genome.fa> <out.idx>\n"; return EXIT_SUCCESS; }
  if (abm_index_build(pos[0].c_str(), pos[1].c_str(), threads) != 0) die_abm("idx");
  return EXIT_SUCCESS;
}

int cmd_map(int argc, char **argv) {
  const Options opt = parse_map(argc, argv);
  if (opt.out.empty()) { std::cerr << "Missing required argument\n-o, -outfile\n"; return EXIT_SUCCESS; }
  if (opt.reads.size() != 1 && opt.reads.size() != 2) { std::cerr << "usage: abismal-amd map -i idx -o out.sam [flags] reads_1.fq [reads_2.fq]\n"; return EXIT_SUCCESS; }
  if (opt.index.empty() == opt.genome.empty()) { std::cerr << "Select one of index file (-i) or genome file (-g)\n"; return EXIT_SUCCESS; }
  if (opt.bam) throw std::runtime_error("BAM output (-B) is not available in this build; write SAM and convert");
  const bool paired = opt.reads.size() == 2;

  std::string index_path = opt.index;
  if (index_path.empty()) {  // -g: index the genome on the fly (src/abismal.cpp:2439-2446)
    index_path = opt.out + ".tmp.idx";
    if (abm_index_build(opt.genome.c_str(), index_path.c_str(), std::max(1u, std::thread::hardware_concurrency())) != 0) die_abm("indexing genome");
  }
  abm_index *ix = nullptr;
  if (abm_index_open(index_path.c_str(), &ix) != 0) die_abm("loading index");
  if (opt.index.empty()) std::remove(index_path.c_str());
  Chroms ch;
  for (uint32_t i = 0; i < abm_index_n_chroms(ix); ++i) ch.names.push_back(abm_index_chrom_name(ix, i));
  ch.starts.assign(abm_index_chrom_starts(ix), abm_index_chrom_starts(ix) + ch.names.size() + 1);

  // one context (= one replica of the index in HBM) per GPU
  int n_gpus = opt.gpus;
  std::vector<abm_ctx *> ctxs;
  for (int d = 0; n_gpus <= 0 || d < n_gpus; ++d) {
    abm_ctx *c = nullptr;
    if (abm_ctx_create(ix, d, &c) != 0) { if (n_gpus <= 0 && d > 0) break; die_abm("creating GPU context"); }
    ctxs.push_back(c);
  }
  n_gpus = static_cast<int>(ctxs.size());

  std::ofstream out(opt.out, std::ios::binary);
  if (!out) throw std::runtime_error("failed to open output file: " + opt.out);
  {  // header, src/abismal.cpp:2265-2293
    std::ostringstream h;
    h << "@HD\tVN:1.0\n";
    for (size_t i = 1; i + 1 < ch.names.size(); ++i) h << "@SQ\tSN:" << ch.names[i] << "\tLN:" << (ch.starts[i + 1] - ch.starts[i]) << '\n';
    h << "@PG\tID:ABISMAL\tVN:" << kVersion << "\tCL:\"";
    for (int i = 0; i < argc; ++i) h << argv[i] << ' ';
    h << "\"\n";
    out << h.str();
  }

  abm_params par;
  abm_default_params(&par);
  par.max_candidates = opt.max_candidates;
  par.valid_frac = opt.max_distance;
  par.min_frag = opt.min_frag;
  par.max_frag = opt.max_frag;
  par.allow_ambig = opt.ambig;
  const int se_mode = opt.rpbat ? ABM_SE_RANDOM : ((opt.arich || opt.pbat) ? ABM_SE_A_RICH : ABM_SE_T_RICH);
  const int pe_mode = opt.rpbat ? ABM_PE_RANDOM : (opt.pbat ? ABM_PE_PBAT : ABM_PE_NORMAL);

  // reader -> per-GPU workers -> in-order writer
  std::mutex mu;
  std::condition_variable cv;
  std::map<uint64_t, std::unique_ptr<Batch>> done;
  uint64_t next_to_write = 0, next_to_read = 0;
  bool reading_finished = false;
  std::exception_ptr failure;
  FastqReader rd1(opt.reads[0]);
  std::unique_ptr<FastqReader> rd2;
  if (paired) rd2.reset(new FastqReader(opt.reads[1]));
  std::vector<Stats3> gpu_stats(n_gpus);
  const auto t_start = std::chrono::steady_clock::now();

  auto worker = [&](int g) {
    try {
      for (;;) {
        std::unique_ptr<Batch> b(new Batch);
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return failure || reading_finished || done.size() < static_cast<size_t>(4 * n_gpus); });
          if (failure || reading_finished) return;
          b->seq = next_to_read++;
          rd1.load(opt.batch, b->names[0], b->blob[0], b->off[0]);
          if (paired) {
            rd2->load(opt.batch, b->names[1], b->blob[1], b->off[1]);
            if (b->names[0].size() != b->names[1].size())
              throw std::runtime_error("paired-end batch sizes differ. Batch 1: " + std::to_string(b->names[0].size()) +
                                       ", batch 2: " + std::to_string(b->names[1].size()) +
                                       ". Are you sure your paired-end inputs have the same number of reads?");
          }
          if (!rd1.alive || (paired && !rd2->alive)) reading_finished = true;
        }
        const size_t n = b->n();
        Stats3 &st = gpu_stats[g];
        if (n) {
          const uint64_t cap = std::max<uint64_t>(1, std::max(b->blob[0].size(), b->blob[1].size()) + 2 * n);
          if (!paired) {
            b->se[0].resize(n); b->cig[0].resize(cap); b->cig_off[0].resize(n + 1);
            if (abm_map_se_batch(ctxs[g], se_mode, &par, n, b->blob[0].data(), b->off[0].data(), b->se[0].data(),
                                 b->cig[0].data(), cap, b->cig_off[0].data()) != 0) die_abm("mapping");
            for (size_t i = 0; i < n; ++i) {
              abm_hit h = b->se[0][i];
              const size_t len = b->off[0][i + 1] - b->off[0][i];
              const uint32_t *cg = b->cig[0].data() + b->cig_off[0][i];
              const size_t ncg = b->cig_off[0][i + 1] - b->cig_off[0][i];
              if (len && emit_se(b->sam, opt.ambig, h, ch, b->names[0][i], b->blob[0].data() + b->off[0][i], len, cg, ncg) == UNMAPPED) h.pos = 0;
              st.s[0].tally(len == 0, h, opt.ambig, ref_len(cg, ncg));
            }
          }
          else {
            b->pairs.resize(n); b->se[0].resize(n); b->se[1].resize(n);
            for (int e = 0; e < 2; ++e) { b->cig[e].resize(cap); b->cig_off[e].resize(n + 1); }
            if (abm_map_pe_batch(ctxs[g], pe_mode, &par, n, b->blob[0].data(), b->off[0].data(), b->blob[1].data(),
                                 b->off[1].data(), b->pairs.data(), b->se[0].data(), b->se[1].data(), b->cig[0].data(),
                                 b->cig_off[0].data(), b->cig[1].data(), b->cig_off[1].data(), cap) != 0) die_abm("mapping");
            for (size_t i = 0; i < n; ++i) {
              abm_pair p = b->pairs[i];
              abm_hit h1 = b->se[0][i], h2 = b->se[1][i];
              const char *s1 = b->blob[0].data() + b->off[0][i], *s2 = b->blob[1].data() + b->off[1][i];
              const size_t l1 = b->off[0][i + 1] - b->off[0][i], l2 = b->off[1][i + 1] - b->off[1][i];
              const uint32_t *c1 = b->cig[0].data() + b->cig_off[0][i], *c2 = b->cig[1].data() + b->cig_off[1][i];
              const size_t nc1 = b->cig_off[0][i + 1] - b->cig_off[0][i], nc2 = b->cig_off[1][i + 1] - b->cig_off[1][i];
              // select_output, src/abismal.cpp:1073-1088
              const Outcome po = emit_pe(b->sam, opt.ambig, p, ch, b->names[0][i], b->names[1][i], s1, l1, s2, l2, c1, nc1, c2, nc2);
              const bool report = p.r1.pos != 0 && (opt.ambig || !(p.r1.flags & 0x100));
              bool pair_ok = report;
              if (!report || po == UNMAPPED) {
                if (po == UNMAPPED) { p.r1.pos = 0; p.r2.pos = 0; pair_ok = false; }
                if (emit_se(b->sam, opt.ambig, h1, ch, b->names[0][i], s1, l1, c1, nc1) == UNMAPPED) h1.pos = 0;
                if (emit_se(b->sam, opt.ambig, h2, ch, b->names[1][i], s2, l2, c2, nc2) == UNMAPPED) h2.pos = 0;
              }
              // paired_end_mapping_statistics::update, :1039-1057
              Stats &ps = st.s[0];
              ++ps.v[0];
              const bool valid = p.r1.pos != 0, amb = p.r1.flags & 0x100;
              ps.v[1] += valid && !amb; ps.v[2] += valid && amb; ps.v[3] += (l1 == 0 || l2 == 0);
              if (pair_ok && valid) { ps.v[4] += static_cast<uint64_t>(static_cast<int64_t>(p.r1.diffs) + p.r2.diffs); ps.v[5] += ref_len(c1, nc1) + ref_len(c2, nc2); }
              else {
                st.s[1].tally(l1 == 0, h1, false, ref_len(c1, nc1));
                st.s[2].tally(l2 == 0, h2, false, ref_len(c2, nc2));
              }
            }
          }
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          done[b->seq] = std::move(b);
        }
        cv.notify_all();
      }
    }
    catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failure) failure = std::current_exception();
      cv.notify_all();
    }
  };

  std::vector<std::thread> workers;
  for (int g = 0; g < n_gpus; ++g) workers.emplace_back(worker, g);
  uint64_t total_records = 0;
  {  // writer: batches leave in input order
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return failure || done.count(next_to_write) || (reading_finished && next_to_write == next_to_read); });
      if (failure) break;
      auto it = done.find(next_to_write);
      if (it == done.end()) {
        if (reading_finished && next_to_write == next_to_read) break;
        continue;
      }
      std::unique_ptr<Batch> b = std::move(it->second);
      done.erase(it);
      ++next_to_write;
      lk.unlock();
      out << b->sam;
      total_records += b->n();
      cv.notify_all();
      lk.lock();
    }
  }
  cv.notify_all();
  for (auto &t : workers) t.join();
  if (failure) std::rethrow_exception(failure);
  out.close();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

  // statistics: one RCCL all-reduce over the GPUs that took part
  std::vector<uint64_t *> ptrs;
  for (auto &s : gpu_stats) ptrs.push_back(&s.s[0].v[0]);
  static_assert(sizeof(Stats3) == 18 * sizeof(uint64_t), "18 counters");
  if (abm_stats_allreduce(ctxs.data(), n_gpus, ptrs.data()) != 0) die_abm("stats all-reduce");
  const Stats3 &tot = gpu_stats[0];
  if (!opt.stats.empty()) {
    std::ofstream so(opt.stats);
    if (!so) std::cerr << "failed to open stats out file: " << opt.stats << '\n';
    else if (opt.json) {
      if (!paired) so << tot.s[0].json();
      else so << "{\"end1_stats\":" << tot.s[1].json() << ",\"end2_stats\":" << tot.s[2].json() << ",\"read_pair_stats\":" << tot.s[0].json() << "}";
    }
    else if (!paired) so << tot.s[0].yaml("read1");
    else { so << tot.s[0].yaml("pairs"); if (!opt.ambig) so << tot.s[1].yaml("read1") << tot.s[2].yaml("read2"); }
  }
  if (opt.verbose)
    std::cerr << "[abismal-amd] " << total_records << (paired ? " pairs" : " reads") << " on " << n_gpus << " GPU(s) in "
              << secs << " s (" << (paired ? 2 : 1) * total_records / secs << " reads/s incl. host I/O)\n";
  for (abm_ctx *c : ctxs) abm_ctx_destroy(c);
  abm_index_close(ix);
  return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char **argv) {
  try {
    if (argc < 2) { std::cout << "Program: abismal-amd\nVersion: " << kVersion << "\nUsage: abismal-amd <command> [options]\nCommands:\n    map:    map FASTQ reads to an index or a FASTA reference genome\n    idx:    make an index for a FASTA reference genome\n"; return EXIT_SUCCESS; }
    const std::string cmd = argv[1];
    if (cmd == "map") return cmd_map(argc - 1, argv + 1);
    if (cmd == "idx") return cmd_idx(argc - 1, argv + 1);
    std::cerr << "ERROR: invalid command " << cmd << '\n';
    return EXIT_SUCCESS;
  }
  catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return EXIT_FAILURE;
  }
}
