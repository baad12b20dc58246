// abismal-amd: command-line host program around the C ABI.
//   abismal-amd map [flags of `abismal map`] reads_1.fq [reads_2.fq]
//   abismal-amd idx [-t n] genome.fa out.idx
// Mirrors the reference driver (src/abismal.cpp:2295-2504): same flags, same SAM
// text and statistics files, output in input order (= the reference at -t 1).
// Batches of reads go round-robin to one worker thread per GPU (index replicated
// in each GPU's HBM); results are written in batch order; the per-GPU mapping
// statistics are summed with one RCCL all-reduce at the end.
#include "../../include/abismal_amd.h"

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
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

namespace abm { int sim_main(int argc, char **argv); }

namespace {

constexpr const char *kVersion = "3.3.0";  // the SAM @PG line carries the reference's version
constexpr uint32_t kMinReadLen = 44, kPadding = 32767;

[[noreturn]] void die_abm(const char *what) { throw std::runtime_error(std::string(what) + ": " + abm_last_error()); }

// ---- FASTQ, with ReadLoader's rules (src/abismal.cpp:164-201) -----------------
// Stage 1 (one thread per input file) only cuts the file into batches of whole records;
// stage 2 (a pool) applies the reference's per-record rules and lays the reads out for the C ABI.
struct Stats { 
  uint64_t v[6] = {0, 0, 0, 0, 0, 0};  // total, unique, ambiguous, skipped, edits, bases
  void tally(bool empty_read, const abm_hit &h, bool count_ambig_error, uint32_t bases);
  std::string yaml(const std::string &label) const;
  std::string json() const;
};
struct Stats3 { Stats s[3]; };  // SE: s[0]; PE: pairs, read1, read2

// A batch's FASTQ text: grown with realloc (large blocks are remapped, not copied, and never
// zero-filled) and recycled through a small pool so that its pages stay faulted in.
struct RawBuf {
  char *p = nullptr;
  size_t n = 0, cap = 0;
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  RawBuf(RawBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
  RawBuf &operator=(RawBuf &&o) noexcept { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); return *this; }
  ~RawBuf() { std::free(p); }
  void reserve(size_t want) {
    if (want <= cap) return;
    char *q = static_cast<char *>(std::realloc(p, want));
    if (!q) throw std::bad_alloc();
    p = q; cap = want;
  }
  void append(const char *src, size_t len) { reserve(n + len); std::memcpy(p + n, src, len); n += len; }
};
struct RawPool {
  std::mutex mu;
  std::vector<RawBuf> free_list;
  RawBuf get() {
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.empty()) return RawBuf();
    RawBuf b = std::move(free_list.back());
    free_list.pop_back();
    b.n = 0;
    return b;
  }
  void put(RawBuf &&b) {
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.size() < 24) free_list.push_back(std::move(b));
  }
};

struct Batch {
  uint64_t seq = 0;        // batch number, defines output order
  uint64_t first_line[2] = {0, 0};
  RawBuf raw[2];           // the FASTQ text of this batch
  std::vector<std::string> names[2];
  std::string blob[2];
  std::vector<uint64_t> off[2];
  // results
  int gpu = 0;
  std::vector<abm_hit> se[2];
  std::vector<abm_pair> pairs;
  std::vector<uint32_t> cig[2];
  std::vector<uint64_t> cig_off[2];
  // formatted output: slices of the batch are formatted independently and written in order
  std::vector<std::string> parts;
  std::vector<Stats3> part_stats;
  int parts_left = 0;
  Stats3 stats;
  size_t n() const { return names[0].size(); }
};

// advances over [p + from, p + len) counting newlines until `need` lines are complete; returns the
// offset just past the last newline consumed (block counts vectorise; only the block in which the
// target falls is walked line by line)
size_t scan_lines(const char *p, size_t from, size_t len, uint64_t need, uint64_t &lines) {
  size_t i = from, last = from;
  while (i < len && lines < need) {
    const size_t blk = std::min<size_t>(len - i, 8192);
    uint32_t c = 0;
    for (size_t k = 0; k < blk; ++k) c += (p[i + k] == '\n');
    if (lines + c < need) {
      if (c) last = static_cast<size_t>(static_cast<const char *>(memrchr(p + i, '\n', blk)) - p) + 1;
      lines += c;
      i += blk;
      continue;
    }
    while (lines < need) {
      const char *nl = static_cast<const char *>(std::memchr(p + i, '\n', len - i));
      i = static_cast<size_t>(nl - p) + 1;
      ++lines;
    }
    return i;
  }
  return last;
}

struct RawSplitter {
  gzFile f = nullptr;  // gzip/bgzip-compressed FASTQ goes through zlib (bamxx::bgzf_file in the reference)
  int fd = -1;         // plain text is read directly
  std::string path, carry;  // carry: text read past the end of the previous batch
  uint64_t line_no = 0;
  bool eof = false;
  explicit RawSplitter(const std::string &p) : path(p) {
    fd = ::open(p.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open reads file: " + p);
    unsigned char magic[2] = {0, 0};
    const ssize_t got = ::pread(fd, magic, 2, 0);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
      ::close(fd);
      fd = -1;
      f = gzopen(p.c_str(), "rb");
      if (!f) throw std::runtime_error("cannot open reads file: " + p);
      gzbuffer(f, 1u << 20);
    }
  }
  ~RawSplitter() { if (f) gzclose(f); if (fd >= 0) ::close(fd); }
  size_t fill(char *dst, size_t want) {
    size_t have = 0;
    while (have < want) {
      long got;
      if (f) got = gzread(f, dst + have, static_cast<unsigned>(std::min<size_t>(want - have, 1u << 30)));
      else got = static_cast<long>(::read(fd, dst + have, want - have));
      if (got < 0) throw std::runtime_error("error reading " + path);
      if (got == 0) break;
      have += static_cast<size_t>(got);
    }
    return have;
  }
  // up to `want` records (4 lines each) of text; returns the number of complete lines delivered
  uint64_t next(size_t want, RawBuf &out, uint64_t &first_line) {
    first_line = line_no;
    out.n = 0;
    out.append(carry.data(), carry.size());
    carry.clear();
    const uint64_t need = 4 * static_cast<uint64_t>(want);
    uint64_t lines = 0;
    size_t scanned = scan_lines(out.p, 0, out.n, need, lines);  // meaningful once lines == need
    while (lines < need && !eof) {
      const size_t old = out.n, chunk = 32u << 20;
      out.reserve(std::max(old + chunk, last_size + chunk));
      const size_t got = fill(out.p + old, chunk);
      out.n = old + got;
      if (got < chunk) eof = true;
      scanned = scan_lines(out.p, old, out.n, need, lines);
    }
    if (lines == need) { carry.assign(out.p + scanned, out.n - scanned); out.n = scanned; }
    else if (out.n && out.p[out.n - 1] != '\n') ++lines;  // a last line without a newline still counts (getline semantics)
    last_size = out.n;
    line_no += lines;
    return lines;
  }
  size_t last_size = 0;
  bool exhausted() const { return eof && carry.empty(); }
};

void parse_raw(const RawBuf &raw, uint64_t first_line, const std::string &path, std::vector<std::string> &names,
               std::string &blob, std::vector<uint64_t> &off) {
  names.clear(); blob.clear(); off.assign(1, 0);
  blob.reserve(raw.n / 2);
  const char *p = raw.p, *end = p + raw.n;
  std::string line;
  for (uint64_t k = 0; p < end; ++k) {
    const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    const char *le = nl ? nl : end;
    if (k % 4 == 0) {
      if (le == p)
        throw std::runtime_error("file " + path + " contains an empty read name at line " + std::to_string(first_line + k));
      const char *q = p + 1;
      while (q < le && *q != ' ' && *q != '\t') ++q;
      names.emplace_back(p + 1, q);
    }
    else if (k % 4 == 1) {
      const size_t len = static_cast<size_t>(le - p);
      if (len >= kPadding)
        throw std::runtime_error("found a read of size " + std::to_string(len) +
                                 ", which is too long. Maximum allowed read size = " + std::to_string(kPadding));
      size_t informative = 0;
      for (const char *c = p; c < le; ++c) informative += (*c != 'N');
      if (informative >= kMinReadLen) {
        const char *e = le;
        while (e > p && e[-1] == 'N') --e;                       // remove Ns from 3'
        const char *b = p;
        while (b < e && *b != 'A' && *b != 'C' && *b != 'G' && *b != 'T') ++b;  // ... and everything before the first base
        if (b == e) throw std::runtime_error("read without A/C/G/T at line " + std::to_string(first_line + k));
        blob.append(b, e);
      }
      off.push_back(blob.size());
    }
    if (!nl) break;
    p = nl + 1;
  }
  names.resize(off.size() - 1);  // a trailing name line without its sequence is not a record
}

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

// SEQ as htslib prints it after its 4-bit round trip: IUPAC upper-cased, everything else N;
// the reverse-strand variant complements first (src/common.hpp:28-44: non-ACGT -> N)
struct SeqTables {
  char fwd[256], rc[256];
  SeqTables() {
    static const char ok[] = "=ACMGRSVTWYHKDBN";
    for (int c = 0; c < 256; ++c) {
      const char u = static_cast<char>(std::toupper(c));
      fwd[c] = (u && std::strchr(ok, u)) ? u : 'N';
      rc[c] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
    }
  }
};
const SeqTables kSeq;

inline void put_uint(std::string &o, uint64_t v) {
  char buf[24];
  int k = 24;
  do { buf[--k] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
  o.append(buf + k, static_cast<size_t>(24 - k));
}
inline void put_int(std::string &o, int64_t v) {
  if (v < 0) { o += '-'; put_uint(o, static_cast<uint64_t>(-v)); }
  else put_uint(o, static_cast<uint64_t>(v));
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

void put_bam_record(std::string &o, const Record &r);
thread_local bool t_bam = false;  // formatter threads switch put_record to BAM encoding
void put_record(std::string &o, const Chroms &ch, const Record &r) {
  if (t_bam) { put_bam_record(o, r); return; }
  o += *r.name; o += '\t'; put_uint(o, r.flag); o += '\t';
  o += ch.names[r.tid + 1]; o += '\t'; put_uint(o, static_cast<uint64_t>(r.pos) + 1); o.append("\t255\t", 5);
  for (size_t i = 0; i < r.n_cig; ++i) { put_uint(o, r.cig[i] >> 4); o += "MIDNSHP=XB"[std::min<uint32_t>(r.cig[i] & 15u, 9)]; }
  o += '\t';
  if (r.mtid < 0) o.append("*\t0\t", 4);
  else {
    if (r.mtid == r.tid) o += '='; else o += ch.names[r.mtid + 1];
    o += '\t'; put_uint(o, static_cast<uint64_t>(r.mpos) + 1); o += '\t';
  }
  put_int(o, r.tlen); o += '\t';
  const size_t at = o.size();
  o.resize(at + r.n_seq);
  char *dst = &o[at];
  if (r.rc) for (size_t i = 0; i < r.n_seq; ++i) dst[i] = kSeq.rc[static_cast<unsigned char>(r.seq[r.n_seq - 1 - i])];
  else for (size_t i = 0; i < r.n_seq; ++i) dst[i] = kSeq.fwd[static_cast<unsigned char>(r.seq[i])];
  o.append("\t*\tNM:i:", 8); put_int(o, r.nm); o.append("\tCV:A:", 6); o += r.cv; o += '\n';
}

// ---- BAM (-B): the same records as binary BAM in BGZF blocks (SAM spec 4.2 / 4.1) ------------------
// htslib's bam_set1 + bam_aux_update_int("NM") + bam_aux_append("CV",'A') in the reference
// (src/abismal.cpp:513-543); quality is absent (0xFF), MAPQ 255.
inline void put_le32(std::string &o, uint32_t v) { char b[4] = {static_cast<char>(v), static_cast<char>(v >> 8), static_cast<char>(v >> 16), static_cast<char>(v >> 24)}; o.append(b, 4); }
inline void put_le16(std::string &o, uint16_t v) { char b[2] = {static_cast<char>(v), static_cast<char>(v >> 8)}; o.append(b, 2); }
inline int reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return static_cast<int>(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return static_cast<int>(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return static_cast<int>(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return static_cast<int>(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return static_cast<int>(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
void put_bam_record(std::string &o, const Record &r) {
  static const char nt16[] = "=ACMGRSVTWYHKDBN";
  const size_t start = o.size();
  put_le32(o, 0);  // block_size, patched below
  put_le32(o, static_cast<uint32_t>(r.tid));
  put_le32(o, r.pos);
  const uint32_t rl = ref_len(r.cig, r.n_cig);
  o += static_cast<char>(r.name->size() + 1);
  o += static_cast<char>(255);
  put_le16(o, static_cast<uint16_t>(reg2bin(r.pos, static_cast<int64_t>(r.pos) + (rl ? rl : 1))));
  put_le16(o, static_cast<uint16_t>(r.n_cig));
  put_le16(o, r.flag);
  put_le32(o, static_cast<uint32_t>(r.n_seq));
  put_le32(o, static_cast<uint32_t>(r.mtid));
  put_le32(o, r.mtid < 0 ? 0xFFFFFFFFu : r.mpos);
  put_le32(o, static_cast<uint32_t>(r.tlen));
  o.append(*r.name); o += '\0';
  for (size_t i = 0; i < r.n_cig; ++i) put_le32(o, r.cig[i]);
  auto code = [&](size_t i) -> int {
    const char c = r.rc ? kSeq.rc[static_cast<unsigned char>(r.seq[r.n_seq - 1 - i])] : kSeq.fwd[static_cast<unsigned char>(r.seq[i])];
    const char *q = std::strchr(nt16, c);
    return q ? static_cast<int>(q - nt16) : 15;
  };
  for (size_t i = 0; i < r.n_seq; i += 2)
    o += static_cast<char>((code(i) << 4) | (i + 1 < r.n_seq ? code(i + 1) : 0));
  o.append(r.n_seq, static_cast<char>(0xFF));
  o.append("NM", 2);  // bam_aux_update_int: smallest type that holds the value
  if (r.nm >= 0 && r.nm <= 255) { o += 'C'; o += static_cast<char>(r.nm); }
  else if (r.nm >= 0) { o += 'S'; put_le16(o, static_cast<uint16_t>(r.nm)); }
  else if (r.nm >= -128) { o += 'c'; o += static_cast<char>(r.nm); }
  else { o += 's'; put_le16(o, static_cast<uint16_t>(static_cast<int16_t>(r.nm))); }
  o.append("CVA", 3); o += r.cv;
  const uint32_t bs = static_cast<uint32_t>(o.size() - start - 4);
  o[start] = static_cast<char>(bs); o[start + 1] = static_cast<char>(bs >> 8); o[start + 2] = static_cast<char>(bs >> 16); o[start + 3] = static_cast<char>(bs >> 24);
}
// raw bytes -> BGZF blocks (each an independent gzip member with the BC extra field)
int g_bgzf_level = 1;  // deflate level of BAM output (-z): decoded content is the same at every level
void bgzf_compress(const std::string &raw, std::string &out) {
  constexpr size_t kBlock = 0xff00;
  std::vector<unsigned char> buf(compressBound(kBlock) + 64);
  for (size_t at = 0; at < raw.size(); at += kBlock) {
    const size_t len = std::min(kBlock, raw.size() - at);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, g_bgzf_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(raw.data() + at));
    zs.avail_in = static_cast<uInt>(len);
    zs.next_out = buf.data();
    zs.avail_out = static_cast<uInt>(buf.size());
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw std::runtime_error("deflate failed"); }
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(raw.data() + at), static_cast<uInt>(len)));
    static const unsigned char head[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    out.append(reinterpret_cast<const char *>(head), 12);
    out.append("BC", 2); put_le16(out, 2); put_le16(out, static_cast<uint16_t>(clen + 25));
    out.append(reinterpret_cast<const char *>(buf.data()), clen);
    put_le32(out, crc); put_le32(out, static_cast<uint32_t>(len));
  }
}
std::string bam_header_bytes(const std::string &text, const Chroms &ch) {
  std::string o("BAM\1", 4);
  put_le32(o, static_cast<uint32_t>(text.size()));
  o += text;
  put_le32(o, static_cast<uint32_t>(ch.names.size() - 2));
  for (size_t i = 1; i + 1 < ch.names.size(); ++i) {
    put_le32(o, static_cast<uint32_t>(ch.names[i].size() + 1));
    o += ch.names[i]; o += '\0';
    put_le32(o, ch.starts[i + 1] - ch.starts[i]);
  }
  return o;
}
// (t_bam is defined above put_record's first use)

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
void Stats::tally(bool empty_read, const abm_hit &h, bool count_ambig_error, uint32_t bases) {
  ++v[0];
  const bool valid = h.pos != 0, amb = h.flags & 0x100;
  v[1] += valid && !amb; v[2] += valid && amb; v[3] += empty_read;
  if (valid && (!amb || count_ambig_error)) { v[4] += static_cast<uint64_t>(static_cast<int64_t>(h.diffs)); v[5] += bases; }
}
std::string Stats::yaml(const std::string &label) const {
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
std::string Stats::json() const {
  std::ostringstream s;
  s << "{\"edit_distance\":" << v[4] << ",\"reads_mapped_ambiguous\":" << static_cast<uint32_t>(v[2])
    << ",\"reads_mapped_unique\":" << static_cast<uint32_t>(v[1]) << ",\"reads_skipped\":" << static_cast<uint32_t>(v[3])
    << ",\"total_bases\":" << v[5] << ",\"total_reads\":" << static_cast<uint32_t>(v[0]) << "}";
  return s.str();
}

struct Options {
  std::string index, genome, out, stats, timing;
  bool bam = false, json = false, ambig = false, pbat = false, rpbat = false, arich = false, verbose = false;
  uint32_t max_candidates = 0, min_frag = 32, max_frag = 3000;
  uint32_t threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));  // host parse/format threads
  int gpus = 0;
  size_t batch = 0;  // reads (pairs) per batch; 0 = default for the input type
  int mappers = 0;  // mapper threads (contexts) per GPU; 0 = 2 for single-end, 3 for paired-end input
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
    else if (k == "mappers") o.mappers = std::stoi(need(i));
    else if (k == "timing") o.timing = need(i);  // JSON: reads, seconds (first batch submitted -> last byte written), stage busy times
    else if (k == "z" || k == "bam-level") g_bgzf_level = std::max(0, std::min(9, std::stoi(need(i))));
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
  const bool paired = opt.reads.size() == 2;

  std::string index_path = opt.index;
  if (index_path.empty()) {  // -g: index the genome on the fly (src/abismal.cpp:2439-2446)
    index_path = opt.out + ".tmp.idx";
    if (abm_index_build(opt.genome.c_str(), index_path.c_str(), std::max(1u, std::thread::hardware_concurrency())) != 0) die_abm("indexing genome");
  }
  abm_index *ix = nullptr;
  const auto t_index = std::chrono::steady_clock::now();
  if (abm_index_open(index_path.c_str(), &ix) != 0) die_abm("loading index");
  if (opt.index.empty()) std::remove(index_path.c_str());
  Chroms ch;
  for (uint32_t i = 0; i < abm_index_n_chroms(ix); ++i) ch.names.push_back(abm_index_chrom_name(ix, i));
  ch.starts.assign(abm_index_chrom_starts(ix), abm_index_chrom_starts(ix) + ch.names.size() + 1);

  // one replica of the index in each GPU's HBM, shared by that GPU's contexts; a context is one
  // mapper thread's workspaces + stream, and two per GPU keep the device busy while the other
  // thread's batch is in transit over PCIe
  int n_gpus = opt.gpus;
  const int per_gpu = opt.mappers > 0 ? opt.mappers : (paired ? 3 : 2);
  std::vector<abm_ctx *> ctxs;
  for (int d = 0; n_gpus <= 0 || d < n_gpus; ++d) {
    abm_ctx *c = nullptr;
    if (abm_ctx_create(ix, d, &c) != 0) { if (n_gpus <= 0 && d > 0) break; die_abm("creating GPU context"); }
    ctxs.push_back(c);
    for (int k = 1; k < per_gpu; ++k) {
      if (abm_ctx_create(ix, d, &c) != 0) die_abm("creating GPU context");
      ctxs.push_back(c);
    }
  }
  n_gpus = static_cast<int>(ctxs.size()) / per_gpu;
  const double index_load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_index).count();  // file -> host -> HBM

  std::ofstream out(opt.out, std::ios::binary);
  if (!out) throw std::runtime_error("failed to open output file: " + opt.out);
  {  // header, src/abismal.cpp:2265-2293
    std::ostringstream h;
    h << "@HD\tVN:1.0\n";
    for (size_t i = 1; i + 1 < ch.names.size(); ++i) h << "@SQ\tSN:" << ch.names[i] << "\tLN:" << (ch.starts[i + 1] - ch.starts[i]) << '\n';
    h << "@PG\tID:ABISMAL\tVN:" << kVersion << "\tCL:\"";
    for (int i = 0; i < argc; ++i) h << argv[i] << ' ';
    h << "\"\n";
    if (!opt.bam) out << h.str();
    else { std::string z; bgzf_compress(bam_header_bytes(h.str(), ch), z); out.write(z.data(), static_cast<std::streamsize>(z.size())); }
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

  // Staged pipeline, every stage order-agnostic except the writer:
  //   splitter (1 thread)   cuts the FASTQ file(s) into raw batches of whole records
  //   parsers  (-t threads) apply ReadLoader's rules and lay reads out for the C ABI
  //   mappers  (-mappers per GPU) abm_map_{se,pe}_batch
  //   formatters (-t)       SAM text + the batch's statistics
  //   writer (this thread)  emits batches in input order
  std::mutex mu;
  std::condition_variable cv;
  struct Slice { Batch *b; size_t lo, hi; int part; };
  std::deque<std::unique_ptr<Batch>> q_parse, q_map;
  std::deque<Slice> q_format;
  std::map<uint64_t, std::unique_ptr<Batch>> formatting;  // batches whose slices are being formatted
  const size_t slice_reads = 1u << 16;
  std::map<uint64_t, std::unique_ptr<Batch>> done;
  uint64_t n_batches = 0, next_to_write = 0;
  size_t in_flight = 0;
  bool split_done = false;
  int parsers_live = 0, mappers_live = 0, formatters_live = 0;
  std::exception_ptr failure;
  // Reads differ in cost by four orders of magnitude and the costliest ones keep a single wave busy
  // for hundreds of milliseconds, so a batch must be large enough to amortise them (measured on
  // MI355X at hg38 scale: 281 ms per 1 M reads, 589 ms per 4 M, 1437 ms per 10 M)
  const size_t batch_reads = opt.batch ? opt.batch : (paired ? (1u << 21) : (1u << 22));
  const size_t max_in_flight = static_cast<size_t>(2 * n_gpus * per_gpu + 3);
  const unsigned n_host = std::max(1u, opt.threads);
  std::vector<Stats3> gpu_stats(n_gpus);
  const auto t_start = std::chrono::steady_clock::now();

  double busy_split = 0, busy_parse = 0, busy_map = 0, busy_format = 0, busy_write = 0;  // seconds, summed over threads
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  RawPool raw_pool;
  auto fail = [&]() {
    std::lock_guard<std::mutex> lk(mu);
    if (!failure) failure = std::current_exception();
    cv.notify_all();
  };

  auto splitter = [&]() {
    try {
      RawSplitter s1(opt.reads[0]);
      std::unique_ptr<RawSplitter> s2;
      if (paired) s2.reset(new RawSplitter(opt.reads[1]));
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return failure || in_flight < max_in_flight; });
          if (failure) break;
        }
        std::unique_ptr<Batch> b(new Batch);
        const auto t0 = now();
        b->raw[0] = raw_pool.get();
        if (paired) b->raw[1] = raw_pool.get();
        const uint64_t l1 = s1.next(batch_reads, b->raw[0], b->first_line[0]);
        uint64_t l2 = 0;
        if (paired) l2 = s2->next(batch_reads, b->raw[1], b->first_line[1]);
        const bool last = s1.exhausted() || (paired && s2->exhausted());
        if (l1 == 0 && (!paired || l2 == 0)) break;
        {
          std::lock_guard<std::mutex> lk(mu);
          b->seq = n_batches++;
          busy_split += since(t0);
          ++in_flight;
          q_parse.push_back(std::move(b));
        }
        cv.notify_all();
        if (last) break;
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    split_done = true;
    cv.notify_all();
  };

  auto parser = [&]() {
    try {
      for (;;) {
        std::unique_ptr<Batch> b;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return failure || !q_parse.empty() || split_done; });
          if (failure || q_parse.empty()) break;
          b = std::move(q_parse.front());
          q_parse.pop_front();
        }
        const auto t0 = now();
        for (int e = 0; e < (paired ? 2 : 1); ++e) {
          parse_raw(b->raw[e], b->first_line[e], opt.reads[e], b->names[e], b->blob[e], b->off[e]);
          raw_pool.put(std::move(b->raw[e]));
        }
        if (paired && b->names[0].size() != b->names[1].size())
          throw std::runtime_error("paired-end batch sizes differ. Batch 1: " + std::to_string(b->names[0].size()) +
                                   ", batch 2: " + std::to_string(b->names[1].size()) +
                                   ". Are you sure your paired-end inputs have the same number of reads?");
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_parse += since(t0);
          q_map.push_back(std::move(b));
        }
        cv.notify_all();
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    --parsers_live;
    cv.notify_all();
  };

  auto mapper = [&](int slot) {
    const int g = slot / per_gpu;
    abm_ctx *ctx = ctxs[slot];
    try {
      for (;;) {
        std::unique_ptr<Batch> b;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return failure || !q_map.empty() || parsers_live == 0; });
          if (failure || q_map.empty()) break;
          b = std::move(q_map.front());
          q_map.pop_front();
        }
        b->gpu = g;
        const size_t n = b->n();
        const auto t0 = now();
        if (n) {
          // a few CIGAR ops per read are typical; the worst case (read length + 2 each) is only
          // allocated if the first size turns out too small
          const uint64_t worst = std::max<uint64_t>(1, std::max(b->blob[0].size(), b->blob[1].size()) + 2 * n);
          uint64_t cap = std::min<uint64_t>(worst, 4 * n + 1024);
          for (;;) {
            int rc;
            if (!paired) {
              b->se[0].resize(n); b->cig[0].resize(cap); b->cig_off[0].resize(n + 1);
              rc = abm_map_se_batch(ctx, se_mode, &par, n, b->blob[0].data(), b->off[0].data(), b->se[0].data(),
                                    b->cig[0].data(), cap, b->cig_off[0].data());
            }
            else {
              b->pairs.resize(n); b->se[0].resize(n); b->se[1].resize(n);
              for (int e = 0; e < 2; ++e) { b->cig[e].resize(cap); b->cig_off[e].resize(n + 1); }
              rc = abm_map_pe_batch(ctx, pe_mode, &par, n, b->blob[0].data(), b->off[0].data(), b->blob[1].data(),
                                    b->off[1].data(), b->pairs.data(), b->se[0].data(), b->se[1].data(), b->cig[0].data(),
                                    b->cig_off[0].data(), b->cig[1].data(), b->cig_off[1].data(), cap);
            }
            if (rc == 0) break;
            if (rc == ABM_ERR_CAPACITY && cap < worst) { cap = worst; continue; }
            die_abm("mapping");
          }
        }
        {
          const int n_parts = static_cast<int>(std::max<size_t>(1, (n + slice_reads - 1) / slice_reads));
          b->parts.resize(n_parts);
          b->part_stats.resize(n_parts);
          b->parts_left = n_parts;
          std::lock_guard<std::mutex> lk(mu);
          busy_map += since(t0);
          Batch *raw = b.get();
          formatting[raw->seq] = std::move(b);
          for (int k = 0; k < n_parts; ++k)
            q_format.push_back(Slice{raw, k * slice_reads, std::min(n, (k + 1) * slice_reads), k});
        }
        cv.notify_all();
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    --mappers_live;
    cv.notify_all();
  };

  auto format_slice = [&](Batch &bt, size_t lo, size_t hi, std::string &sam, Stats3 &st) {
    Batch *b = &bt;
    t_bam = opt.bam;
    sam.reserve((hi - lo) * (paired ? 2 : 1) * 320);
    if (!paired) {
      for (size_t i = lo; i < hi; ++i) {
        abm_hit h = b->se[0][i];
        const size_t len = b->off[0][i + 1] - b->off[0][i];
        const uint32_t *cg = b->cig[0].data() + b->cig_off[0][i];
        const size_t ncg = b->cig_off[0][i + 1] - b->cig_off[0][i];
        if (len && emit_se(sam, opt.ambig, h, ch, b->names[0][i], b->blob[0].data() + b->off[0][i], len, cg, ncg) == UNMAPPED) h.pos = 0;
        st.s[0].tally(len == 0, h, opt.ambig, ref_len(cg, ncg));
      }
      return;
    }
    for (size_t i = lo; i < hi; ++i) {
      abm_pair p = b->pairs[i];
      abm_hit h1 = b->se[0][i], h2 = b->se[1][i];
      const char *s1 = b->blob[0].data() + b->off[0][i], *s2 = b->blob[1].data() + b->off[1][i];
      const size_t l1 = b->off[0][i + 1] - b->off[0][i], l2 = b->off[1][i + 1] - b->off[1][i];
      const uint32_t *c1 = b->cig[0].data() + b->cig_off[0][i], *c2 = b->cig[1].data() + b->cig_off[1][i];
      const size_t nc1 = b->cig_off[0][i + 1] - b->cig_off[0][i], nc2 = b->cig_off[1][i + 1] - b->cig_off[1][i];
      // select_output, src/abismal.cpp:1073-1088
      const Outcome po = emit_pe(sam, opt.ambig, p, ch, b->names[0][i], b->names[1][i], s1, l1, s2, l2, c1, nc1, c2, nc2);
      const bool report = p.r1.pos != 0 && (opt.ambig || !(p.r1.flags & 0x100));
      bool pair_ok = report;
      if (!report || po == UNMAPPED) {
        if (po == UNMAPPED) { p.r1.pos = 0; p.r2.pos = 0; pair_ok = false; }
        if (emit_se(sam, opt.ambig, h1, ch, b->names[0][i], s1, l1, c1, nc1) == UNMAPPED) h1.pos = 0;
        if (emit_se(sam, opt.ambig, h2, ch, b->names[1][i], s2, l2, c2, nc2) == UNMAPPED) h2.pos = 0;
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
  };

  auto formatter = [&]() {
    try {
      for (;;) {
        Slice sl;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return failure || !q_format.empty() || mappers_live == 0; });
          if (failure || q_format.empty()) break;
          sl = q_format.front();
          q_format.pop_front();
        }
        const auto t0 = now();
        std::string &text = sl.b->parts[sl.part];
        format_slice(*sl.b, sl.lo, sl.hi, text, sl.b->part_stats[sl.part]);
        if (opt.bam) { std::string z; bgzf_compress(text, z); text.swap(z); }
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_format += since(t0);
          if (--sl.b->parts_left == 0) {
            auto it = formatting.find(sl.b->seq);
            done[sl.b->seq] = std::move(it->second);
            formatting.erase(it);
          }
        }
        cv.notify_all();
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    --formatters_live;
    cv.notify_all();
  };

  std::vector<std::thread> threads;
  parsers_live = static_cast<int>(n_host);
  mappers_live = n_gpus * per_gpu;
  formatters_live = static_cast<int>(n_host);
  threads.emplace_back(splitter);
  for (unsigned t = 0; t < n_host; ++t) threads.emplace_back(parser);
  for (int slot = 0; slot < n_gpus * per_gpu; ++slot) threads.emplace_back(mapper, slot);
  for (unsigned t = 0; t < n_host; ++t) threads.emplace_back(formatter);
  uint64_t total_records = 0;
  {  // writer: batches leave in input order
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return failure || done.count(next_to_write) || (formatters_live == 0 && done.empty()); });
      if (failure) break;
      auto it = done.find(next_to_write);
      if (it == done.end()) {
        if (formatters_live == 0 && done.empty()) break;
        continue;
      }
      std::unique_ptr<Batch> b = std::move(it->second);
      done.erase(it);
      ++next_to_write;
      --in_flight;
      lk.unlock();
      const auto t0 = now();
      for (const std::string &part : b->parts) out.write(part.data(), static_cast<std::streamsize>(part.size()));
      if (!out) {  // a full disk must not end in a truncated file and exit code 0
        lk.lock();
        if (!failure) failure = std::make_exception_ptr(std::runtime_error("failed writing output file: " + opt.out));
        break;
      }
      total_records += b->n();
      for (const Stats3 &ps : b->part_stats)
        for (int k = 0; k < 3; ++k)
          for (int j = 0; j < 6; ++j) gpu_stats[b->gpu].s[k].v[j] += ps.s[k].v[j];
      b.reset();
      busy_write += since(t0);
      cv.notify_all();
      lk.lock();
    }
  }
  cv.notify_all();
  for (auto &t : threads) t.join();
  if (failure) std::rethrow_exception(failure);
  if (opt.bam) {  // BGZF end-of-file marker
    static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    out.write(reinterpret_cast<const char *>(eof_block), 28);
  }
  out.close();
  if (!out) throw std::runtime_error("failed writing output file: " + opt.out);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();

  // statistics (6 counters x 3 structs, src/abismal.cpp:865-895, :1034-1037).  Every GPU's counters
  // already sit in this process, so the total is a host sum; with more than one GPU the same sum is
  // also taken with the path's one collective (RCCL all-reduce over xGMI, abm_stats_allreduce) and
  // the two must agree.  A collective that cannot run (no RCCL transport on this box) costs a
  // warning, never the statistics file of a finished run.
  static_assert(sizeof(Stats3) == 18 * sizeof(uint64_t), "18 counters");
  Stats3 tot;
  for (const Stats3 &g : gpu_stats)
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 6; ++j) tot.s[k].v[j] += g.s[k].v[j];
  if (n_gpus > 1 && !std::getenv("ABM_CLI_NO_RCCL")) {
    std::vector<Stats3> reduced(gpu_stats);
    std::vector<uint64_t *> ptrs;
    for (auto &s : reduced) ptrs.push_back(&s.s[0].v[0]);
    std::vector<abm_ctx *> primary;
    for (int g = 0; g < n_gpus; ++g) primary.push_back(ctxs[static_cast<size_t>(g) * per_gpu]);
    if (abm_stats_allreduce(primary.data(), n_gpus, ptrs.data()) != 0)
      std::cerr << "[abismal-amd] warning: RCCL statistics all-reduce failed (" << abm_last_error() << "); using the host sum\n";
    else {
      for (int g = 0; g < n_gpus; ++g)
        if (std::memcmp(&reduced[g], &tot, sizeof(Stats3)) != 0)
          throw std::runtime_error("statistics all-reduce disagrees with the host sum on GPU " + std::to_string(g));
      if (opt.verbose) std::cerr << "[abismal-amd] statistics summed over " << n_gpus << " GPUs with one RCCL all-reduce\n";
    }
  }
  if (!opt.stats.empty()) {
    std::ofstream so(opt.stats);
    if (!so) std::cerr << "failed to open stats out file: " << opt.stats << '\n';
    else if (opt.json) {
      if (!paired) so << tot.s[0].json();
      else so << "{\"end1_stats\":" << tot.s[1].json() << ",\"end2_stats\":" << tot.s[2].json() << ",\"read_pair_stats\":" << tot.s[0].json() << "}";
    }
    else if (!paired) so << tot.s[0].yaml("read1");
    else { so << tot.s[0].yaml("pairs"); if (!opt.ambig) so << tot.s[1].yaml("read1") << tot.s[2].yaml("read2"); }
    so.close();
    if (!so) throw std::runtime_error("failed writing stats file: " + opt.stats);
  }
  if (!opt.timing.empty()) {
    std::ofstream tj(opt.timing);
    tj << "{\"records\": " << total_records << ", \"reads\": " << (paired ? 2 : 1) * total_records << ", \"seconds\": " << secs
       << ", \"index_load_s\": " << index_load_s << ", \"gpus\": " << n_gpus << ", \"mappers_per_gpu\": " << per_gpu
       << ", \"host_threads\": " << n_host << ", \"batch_reads\": " << batch_reads << ", \"busy_s\": {\"split\": " << busy_split
       << ", \"parse\": " << busy_parse << ", \"map\": " << busy_map << ", \"format\": " << busy_format << ", \"write\": "
       << busy_write << "}}\n";
  }
  if (opt.verbose)
    std::cerr << "[abismal-amd] " << total_records << (paired ? " pairs" : " reads") << " on " << n_gpus << " GPU(s) in "
              << secs << " s (" << (paired ? 2 : 1) * total_records / secs << " reads/s incl. host I/O)\n"
              << "[abismal-amd] busy seconds: split " << busy_split << ", parse " << busy_parse << " (" << n_host
              << " threads), map " << busy_map << " (" << n_gpus * per_gpu << " threads), format " << busy_format << " ("
              << n_host << " threads), write " << busy_write << "\n";
  for (abm_ctx *c : ctxs) abm_ctx_destroy(c);
  abm_index_close(ix);
  return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char **argv) {
  try {
    if (argc < 2) { std::cout << "Program: abismal-amd\nVersion: " << kVersion << "\nUsage: abismal-amd <command> [options]\nCommands:\n    map:    map FASTQ reads to an index or a FASTA reference genome\n    idx:    make an index for a FASTA reference genome\n    sim:    simulate WGBS reads for a FASTA reference genome\n"; return EXIT_SUCCESS; }
    const std::string cmd = argv[1];
    if (cmd == "map") return cmd_map(argc - 1, argv + 1);
    if (cmd == "idx") return cmd_idx(argc - 1, argv + 1);
    if (cmd == "sim") return abm::sim_main(argc - 1, argv + 1);
    std::cerr << "ERROR: invalid command " << cmd << '\n';
    return EXIT_SUCCESS;
  }
  catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return EXIT_FAILURE;
  }
}
