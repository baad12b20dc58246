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
#include <sys/stat.h>
#include <unistd.h>
#include <malloc.h>
#include <sys/mman.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cerrno>
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
constexpr uint32_t kPadding = 32767;
uint32_t g_min_read_len = 44;  // key weight + the index's window - 1 (src/abismal.cpp:212-213): 36 with a short-read index

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
  bool pinned = false;  // page-locked memory from the library (abm_host_alloc): what a batch's reads are uploaded from
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  RawBuf(RawBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap), pinned(o.pinned) { o.p = nullptr; o.n = o.cap = 0; }
  RawBuf &operator=(RawBuf &&o) noexcept { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(pinned, o.pinned); return *this; }
  ~RawBuf() { if (pinned) abm_host_free(p); else std::free(p); }
  // Big blocks are 2 MB-aligned and advised to use huge pages: a run touches gigabytes of fresh memory
  // from a hundred threads at once, and with 4 KB pages that is a million page faults on one address space.
  void reserve(size_t want) {
    if (want <= cap) return;
    want = std::max(want, cap + cap / 2);
    char *q;
    if (pinned) {
      void *v = nullptr;
      if (abm_host_alloc(want, &v) != 0) throw std::bad_alloc();
      q = static_cast<char *>(v);
      if (n) std::memcpy(q, p, n);
      abm_host_free(p);
    }
    else if (want >= (4u << 20)) {
      want = (want + (2u << 20) - 1) & ~static_cast<size_t>((2u << 20) - 1);
      q = static_cast<char *>(std::aligned_alloc(2u << 20, want));
      if (!q) throw std::bad_alloc();
      ::madvise(q, want, MADV_HUGEPAGE);
      if (n) std::memcpy(q, p, n);
      std::free(p);
    }
    else {
      q = static_cast<char *>(std::realloc(p, want));
      if (!q) throw std::bad_alloc();
    }
    p = q; cap = want;
  }
  void append(const char *src, size_t len) { reserve(n + len); std::memcpy(p + n, src, len); n += len; }
  // the part of std::string's interface the formatting code uses (contents are never zero-filled)
  void append(size_t count, char c) { reserve(n + count); std::memset(p + n, c, count); n += count; }
  void append(const std::string &t) { append(t.data(), t.size()); }
  RawBuf &operator+=(char c) { if (n == cap) reserve(n + 1); p[n++] = c; return *this; }
  RawBuf &operator+=(const std::string &t) { append(t.data(), t.size()); return *this; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  void resize(size_t m) { reserve(m); n = m; }
  void clear() { n = 0; }
  char &operator[](size_t i) { return p[i]; }
  const char &operator[](size_t i) const { return p[i]; }
  char *data() { return p; }
  const char *data() const { return p; }
  void swap(RawBuf &o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); }
};
// std::vector<T>'s resize/data/[] for plain-data T without the zero fill (a batch's result arrays are a few
// hundred megabytes that the C ABI overwrites entirely; filling them first, single-threaded, cost a 8 M-read
// batch 0.3 s before its upload could start)
template <class T> struct PodVec {
  RawBuf b;
  void resize(size_t n) { b.resize(n * sizeof(T)); }
  void assign(size_t n, T v) { resize(n); for (size_t i = 0; i < n; ++i) data()[i] = v; }
  size_t size() const { return b.size() / sizeof(T); }
  T *data() { return reinterpret_cast<T *>(b.p); }
  const T *data() const { return reinterpret_cast<const T *>(b.p); }
  T &operator[](size_t i) { return data()[i]; }
  const T &operator[](size_t i) const { return data()[i]; }
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
    if (free_list.size() < 1024) free_list.push_back(std::move(b));
  }
};

struct NameRef {  // a read name inside its slice's FASTQ text
  const char *p;
  uint32_t n;
};
struct Batch;

// The unit of host work: up to `slice_reads` records of the input, in file order.  Slices are cut,
// parsed, formatted and written independently; a batch handed to a GPU is a run of consecutive slices.
struct Slice {
  uint64_t g = 0;                    // slice number = output order
  uint64_t first_line[2] = {0, 0};
  uint64_t byte_lo[2] = {0, 0}, byte_hi[2] = {0, 0};  // plain files: the slice's text in each file
  RawBuf raw[2];                     // the FASTQ text (names point into it)
  std::vector<NameRef> names[2];
  RawBuf blob[2];                    // reads as ReadLoader hands them over, concatenated
  std::vector<uint64_t> off[2];
  size_t n() const { return names[0].size(); }
  Batch *batch = nullptr;            // once mapped: the batch whose arrays hold this slice's results ...
  size_t base = 0;                   // ... from this index on
  RawBuf text;                       // formatted output
  Stats3 stats;
  // single-end batches hand their results over slice by slice while the kernel runs (abm_map_se_batch_sliced): the
  // slice then holds its own copy -- hits and a compact CIGAR blob with n() + 1 offsets
  bool own = false;
  PodVec<abm_hit> own_se;
  PodVec<uint32_t> own_cig;
  PodVec<uint64_t> own_cig_off;
};

// written slices are recycled with their buffers (names, reads, output text keep their capacity): a
// process with a hundred threads that keeps allocating and freeing multi-megabyte blocks spends its
// time on the address-space lock
struct SlicePool {
  std::mutex mu;
  std::vector<std::unique_ptr<Slice>> free_list;
  std::unique_ptr<Slice> get() {
    std::unique_ptr<Slice> s;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_list.empty()) { s = std::move(free_list.back()); free_list.pop_back(); }
    }
    if (!s) s.reset(new Slice);
    return s;
  }
  void put(std::unique_ptr<Slice> s) {
    for (int e = 0; e < 2; ++e) { s->names[e].clear(); s->blob[e].clear(); s->off[e].clear(); s->raw[e].n = 0; }
    s->text.clear();
    s->stats = Stats3();
    s->batch = nullptr;
    s->base = 0;
    s->own = false;
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.size() < 1024) free_list.push_back(std::move(s));
  }
};

struct Batch {
  Batch() { for (int e = 0; e < 2; ++e) blob[e].pinned = off_bytes[e].pinned = true; }
  uint64_t seq = 0;
  int gpu = 0;
  std::vector<std::unique_ptr<Slice>> slices;
  size_t n = 0;
  std::vector<std::string> carry[2]; // reads of the input just before this batch, mapped along for their side effects only
  RawBuf blob[2];                    // carry + the slices' reads concatenated, as the C ABI takes them
  RawBuf off_bytes[2];               // ... and their n + 1 offsets (uint64_t)
  uint64_t *off_of(int e) { return reinterpret_cast<uint64_t *>(off_bytes[e].p); }
  PodVec<abm_hit> se[2];
  PodVec<abm_pair> pairs;
  PodVec<uint32_t> cig[2];
  PodVec<uint64_t> cig_off[2];
  int slices_left = 0;               // not yet written
};

// batches are recycled with their buffers as well (a full batch's arrays are a gigabyte)
struct BatchPool {
  std::mutex mu;
  std::vector<std::unique_ptr<Batch>> free_list;
  std::unique_ptr<Batch> get() {
    std::unique_ptr<Batch> b;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!free_list.empty()) { b = std::move(free_list.back()); free_list.pop_back(); }
    }
    if (!b) b.reset(new Batch);
    return b;
  }
  void put(std::unique_ptr<Batch> b) {
    b->slices.clear();
    b->n = 0; b->seq = 0; b->gpu = 0; b->slices_left = 0;
    for (int e = 0; e < 2; ++e) { b->carry[e].clear(); b->blob[e].clear(); b->off_bytes[e].clear(); }
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.size() < 64) free_list.push_back(std::move(b));
  }
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

void parse_raw(const RawBuf &raw, uint64_t first_line, const std::string &path, std::vector<NameRef> &names,
               RawBuf &blob, std::vector<uint64_t> &off) {
  names.clear(); blob.clear(); off.assign(1, 0);
  blob.reserve(raw.n / 2);
  names.reserve(raw.n / 200 + 16);
  off.reserve(raw.n / 200 + 16);
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
      names.push_back(NameRef{p + 1, static_cast<uint32_t>(q - (p + 1))});
    }
    else if (k % 4 == 1) {
      const size_t len = static_cast<size_t>(le - p);
      if (len >= kPadding)
        throw std::runtime_error("found a read of size " + std::to_string(len) +
                                 ", which is too long. Maximum allowed read size = " + std::to_string(kPadding));
      size_t informative = 0;
      for (const char *c = p; c < le; ++c) informative += (*c != 'N');
      if (informative >= g_min_read_len) {
        const char *e = le;
        while (e > p && e[-1] == 'N') --e;                       // remove Ns from 3'
        const char *b = p;
        while (b < e && *b != 'A' && *b != 'C' && *b != 'G' && *b != 'T') ++b;  // ... and everything before the first base
        if (b == e) throw std::runtime_error("read without A/C/G/T at line " + std::to_string(first_line + k));
        blob.append(b, static_cast<size_t>(e - b));
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

template <class S> inline void put_uint(S &o, uint64_t v) {
  char buf[24];
  int k = 24;
  do { buf[--k] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
  o.append(buf + k, static_cast<size_t>(24 - k));
}
template <class S> inline void put_int(S &o, int64_t v) {
  if (v < 0) { o += '-'; put_uint(o, static_cast<uint64_t>(-v)); }
  else put_uint(o, static_cast<uint64_t>(v));
}

struct Record {
  const NameRef *name;
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

template <class S> void put_bam_record(S &o, const Record &r);
thread_local bool t_bam = false;  // formatter threads switch put_record to BAM encoding
template <class S> void put_record(S &o, const Chroms &ch, const Record &r) {
  if (t_bam) { put_bam_record(o, r); return; }
  o.append(r.name->p, r.name->n); o += '\t'; put_uint(o, r.flag); o += '\t';
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
template <class S> inline void put_le32(S &o, uint32_t v) { char b[4] = {static_cast<char>(v), static_cast<char>(v >> 8), static_cast<char>(v >> 16), static_cast<char>(v >> 24)}; o.append(b, 4); }
template <class S> inline void put_le16(S &o, uint16_t v) { char b[2] = {static_cast<char>(v), static_cast<char>(v >> 8)}; o.append(b, 2); }
inline int reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return static_cast<int>(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return static_cast<int>(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return static_cast<int>(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return static_cast<int>(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return static_cast<int>(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
template <class S> void put_bam_record(S &o, const Record &r) {
  static const char nt16[] = "=ACMGRSVTWYHKDBN";
  const size_t start = o.size();
  put_le32(o, 0);  // block_size, patched below
  put_le32(o, static_cast<uint32_t>(r.tid));
  put_le32(o, r.pos);
  const uint32_t rl = ref_len(r.cig, r.n_cig);
  o += static_cast<char>(r.name->n + 1);
  o += static_cast<char>(255);
  put_le16(o, static_cast<uint16_t>(reg2bin(r.pos, static_cast<int64_t>(r.pos) + (rl ? rl : 1))));
  put_le16(o, static_cast<uint16_t>(r.n_cig));
  put_le16(o, r.flag);
  put_le32(o, static_cast<uint32_t>(r.n_seq));
  put_le32(o, static_cast<uint32_t>(r.mtid));
  put_le32(o, r.mtid < 0 ? 0xFFFFFFFFu : r.mpos);
  put_le32(o, static_cast<uint32_t>(r.tlen));
  o.append(r.name->p, r.name->n); o += '\0';
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
template <class A, class B> void bgzf_compress(const A &raw, B &out) {
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

template <class S> Outcome emit_se(S &o, bool allow_ambig, const abm_hit &h, const Chroms &ch, const NameRef &name,
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

template <class S> Outcome emit_pe(S &o, bool allow_ambig, const abm_pair &p, const Chroms &ch, const NameRef &n1,
                const NameRef &n2, const char *s1, size_t l1, const char *s2, size_t l2, const uint32_t *c1,
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
  uint32_t threads = 0;  // host parse/format threads (-t); 0 = chosen from the GPU count (see cmd_map)
  int gpus = 0;
  size_t batch = 0;  // reads (pairs) per batch; 0 = default for the input type
  int mappers = 0;  // mapper threads (contexts) per GPU; 0 = 2 for single-end, 3 for paired-end input
  int ext2 = -1, ext3 = -1;  // -seed-ext a,b: letters of the seed-extension tables (default: chosen from the index's size)
  bool skip_long = false;     // -skip-long: pairs with an end beyond the paired-end kernels' 1024 bases are written unmapped
                              // (and counted in the warning) instead of failing the run
  bool host_ceiling = false;  // -host-ceiling (diagnostic): no mapping call; every read gets a made-up hit, so that cut ->
                              // parse -> format -> write run at the rate the host can carry (single-end input)
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
    else if (k == "host-ceiling") o.host_ceiling = true;
    else if (k == "skip-long") o.skip_long = true;
    else if (k == "seed-ext") { const std::string v = need(i); if (std::sscanf(v.c_str(), "%d,%d", &o.ext2, &o.ext3) != 2) throw std::runtime_error("-seed-ext wants two numbers: a,b"); }
    else if (k == "timing") o.timing = need(i);  // JSON: reads, seconds (first batch submitted -> last byte written), stage busy times
    else if (k == "z" || k == "bam-level") g_bgzf_level = std::max(0, std::min(9, std::stoi(need(i))));
    else throw std::runtime_error("unknown option " + a);
  }
  return o;
}

int cmd_idx(int argc, char **argv) {
  unsigned threads = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::string> pos;
  std::string targets;  // -A: index only these regions (src/abismalidx.cpp:51-52, :91-92)
  uint32_t window = 20;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if ((a == "-t" || a == "-threads" || a == "--threads") && i + 1 < argc) threads = static_cast<unsigned>(std::stoul(argv[++i]));
    else if ((a == "-A" || a == "-targets" || a == "--targets") && i + 1 < argc) targets = argv[++i];
    else if ((a == "-w" || a == "-window" || a == "--window") && i + 1 < argc) window = static_cast<uint32_t>(std::stoul(argv[++i]));
    else if (a == "-short" || a == "--short") window = 12;  // what the reference's --enable-short build indexes with
    else if (a == "-v" || a == "-verbose") {}
    else pos.push_back(a);
  }
  if (pos.size() != 2) { std::cerr << "usage: abismal-amd idx [-t n] [-A targets] [-short | -w 12] <genome.fa> <out.idx>\n"; return EXIT_SUCCESS; }
  if (abm_index_build_opts(pos[0].c_str(), targets.c_str(), window, pos[1].c_str(), threads) != 0) die_abm("idx");
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
  g_min_read_len = 24 + abm_index_window(ix);
  Chroms ch;
  for (uint32_t i = 0; i < abm_index_n_chroms(ix); ++i) ch.names.push_back(abm_index_chrom_name(ix, i));
  ch.starts.assign(abm_index_chrom_starts(ix), abm_index_chrom_starts(ix) + ch.names.size() + 1);

  // one replica of the index in each GPU's HBM, shared by that GPU's contexts; a context is one
  // mapper thread's workspaces + stream, and two per GPU keep the device busy while the other
  // thread's batch is in transit over PCIe
  int n_gpus = opt.gpus;
  const int per_gpu = opt.mappers > 0 ? opt.mappers : (paired ? 3 : 2);
  if (opt.ext2 >= 0 && abm_index_set_seed_extension(ix, opt.ext2, opt.ext3) != 0) die_abm("seed extension");
  if (opt.max_candidates && abm_index_set_max_candidates(ix, opt.max_candidates) != 0) die_abm("max candidates");
  std::vector<abm_ctx *> ctxs;
  {
    // the first context on a GPU uploads the index and derives its tables there: every GPU's at the same time
    if (n_gpus <= 0) n_gpus = abm_device_count();  // all that are visible
    if (n_gpus <= 0) { std::cerr << "creating GPU context: no HIP device present (the mapping path has no CPU fallback)\n"; return EXIT_FAILURE; }
    std::vector<abm_ctx *> first(n_gpus, nullptr);
    std::vector<std::thread> th;
    std::mutex emu;
    std::string err;
    for (int d = 0; d < n_gpus; ++d)
      if (!first[d]) th.emplace_back([&, d] {
        if (abm_ctx_create(ix, d, &first[d]) != 0) { std::lock_guard<std::mutex> lk(emu); err = abm_last_error(); }
      });
    for (auto &t : th) t.join();
    if (!err.empty()) { std::cerr << "creating GPU context: " << err << "\n"; return EXIT_FAILURE; }
    for (int d = 0; d < n_gpus; ++d) {
      ctxs.push_back(first[d]);
      for (int k = 1; k < per_gpu; ++k) {
        abm_ctx *c = nullptr;
        if (abm_ctx_create(ix, d, &c) != 0) die_abm("creating GPU context");
        ctxs.push_back(c);
      }
    }
  }
  {
    // set-up, like the index upload: workspaces for full batches of reads as long as the input's first one, and the
    // kernels' code loaded, before the clock of the run starts (a longer read later only makes the buffers grow)
    uint32_t first_len = 100;
    if (gzFile zf = gzopen(opt.reads[0].c_str(), "rb")) {
      char line[65536];
      if (gzgets(zf, line, sizeof(line)) && gzgets(zf, line, sizeof(line))) first_len = static_cast<uint32_t>(std::strcspn(line, "\r\n"));
      gzclose(zf);
    }
    // (the same expression the mappers use for a full batch, rounded up to whole slices as they do)
    auto env_reads = [](const char *name, size_t dflt) { const char *e = std::getenv(name); return e && std::atoll(e) > 0 ? static_cast<size_t>(std::atoll(e)) : dflt; };
    const size_t slice_for_reserve = env_reads("ABM_CLI_SLICE_READS", 1u << 15);
    size_t reserve_reads = opt.batch ? opt.batch : env_reads("ABM_CLI_BATCH_READS", paired ? (1u << 21) : (1u << 23));
    reserve_reads = (reserve_reads + slice_for_reserve - 1) / slice_for_reserve * slice_for_reserve + 256;
    {  // (no more than the input can hold: a record is at least two sequence-length lines)
      struct stat sb;
      if (::stat(opt.reads[0].c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
        unsigned char magic[2] = {0, 0};
        const int fd = ::open(opt.reads[0].c_str(), O_RDONLY);
        const bool gz = fd >= 0 && ::pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (fd >= 0) ::close(fd);
        // (gzip: FASTQ deflates five- to sixfold at most in practice; ten is a safe bound on what the file can hold)
        reserve_reads = std::min<size_t>(reserve_reads, static_cast<size_t>(sb.st_size) * (gz ? 10 : 1) / (2 * std::max<uint32_t>(first_len, 1) + 4) + 256);
      }
    }
    std::vector<std::thread> warm;
    std::exception_ptr werr;
    std::mutex wmu;
    for (abm_ctx *c : ctxs)
      warm.emplace_back([&, c] {
        if (abm_ctx_reserve(c, reserve_reads, first_len, paired ? 1 : 0) != 0) {
          std::lock_guard<std::mutex> lk(wmu);
          if (!werr) werr = std::make_exception_ptr(std::runtime_error(std::string("preparing GPU context: ") + abm_last_error()));
        }
      });
    for (auto &t : warm) t.join();
    if (werr) std::rethrow_exception(werr);
  }
  const double index_load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_index).count();  // file -> host -> HBM

  // ---- output file.  A slice's place is fixed in slice order; then whichever thread is free pwrite()s it.
  // (Writes to one file take its inode lock, so they run one at a time at ~4 GB/s on tmpfs; copying into a
  // shared mapping of the file from all threads instead was measured 3x SLOWER -- page faults on the
  // mapping contend far worse than the lock.)
  mallopt(M_MMAP_THRESHOLD, 32 << 20);  // (the largest value the library takes: blocks below it come from its arenas ...)
  mallopt(M_TRIM_THRESHOLD, 1 << 30);   // (... and stay there)
  const int out_fd = ::open(opt.out.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (out_fd < 0) throw std::runtime_error("failed to open output file: " + opt.out);
  struct FdCloser { int fd; ~FdCloser() { if (fd >= 0) ::close(fd); } } out_closer{out_fd};
  const bool seekable = ::lseek(out_fd, 0, SEEK_CUR) >= 0;
  auto write_all = [&](const char *p, size_t len, uint64_t at) {
    while (len) {
      const ssize_t w = seekable ? ::pwrite(out_fd, p, len, static_cast<off_t>(at)) : ::write(out_fd, p, len);
      if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error("failed writing output file: " + opt.out); }
      p += w; at += static_cast<uint64_t>(w); len -= static_cast<size_t>(w);
    }
  };
  uint64_t file_offset = 0;  // bytes of output whose place is fixed
  {  // header, src/abismal.cpp:2265-2293
    std::ostringstream h;
    h << "@HD\tVN:1.0\n";
    for (size_t i = 1; i + 1 < ch.names.size(); ++i) h << "@SQ\tSN:" << ch.names[i] << "\tLN:" << (ch.starts[i + 1] - ch.starts[i]) << '\n';
    h << "@PG\tID:ABISMAL\tVN:" << kVersion << "\tCL:\"";
    for (int i = 0; i < argc; ++i) h << argv[i] << ' ';
    h << "\"\n";
    std::string z;
    if (!opt.bam) z = h.str();
    else bgzf_compress(bam_header_bytes(h.str(), ch), z);
    write_all(z.data(), z.size(), 0);
    file_offset = z.size();
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

  // Pipeline.  The unit of host work is a SLICE (32 k records, in file order); every stage runs on many
  // slices at once and only the assignment of output offsets looks at their order:
  //   cut      plain files: a pool counts newlines chunk by chunk (pread), one thread turns the counts into
  //            slice byte ranges -- record j starts at line 4j, so no guessing at record boundaries;
  //            gzip files: one thread per file inflates and cuts (inherently serial)
  //   parse    (-t threads) pread the slice's text, apply ReadLoader's rules, lay the reads out for the C ABI
  //   map      (-mappers per GPU) a mapper takes EVERY consecutive parsed slice that is ready, up to -batch
  //            reads: batches start small (the GPU is busy a few ms after the first slice is cut) and grow
  //            to the size at which the kernel is efficient once the host runs ahead
  //   format   (-t) SAM text / BAM blocks and statistics per slice
  //   write    a slice's place in the file is known once every earlier slice's size is; pwrite from any thread
  std::mutex mu;
  // one mutex, one condition variable per kind of waiter: an event wakes the threads it concerns, not all two hundred
  std::condition_variable cv_flow,   // the cutter: room for more reads in flight
                          cv_chunk,  // the cutter of plain files: a chunk's newline counts are there
                          cv_parse,  // parsers: a slice has been cut
                          cv_map,    // mappers: a slice has been parsed
                          cv_work,   // formatters: a batch has been mapped
                          cv_write;  // the writer: a slice's place in the file is fixed
  auto wake_everyone = [&] { cv_flow.notify_all(); cv_chunk.notify_all(); cv_parse.notify_all(); cv_map.notify_all(); cv_work.notify_all(); cv_write.notify_all(); };
  // (test hooks: ABM_CLI_SLICE_READS / ABM_CLI_CHUNK_BYTES / ABM_CLI_MARK_LINES shrink the units so that small
  // fixtures cross many slice, chunk and mark boundaries)
  auto env_or = [](const char *name, uint64_t dflt) { const char *e = std::getenv(name); return e && std::atoll(e) > 0 ? static_cast<uint64_t>(std::atoll(e)) : dflt; };
  const size_t slice_reads = static_cast<size_t>(env_or("ABM_CLI_SLICE_READS", 1u << 15));
  // size of the run's very first batch (see the mapper's target()); ABM_CLI_FIRST_BATCH=n overrides, a huge n = no special first batch
  const size_t first_batch_reads = static_cast<size_t>(env_or("ABM_CLI_FIRST_BATCH", (opt.reads.size() == 1 && !opt.host_ceiling && !std::getenv("ABM_CLI_NO_STREAM")) ? 1u << 19 : 1u << 21));
  const bool plain_input = [&] {
    for (const std::string &path : opt.reads) {
      const int fd = ::open(path.c_str(), O_RDONLY);
      if (fd < 0) throw std::runtime_error("cannot open reads file: " + path);
      unsigned char magic[2] = {0, 0};
      const ssize_t got = ::pread(fd, magic, 2, 0);
      struct stat sb;
      const bool regular = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
      ::close(fd);
      if (!regular || (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) return false;
    }
    return true;
  }();
  // Batches are FULL (-batch reads) except at the end of the input: the mapping kernel's time has a floor set
  // by its costliest reads (a quarter of a second, whatever the batch), so small batches waste the GPU;
  // cutting and parsing run far ahead of it, so a full batch is ready within a fraction of a kernel's time.
  const size_t batch_reads = opt.batch ? opt.batch : static_cast<size_t>(env_or("ABM_CLI_BATCH_READS", paired ? (1u << 21) : (1u << 23)));
  // host threads: -t, else 64 for one GPU and 24 more per further GPU (what 10 M reads/s per GPU of cutting, parsing and
  // formatting take on the measured busy times), never more than the box has
  const unsigned hw_threads = std::max(1u, std::thread::hardware_concurrency());
  const unsigned n_host = opt.threads ? std::max(1u, opt.threads) : std::min(hw_threads, std::max(64u, 40u + 24u * static_cast<unsigned>(n_gpus)));
  const size_t max_reads_in_flight = (static_cast<size_t>(n_gpus) * per_gpu + 2) * batch_reads + 4 * slice_reads * n_host;
  std::deque<std::unique_ptr<Slice>> q_parse;               // cut, waiting for a parser
  std::map<uint64_t, std::unique_ptr<Slice>> parsed;        // parsed, waiting for a mapper (by slice number)
  std::deque<Slice *> q_format;                             // mapped, waiting for a formatter
  std::deque<Slice *> q_write;                              // formatted and placed, waiting to be written
  std::map<uint64_t, Slice *> formatted;                    // formatted, place not yet known
  std::map<uint64_t, uint64_t> place;                       // slice -> file offset
  std::vector<std::unique_ptr<Batch>> live_batches;
  std::vector<std::string> carry[2];                        // tail of the input already handed to a batch (see the mapper)
  uint64_t n_slices = 0, next_to_map = 0, next_to_place = 0, slices_written = 0, n_batches = 0;
  uint64_t run_end = 0, n_parsed = 0;  // parsed[next_to_map .. run_end) are all there; slices parsed so far
  size_t run_reads = 0;                // reads in that run
  SlicePool slice_pool;
  BatchPool batch_pool;
  size_t reads_in_flight = 0;
  bool cut_done = false;
  int parsers_live = 0, mappers_live = 0;
  std::exception_ptr failure;
  std::vector<Stats3> gpu_stats(n_gpus);
  std::vector<uint64_t> gpu_batches(n_gpus, 0), gpu_reads(n_gpus, 0);  // what each GPU was handed
  uint64_t total_records = 0;

  // Set-up, like the index upload and abm_ctx_reserve: the slices and batches the run will have in flight, with their
  // buffers sized from the input's first records and their pages touched, on all host threads at once.  A run's first
  // second otherwise touches gigabytes of fresh memory from a hundred threads that share one address space -- page
  // faults and the allocator's calls for more memory, which stall one another (profiles/r03_host_ceiling.log: the
  // formatting threads' busy time grew sixfold from 32 to 128 threads).
  double host_prepare_s = 0;
  if (plain_input && !std::getenv("ABM_CLI_NO_PREWARM")) {
    const auto tp = std::chrono::steady_clock::now();
    uint64_t in_bytes = 0, rec_bytes = 0, read_len = 0;
    {
      struct stat sb;
      if (::stat(opt.reads[0].c_str(), &sb) == 0) in_bytes = static_cast<uint64_t>(sb.st_size);
      std::vector<char> head(1u << 18);
      const int fd = ::open(opt.reads[0].c_str(), O_RDONLY);
      const ssize_t got = fd >= 0 ? ::pread(fd, head.data(), head.size(), 0) : 0;
      if (fd >= 0) ::close(fd);
      uint64_t lines = 0, last = 0;
      size_t second_len = 0, line_start = 0;
      for (ssize_t i = 0; i < got; ++i)
        if (head[i] == '\n') {
          if (lines == 1) second_len = static_cast<size_t>(i) - line_start;
          ++lines; last = static_cast<uint64_t>(i) + 1; line_start = static_cast<size_t>(i) + 1;
        }
      if (lines >= 4) { rec_bytes = last * 4 / (lines - lines % 4 ? lines - lines % 4 : lines); read_len = second_len; }
    }
    if (rec_bytes && read_len) {
      const uint64_t n_recs = in_bytes / rec_bytes + 1;
      const uint64_t in_flight = std::min<uint64_t>(n_recs, max_reads_in_flight);
      const size_t want_slices = static_cast<size_t>(std::min<uint64_t>((in_flight + slice_reads - 1) / slice_reads + n_host / 4, 768));
      const size_t batch_cap = static_cast<size_t>(std::min<uint64_t>(batch_reads, n_recs)) + 512;
      const size_t want_batches = static_cast<size_t>(std::min<uint64_t>(static_cast<uint64_t>(n_gpus) * (per_gpu + 1), (n_recs + batch_cap - 1) / batch_cap + static_cast<uint64_t>(n_gpus)));
      const int ends = paired ? 2 : 1;
      std::vector<std::unique_ptr<Slice>> sl(want_slices);
      std::vector<std::unique_ptr<Batch>> bt(want_batches);
      for (auto &x : bt) x.reset(new Batch);
      std::atomic<size_t> next{0};
      auto touch = [](char *q, size_t bytes) { for (size_t i = 0; i < bytes; i += 4096) q[i] = 0; };
      std::vector<std::thread> th;
      for (unsigned t = 0; t < n_host; ++t)
        th.emplace_back([&] {
          for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= want_slices + want_batches * 8) break;
            if (k < want_slices) {
              std::unique_ptr<Slice> x(new Slice);
              for (int e = 0; e < ends; ++e) {
                x->raw[e].reserve(slice_reads * rec_bytes + (1u << 16)); touch(x->raw[e].p, x->raw[e].cap);
                x->blob[e].reserve(slice_reads * (read_len + 2)); touch(x->blob[e].p, x->blob[e].cap);
                x->names[e].reserve(slice_reads + 16);
                x->off[e].reserve(slice_reads + 16);
                touch(reinterpret_cast<char *>(x->names[e].data()), (slice_reads + 16) * sizeof(NameRef));
                touch(reinterpret_cast<char *>(x->off[e].data()), (slice_reads + 16) * sizeof(uint64_t));
              }
              x->text.reserve(slice_reads * ends * 330); touch(x->text.p, x->text.cap);
              sl[k] = std::move(x);
            }
            else {  // a batch's arrays, one piece per task
              const size_t bi = (k - want_slices) / 8, piece = (k - want_slices) % 8;
              Batch &b = *bt[bi];
              const size_t n = batch_cap;
              const int e = static_cast<int>(piece & 1);
              if (e >= ends) continue;
              switch (piece >> 1) {
                case 0: b.blob[e].reserve(n * (read_len + 2)); touch(b.blob[e].p, b.blob[e].cap); break;
                case 1: b.off_bytes[e].reserve((n + 1) * 8); touch(b.off_bytes[e].p, b.off_bytes[e].cap); b.se[e].b.reserve(n * sizeof(abm_hit)); touch(b.se[e].b.p, b.se[e].b.cap); break;
                case 2: b.cig[e].b.reserve((4 * n + 1024) * 4); touch(b.cig[e].b.p, b.cig[e].b.cap); break;
                default: b.cig_off[e].b.reserve((n + 1) * 8); touch(b.cig_off[e].b.p, b.cig_off[e].b.cap);
                         if (paired && e == 0) { b.pairs.b.reserve(n * sizeof(abm_pair)); touch(b.pairs.b.p, b.pairs.b.cap); }
              }
            }
          }
        });
      for (auto &t : th) t.join();
      for (auto &x : sl) if (x) slice_pool.put(std::move(x));
      for (auto &x : bt) if (x) batch_pool.put(std::move(x));
    }
    host_prepare_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
  }
  const auto t_start = std::chrono::steady_clock::now();

  double busy_split = 0, busy_parse = 0, busy_map = 0, busy_format = 0, busy_write = 0;  // seconds, summed over threads
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  RawPool raw_pool;
  // ABM_CLI_TRACE=1: one line per pipeline event on stderr (milliseconds since the pipeline started)
  const bool tracing = std::getenv("ABM_CLI_TRACE") != nullptr;
  auto trace = [&](const char *what, uint64_t a, uint64_t b) {
    if (!tracing) return;
    std::fprintf(stderr, "[abm cli] t=%9.2f ms %-14s %llu %llu\n", since(t_start) * 1e3, what, static_cast<unsigned long long>(a),
                 static_cast<unsigned long long>(b));
  };
  auto fail = [&]() {
    std::lock_guard<std::mutex> lk(mu);
    if (!failure) failure = std::current_exception();
    wake_everyone();
  };
  // hands a cut slice to the parsers (blocks while too much is in flight)
  auto emit_slice = [&](std::unique_ptr<Slice> sl, size_t records) -> bool {
    std::unique_lock<std::mutex> lk(mu);
    cv_flow.wait(lk, [&] { return failure || reads_in_flight < max_reads_in_flight; });
    if (failure) return false;
    sl->g = n_slices++;
    trace("cut", sl->g, records);
    reads_in_flight += records;
    q_parse.push_back(std::move(sl));
    cv_parse.notify_one();
    return true;
  };

  // ---- cut, gzip (or non-regular) input: one inflating reader, slices carry their text
  auto cutter_stream = [&]() {
    try {
      RawSplitter s1(opt.reads[0]);
      std::unique_ptr<RawSplitter> s2;
      if (paired) s2.reset(new RawSplitter(opt.reads[1]));
      for (;;) {
        std::unique_ptr<Slice> sl = slice_pool.get();
        const auto t0 = now();
        sl->raw[0] = raw_pool.get();
        if (paired) sl->raw[1] = raw_pool.get();
        const uint64_t l1 = s1.next(slice_reads, sl->raw[0], sl->first_line[0]);
        uint64_t l2 = 0;
        if (paired) l2 = s2->next(slice_reads, sl->raw[1], sl->first_line[1]);
        const bool last = s1.exhausted() || (paired && s2->exhausted());
        if (l1 == 0 && (!paired || l2 == 0)) break;
        { std::lock_guard<std::mutex> lk(mu); busy_split += since(t0); }
        if (!emit_slice(std::move(sl), (l1 + 3) / 4)) break;
        if (last) break;
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    cut_done = true;
    wake_everyone();
  };

  // ---- cut, plain files: newline counts per chunk (parallel), then slice byte ranges (serial, cheap)
  struct ChunkInfo { uint64_t lines = 0; std::vector<uint32_t> marks; bool ready = false; };  // marks: offset just past every kMark-th newline
  const uint64_t kChunk = env_or("ABM_CLI_CHUNK_BYTES", 8u << 20), kMark = env_or("ABM_CLI_MARK_LINES", 1024);
  struct LineFile {
    int fd = -1;
    uint64_t size = 0, n_chunks = 0;
    std::vector<ChunkInfo> chunks;
    uint64_t next_chunk = 0;  // next chunk a counter thread takes
    bool ends_with_newline = true;
  };
  std::vector<LineFile> lf(plain_input ? opt.reads.size() : 0);
  for (size_t e = 0; e < lf.size(); ++e) {
    lf[e].fd = ::open(opt.reads[e].c_str(), O_RDONLY);
    if (lf[e].fd < 0) throw std::runtime_error("cannot open reads file: " + opt.reads[e]);
    struct stat sb;
    if (::fstat(lf[e].fd, &sb) != 0) throw std::runtime_error("cannot stat reads file: " + opt.reads[e]);
    lf[e].size = static_cast<uint64_t>(sb.st_size);
    lf[e].n_chunks = (lf[e].size + kChunk - 1) / kChunk;
    lf[e].chunks.resize(lf[e].n_chunks);
    if (lf[e].size) { char c = 0; if (::pread(lf[e].fd, &c, 1, static_cast<off_t>(lf[e].size - 1)) == 1) lf[e].ends_with_newline = c == '\n'; }
  }
  auto read_range = [&](int fd, const std::string &path, char *dst, uint64_t lo, uint64_t hi) {
    while (lo < hi) {
      const ssize_t got = ::pread(fd, dst, hi - lo, static_cast<off_t>(lo));
      if (got < 0) { if (errno == EINTR) continue; throw std::runtime_error("error reading " + path); }
      if (got == 0) throw std::runtime_error("unexpected end of " + path);
      dst += got; lo += static_cast<uint64_t>(got);
    }
  };
  auto counter = [&]() {  // counts the newlines of whole chunks, in file order per file, ahead of the cutter
    try {
      std::vector<char> buf(kChunk);
      for (;;) {
        size_t e = 0; uint64_t k = 0; bool got = false;
        {
          std::unique_lock<std::mutex> lk(mu);
          if (failure) break;
          // the file whose index is least advanced (both files of a pair are cut in step)
          size_t best = lf.size();
          for (size_t f = 0; f < lf.size(); ++f)
            if (lf[f].next_chunk < lf[f].n_chunks && (best == lf.size() || lf[f].next_chunk < lf[best].next_chunk)) best = f;
          if (best != lf.size()) { e = best; k = lf[e].next_chunk++; got = true; }
        }
        if (!got) break;
        const auto t0 = now();
        const uint64_t lo = k * kChunk, hi = std::min(lf[e].size, lo + kChunk);
        read_range(lf[e].fd, opt.reads[e], buf.data(), lo, hi);
        ChunkInfo ci;
        const char *p = buf.data(), *end = p + (hi - lo);
        uint64_t until_mark = kMark;
        while (p < end) {  // block counts vectorise; a block holding a mark is walked newline by newline
          const size_t blk = std::min<size_t>(static_cast<size_t>(end - p), 4096);
          uint32_t c = 0;
          for (size_t i = 0; i < blk; ++i) c += (p[i] == '\n');
          if (c < until_mark) { until_mark -= c; ci.lines += c; p += blk; continue; }
          const char *q = p, *bend = p + blk;
          while (q < bend) {
            const char *nl = static_cast<const char *>(std::memchr(q, '\n', static_cast<size_t>(bend - q)));
            if (!nl) break;
            ++ci.lines;
            q = nl + 1;
            if (--until_mark == 0) { ci.marks.push_back(static_cast<uint32_t>(q - buf.data())); until_mark = kMark; }
          }
          p = bend;
        }
        ci.ready = true;
        {
          std::lock_guard<std::mutex> lk(mu);
          lf[e].chunks[k] = std::move(ci);
          busy_split += since(t0);
        }
        cv_chunk.notify_all();
      }
    }
    catch (...) { fail(); }
  };
  // byte offset just past newline number `line` (1-based count of newlines) of file e; chunks up to the one
  // holding it must be ready.  cum[k] = newlines before chunk k.
  struct Cursor { uint64_t chunk = 0, cum = 0; };  // first chunk not yet passed, newlines before it
  auto offset_after_line = [&](size_t e, Cursor &cur, uint64_t line, uint64_t &off_out) -> bool {
    // returns false if the file has fewer newlines
    LineFile &F = lf[e];
    for (;;) {
      if (cur.chunk >= F.n_chunks) return false;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_chunk.wait(lk, [&] { return failure || F.chunks[cur.chunk].ready; });
        if (failure) throw std::runtime_error("aborted");
      }
      const ChunkInfo &ci = F.chunks[cur.chunk];
      if (cur.cum + ci.lines >= line) break;
      cur.cum += ci.lines;
      ++cur.chunk;
    }
    const ChunkInfo &ci = F.chunks[cur.chunk];
    const uint64_t local = line - cur.cum;          // the local-th newline of this chunk, 1-based
    const uint64_t mark = local / kMark;            // marks[m - 1] = offset past newline m * kMark
    uint64_t at = mark ? ci.marks[mark - 1] : 0, seen = mark * kMark;
    const uint64_t base = cur.chunk * kChunk, hi = std::min(F.size, base + kChunk);
    std::vector<char> buf;
    while (seen < local) {  // walk the <= kMark lines after the mark
      const uint64_t want = std::min<uint64_t>(hi - (base + at), 1u << 16);
      if (want == 0) return false;
      buf.resize(want);
      read_range(F.fd, opt.reads[e], buf.data(), base + at, base + at + want);
      const char *p = buf.data(), *end = p + want;
      while (p < end && seen < local) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!nl) { p = end; break; }
        ++seen;
        p = nl + 1;
      }
      at += static_cast<uint64_t>(p - buf.data());
    }
    off_out = base + at;
    return true;
  };
  auto cutter_plain = [&]() {
    try {
      Cursor cur[2];
      uint64_t lo[2] = {0, 0}, line = 0;
      bool done[2] = {false, false};
      const size_t nf = lf.size();
      for (;;) {
        std::unique_ptr<Slice> sl = slice_pool.get();
        const uint64_t target = line + 4 * slice_reads;  // newlines before the next slice
        uint64_t recs = 0;
        bool any = false;
        for (size_t e = 0; e < nf; ++e) {
          sl->first_line[e] = line;
          sl->byte_lo[e] = lo[e];
          uint64_t hi = 0;
          if (!done[e] && offset_after_line(e, cur[e], target, hi)) sl->byte_hi[e] = hi;
          else { sl->byte_hi[e] = lf[e].size; done[e] = true; }
          lo[e] = sl->byte_hi[e];
          any |= sl->byte_hi[e] > sl->byte_lo[e];
        }
        if (!any) break;
        recs = slice_reads;  // (the last slice may hold fewer; the figure only bounds memory in flight)
        const bool last = done[0] || (nf == 2 && done[1]);
        if (!emit_slice(std::move(sl), recs)) break;
        line = target;
        if (last) break;
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    cut_done = true;
    wake_everyone();
  };

  auto parser = [&]() {
    try {
      for (;;) {
        std::unique_ptr<Slice> sl;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_parse.wait(lk, [&] { return failure || !q_parse.empty() || cut_done; });
          if (failure || q_parse.empty()) break;
          sl = std::move(q_parse.front());
          q_parse.pop_front();
        }
        const auto t0 = now();
        for (int e = 0; e < (paired ? 2 : 1); ++e) {
          if (plain_input) {
            if (!sl->raw[e].p) sl->raw[e] = raw_pool.get();
            const uint64_t len = sl->byte_hi[e] - sl->byte_lo[e];
            sl->raw[e].reserve(len + 1);
            read_range(lf[e].fd, opt.reads[e], sl->raw[e].p, sl->byte_lo[e], sl->byte_hi[e]);
            sl->raw[e].n = len;
          }
          parse_raw(sl->raw[e], sl->first_line[e], opt.reads[e], sl->names[e], sl->blob[e], sl->off[e]);
        }
        if (paired && sl->names[0].size() != sl->names[1].size())
          throw std::runtime_error("paired-end batch sizes differ. Batch 1: " + std::to_string(sl->names[0].size()) +
                                   ", batch 2: " + std::to_string(sl->names[1].size()) +
                                   ". Are you sure your paired-end inputs have the same number of reads?");
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_parse += since(t0);
          const uint64_t g = sl->g;
          trace("parsed", g, static_cast<uint64_t>(since(t0) * 1e6));
          parsed[g] = std::move(sl);
          ++n_parsed;
          for (auto it = parsed.find(run_end); it != parsed.end(); it = parsed.find(run_end)) { run_reads += it->second->n(); ++run_end; }
        }
        cv_map.notify_all();
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    --parsers_live;
    cv_map.notify_all();
  };

  // single-end results leave the library slice by slice while the kernel runs (ABM_CLI_NO_STREAM=1: whole batches, as
  // the paired-end path takes them)
  const bool stream_slices = !paired && !opt.host_ceiling && !std::getenv("ABM_CLI_NO_STREAM");
  auto mapper = [&](int slot) {
    const int g = slot / per_gpu;
    abm_ctx *ctx = ctxs[slot];
    try {
      for (;;) {
        std::unique_ptr<Batch> owned = batch_pool.get();
        Batch *b = owned.get();
        {
          std::unique_lock<std::mutex> lk(mu);
          auto all_parsed = [&] { return cut_done && n_parsed == n_slices; };
          // How many reads this batch should hold.  A GPU's batches grow geometrically from the first (2 M reads, then
          // 4 M ... up to -batch): the device starts on the first million reads a few milliseconds into the run,
          // and each batch is parsed and ready by the time the one before it has been handed out on the device --
          // waiting for a full batch (or for the input's extent) left the device idle for 0.15 s after its first
          // batch (profiles/r03_cli_timeline_before.log).  Once the input's extent is known (cutting runs far ahead
          // of parsing) what is left is split evenly into batches of at most that size -- and into two even when one
          // would do, if each half still has a few million reads: a batch's output is formatted and written while
          // the next one is being mapped.
          // Batches whose results leave slice by slice (single-end) are sized the other way round at the end: a
          // slice is complete when its costliest read is, a fifth of a second into the kernel, so only a LONG last
          // kernel leaves time to format and write most of its output while it still runs -- 512 k reads first, then
          // four times as many per batch up to -batch, and what is left in as few batches as possible (10 M reads:
          // 0.5 M, 2 M, 7.5 M -- 0.737-0.796 s against 0.817-0.831 s with 1 M, 3 M, 6 M on the same box,
          // profiles/r03_exp_e2e_first_batch.log).
          auto target = [&]() -> size_t {
            size_t cap = batch_reads;
            const uint64_t grown = std::min<uint64_t>(gpu_batches[g], 10) * (stream_slices ? 2 : 1);
            if (first_batch_reads) cap = std::min<size_t>(batch_reads, std::max<size_t>(slice_reads, first_batch_reads << grown));
            if (!cut_done) return cap;
            const size_t left = static_cast<size_t>(n_slices - next_to_map) * slice_reads;
            size_t k = (left + cap - 1) / cap;
            if (k <= 1) k = (!stream_slices && left >= (1u << 22)) ? 2 : 1;
            return std::max<size_t>(slice_reads, (left + k - 1) / k);
          };
          cv_map.wait(lk, [&] { return failure || run_reads >= target() || (all_parsed() && (run_end > next_to_map || parsed.empty())); });
          if (failure || run_end == next_to_map) {
            // (the unused batch goes back to the pool: destroying it here would free its page-locked buffers -- a
            // device-wide wait and 0.1-0.2 s of unpinning -- inside the run's clock; the trace showed the run's end
            // waiting on exactly these two threads)
            lk.unlock();
            batch_pool.put(std::move(owned));
            break;
          }
          const size_t want = std::min(target(), std::max<size_t>(run_reads, 1));
          while (next_to_map < run_end) {
            auto it = parsed.find(next_to_map);
            if (!b->slices.empty() && b->n + it->second->n() > want) break;
            b->n += it->second->n();
            run_reads -= it->second->n();
            b->slices.push_back(std::move(it->second));
            parsed.erase(it);
            ++next_to_map;
          }
          b->seq = n_batches++;
          b->gpu = g;
          ++gpu_batches[g];
          gpu_reads[g] += b->n;
          b->slices_left = static_cast<int>(b->slices.size());
          // Reads of 44-46 bases see what earlier reads left in the reference's reused buffers:
          // the mapper looks for that among the reads handed over in the same call, so a batch is led by the
          // tail of the input before it -- from the last record whose reads are all longer than 46 bases on
          // (nearly always just that one record) -- whose results are dropped.
          b->carry[0] = carry[0];
          b->carry[1] = carry[1];
          {
            std::vector<std::string> next[2];
            bool closed = false;
            for (size_t si = b->slices.size(); si-- > 0 && !closed;) {
              const Slice &sl = *b->slices[si];
              for (size_t k = sl.n(); k-- > 0 && !closed;) {
                bool all_long = true;
                for (int e = 0; e < (paired ? 2 : 1); ++e) {
                  const size_t len = sl.off[e][k + 1] - sl.off[e][k];
                  next[e].emplace_back(sl.blob[e].data() + sl.off[e][k], len);
                  all_long &= len > 46;
                }
                closed = all_long;  // (no cap on the records: a heavily trimmed library has long runs of short ones)
              }
            }
            if (!closed)  // the whole batch had no such record: keep the older tail too
              for (int e = 0; e < (paired ? 2 : 1); ++e)
                for (size_t k = carry[e].size(); k-- > 0;) next[e].push_back(carry[e][k]);
            for (int e = 0; e < (paired ? 2 : 1); ++e) { std::reverse(next[e].begin(), next[e].end()); carry[e].swap(next[e]); }
          }
          live_batches.push_back(std::move(owned));
        }
        const size_t lead = b->carry[0].size();
        const size_t n = b->n + lead;
        const uint64_t seq_no = b->seq;  // (a batch whose slices were handed over during the call may be recycled before it returns)
        bool queued = false;
        const auto t0 = now();
        trace("batch formed", b->seq, n);
        // the slices' reads, concatenated as the C ABI takes them (a single slice with nothing to lead it is used in place)
        const bool one = b->slices.size() == 1 && lead == 0;
        for (int e = 0; e < (paired ? 2 : 1); ++e) {
          if (one) continue;
          size_t bytes = 0;
          for (const std::string &c : b->carry[e]) bytes += c.size();
          for (auto &sl : b->slices) bytes += sl->blob[e].size();
          b->blob[e].resize(bytes);
          b->off_bytes[e].resize((n + 1) * sizeof(uint64_t));
          uint64_t *boff = b->off_of(e);
          size_t at = 0, r = 0;
          for (const std::string &c : b->carry[e]) {
            std::memcpy(&b->blob[e][at], c.data(), c.size());
            boff[r++] = at;
            at += c.size();
          }
          // where each slice goes, then the copies on a few threads
          std::vector<size_t> s_at(b->slices.size()), s_r(b->slices.size());
          for (size_t k = 0; k < b->slices.size(); ++k) {
            s_at[k] = at; s_r[k] = r;
            at += b->slices[k]->blob[e].size();
            r += b->slices[k]->n();
          }
          boff[n] = at;
          auto copy_range = [&, e](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; ++k) {
              const Slice &sl = *b->slices[k];
              std::memcpy(&b->blob[e][s_at[k]], sl.blob[e].data(), sl.blob[e].size());
              const size_t m = sl.n();
              uint64_t *dst = boff + s_r[k];
              for (size_t i = 0; i < m; ++i) dst[i] = sl.off[e][i] + s_at[k];
            }
          };
          const size_t n_copy = std::min<size_t>(std::max<size_t>(1, n_host / 4), std::max<size_t>(1, b->slices.size() / 4));
          std::vector<std::thread> copiers;
          for (size_t t = 1; t < n_copy; ++t) copiers.emplace_back(copy_range, b->slices.size() * t / n_copy, b->slices.size() * (t + 1) / n_copy);
          copy_range(0, b->slices.size() / n_copy);
          for (auto &t : copiers) t.join();
        }
        {
          size_t base = lead;
          for (auto &sl : b->slices) { sl->batch = b; sl->base = base; base += sl->n(); }
        }
        const char *blob_p[2];
        const uint64_t *off_p[2];
        size_t blob_n[2];
        for (int e = 0; e < 2; ++e) {
          blob_p[e] = one ? b->slices[0]->blob[e].data() : b->blob[e].data();
          off_p[e] = one ? b->slices[0]->off[e].data() : b->off_of(e);
          blob_n[e] = one ? b->slices[0]->blob[e].size() : b->blob[e].size();
        }
        trace("batch ready", b->seq, n);
        if (n) {
          // a few CIGAR ops per read are typical; the worst case (read length + 2 each) is only
          // allocated if the first size turns out too small
          const uint64_t worst = std::max<uint64_t>(1, std::max(blob_n[0], blob_n[1]) + 2 * n);
          uint64_t cap = std::min<uint64_t>(worst, 4 * n + 1024);
          queued = false;
          for (;;) {
            int rc;
            if (!paired && opt.host_ceiling) {
              // diagnostic: what the pipeline around the mapper can carry.  Every read "maps" somewhere inside the first
              // chromosome with one mismatch and a single-op CIGAR; nothing is sent to the GPU.
              cap = std::max<uint64_t>(cap, n);
              b->se[0].resize(n); b->cig[0].resize(cap); b->cig_off[0].resize(n + 1);
              const uint32_t c0 = ch.starts.size() > 2 ? ch.starts[1] : 0, c1 = ch.starts.size() > 2 ? ch.starts[2] : 0;
              const uint32_t span = c1 > c0 + 70000 ? c1 - c0 - 66000 : 1;
              std::vector<std::thread> fill;
              for (unsigned t = 0; t < 8; ++t)
                fill.emplace_back([&, t] {
                  for (size_t i = n * t / 8; i < n * (t + 1) / 8; ++i) {
                    const uint32_t len = static_cast<uint32_t>(off_p[0][i + 1] - off_p[0][i]);
                    abm_hit h;
                    h.diffs = 1; h.flags = (i & 1) ? 0x10 : 0; h.pos = len ? c0 + static_cast<uint32_t>((i * 7919u) % span) : 0;
                    b->se[0][i] = h;
                    b->cig[0][i] = len << 4;
                    b->cig_off[0][i] = i;
                  }
                });
              for (auto &t : fill) t.join();
              b->cig_off[0][n] = n;
              rc = 0;
            }
            else if (!paired && stream_slices) {
              // results arrive slice by slice while the kernel runs: each slice takes its own copy inside the callback
              // and goes straight to the formatters
              std::vector<uint64_t> first(b->slices.size() + 1);
              first[0] = lead;
              for (size_t k = 0; k < b->slices.size(); ++k) first[k + 1] = first[k] + b->slices[k]->n();
              struct Taker {
                abm_ctx *ctx; Batch *b; const std::vector<uint64_t> *first; std::mutex *mu; std::deque<Slice *> *q;
                std::condition_variable *cv; int rc; std::string err;
              } taker{ctx, b, &first, &mu, &q_format, &cv_work, 0, std::string()};
              auto on_done = [](void *user, uint32_t s) {
                Taker &t = *static_cast<Taker *>(user);
                Slice &sl = *t.b->slices[s];
                const uint64_t lo = (*t.first)[s], hi = (*t.first)[s + 1], m = hi - lo;
                sl.own_se.resize(std::max<uint64_t>(m, 1));
                sl.own_cig_off.resize(m + 1);
                uint64_t room = 4 * m + 64;
                for (int attempt = 0; attempt < 2 && t.rc == 0; ++attempt) {
                  sl.own_cig.resize(room);
                  const int rc = abm_ctx_slice_results(t.ctx, lo, hi, sl.own_se.data(), sl.own_cig.data(), room, sl.own_cig_off.data());
                  if (rc == 0) break;
                  if (rc == ABM_ERR_CAPACITY && attempt == 0) { room = sl.own_cig_off[m]; continue; }
                  t.rc = rc;
                  t.err = abm_last_error();
                }
                sl.own = true;
                {
                  std::lock_guard<std::mutex> lk(*t.mu);
                  t.q->push_back(&sl);
                }
                t.cv->notify_one();
              };
              rc = abm_map_se_batch_sliced(ctx, se_mode, &par, n, blob_p[0], off_p[0], static_cast<uint32_t>(b->slices.size()),
                                           first.data(), on_done, &taker);
              if (rc == 0 && taker.rc != 0) throw std::runtime_error("taking a slice's results: " + taker.err);
              queued = true;
            }
            else if (!paired) {
              b->se[0].resize(n); b->cig[0].resize(cap); b->cig_off[0].resize(n + 1);
              rc = abm_map_se_batch(ctx, se_mode, &par, n, blob_p[0], off_p[0], b->se[0].data(),
                                    b->cig[0].data(), cap, b->cig_off[0].data());
            }
            else {
              b->pairs.resize(n); b->se[0].resize(n); b->se[1].resize(n);
              for (int e = 0; e < 2; ++e) { b->cig[e].resize(cap); b->cig_off[e].resize(n + 1); }
              rc = abm_map_pe_batch(ctx, pe_mode, &par, n, blob_p[0], off_p[0], blob_p[1], off_p[1], b->pairs.data(),
                                    b->se[0].data(), b->se[1].data(), b->cig[0].data(), b->cig_off[0].data(),
                                    b->cig[1].data(), b->cig_off[1].data(), cap);
            }
            if (rc == 0) break;
            if (rc == ABM_ERR_CAPACITY && cap < worst) { cap = worst; continue; }
            die_abm("mapping");
          }
        }
        else { b->cig_off[0].assign(1, 0); b->cig_off[1].assign(1, 0); }
        trace("batch mapped", seq_no, n);
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_map += since(t0);
          if (!queued)
            for (auto &sl : b->slices) q_format.push_back(sl.get());
        }
        cv_work.notify_all();
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    --mappers_live;
    cv_work.notify_all();
    cv_write.notify_all();
  };

  auto format_slice = [&](Slice &sl) {
    const Batch *b = sl.batch;
    t_bam = opt.bam;
    RawBuf &sam = sl.text;
    Stats3 &st = sl.stats;
    const size_t m = sl.n(), base = sl.base;
    sam.reserve(m * (paired ? 2 : 1) * 320);
    if (!paired) {
      // (a slice that took its own copy of the results indexes it by its own read numbers)
      const abm_hit *hits = sl.own ? sl.own_se.data() : b->se[0].data() + base;
      const uint32_t *cig_blob = sl.own ? sl.own_cig.data() : b->cig[0].data();
      const uint64_t *cig_off = sl.own ? sl.own_cig_off.data() : b->cig_off[0].data() + base;
      for (size_t k = 0; k < m; ++k) {
        abm_hit h = hits[k];
        const size_t len = sl.off[0][k + 1] - sl.off[0][k];
        const uint32_t *cg = cig_blob + cig_off[k];
        const size_t ncg = cig_off[k + 1] - cig_off[k];
        if (len && emit_se(sam, opt.ambig, h, ch, sl.names[0][k], sl.blob[0].data() + sl.off[0][k], len, cg, ncg) == UNMAPPED) h.pos = 0;
        st.s[0].tally(len == 0, h, opt.ambig, ref_len(cg, ncg));
      }
      return;
    }
    for (size_t k = 0; k < m; ++k) {
      const size_t i = base + k;
      abm_pair p = b->pairs[i];
      abm_hit h1 = b->se[0][i], h2 = b->se[1][i];
      const char *s1 = sl.blob[0].data() + sl.off[0][k], *s2 = sl.blob[1].data() + sl.off[1][k];
      const size_t l1 = sl.off[0][k + 1] - sl.off[0][k], l2 = sl.off[1][k + 1] - sl.off[1][k];
      const uint32_t *c1 = b->cig[0].data() + b->cig_off[0][i], *c2 = b->cig[1].data() + b->cig_off[1][i];
      const size_t nc1 = b->cig_off[0][i + 1] - b->cig_off[0][i], nc2 = b->cig_off[1][i + 1] - b->cig_off[1][i];
      // select_output, src/abismal.cpp:1073-1088
      const Outcome po = emit_pe(sam, opt.ambig, p, ch, sl.names[0][k], sl.names[1][k], s1, l1, s2, l2, c1, nc1, c2, nc2);
      const bool report = p.r1.pos != 0 && (opt.ambig || !(p.r1.flags & 0x100));
      bool pair_ok = report;
      if (!report || po == UNMAPPED) {
        if (po == UNMAPPED) { p.r1.pos = 0; p.r2.pos = 0; pair_ok = false; }
        if (emit_se(sam, opt.ambig, h1, ch, sl.names[0][k], s1, l1, c1, nc1) == UNMAPPED) h1.pos = 0;
        if (emit_se(sam, opt.ambig, h2, ch, sl.names[1][k], s2, l2, c2, nc2) == UNMAPPED) h2.pos = 0;
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

  // formats slices (any number of workers); a slice's place in the file is fixed once every earlier slice's size is known
  auto worker = [&]() {
    try {
      for (;;) {
        Slice *to_format = nullptr;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_work.wait(lk, [&] { return failure || !q_format.empty() || mappers_live == 0; });
          if (failure || q_format.empty()) break;
          to_format = q_format.front();
          q_format.pop_front();
        }
        const auto t0 = now();
        format_slice(*to_format);
        if (opt.bam) { RawBuf z; bgzf_compress(to_format->text, z); to_format->text.swap(z); }
        std::lock_guard<std::mutex> lk(mu);
        busy_format += since(t0);
        trace("formatted", to_format->g, static_cast<uint64_t>(since(t0) * 1e6));
        formatted[to_format->g] = to_format;
        // fix the place of every slice whose predecessors are all formatted
        bool placed = false;
        for (auto it = formatted.find(next_to_place); it != formatted.end(); it = formatted.find(next_to_place)) {
          place[next_to_place] = file_offset;
          file_offset += it->second->text.size();
          q_write.push_back(it->second);
          formatted.erase(it);
          ++next_to_place;
          placed = true;
        }
        if (placed) cv_write.notify_one();
      }
    }
    catch (...) { fail(); }
  };
  // ONE writer: slices leave in order, each with a single pwrite at its place (or write, into a pipe).  One thread
  // writing sequentially is what a tmpfs file takes fastest -- measured on the GPU box, 2 GB in 8 MB pieces: 7.9 GB/s
  // from one thread, 4.3 from four, 2.8 from sixteen, which contend for the file's lock (profiles/r03_write_probe.log).
  auto writer = [&]() {
    try {
      for (;;) {
        Slice *to_write = nullptr;
        uint64_t at = 0;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv_write.wait(lk, [&] { return failure || !q_write.empty() || (mappers_live == 0 && cut_done && slices_written == n_slices); });
          if (failure || q_write.empty()) break;
          to_write = q_write.front(); q_write.pop_front(); at = place[to_write->g]; place.erase(to_write->g);
        }
        const auto t0 = now();
        write_all(to_write->text.data(), to_write->text.size(), at);
        std::unique_ptr<Batch> done_batch;
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_write += since(t0);
          trace("written", to_write->g, static_cast<uint64_t>(since(t0) * 1e6));
          Batch *b = to_write->batch;
          total_records += to_write->n();
          for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 6; ++j) gpu_stats[b->gpu].s[k].v[j] += to_write->stats.s[k].v[j];
          reads_in_flight -= std::min<size_t>(reads_in_flight, slice_reads);
          ++slices_written;
          // the batch goes when its last slice is written; its slices and its own buffers are recycled
          if (--b->slices_left == 0) {
            for (auto it = live_batches.begin(); it != live_batches.end(); ++it)
              if (it->get() == b) {
                for (auto &sl : (*it)->slices) slice_pool.put(std::move(sl));
                done_batch = std::move(*it);
                live_batches.erase(it);
                break;
              }
          }
          cv_flow.notify_one();
          if (slices_written == n_slices) cv_write.notify_all();
        }
        if (done_batch) batch_pool.put(std::move(done_batch));
      }
    }
    catch (...) { fail(); }
  };

  std::vector<std::thread> threads;
  parsers_live = static_cast<int>(n_host);
  mappers_live = n_gpus * per_gpu;
  if (plain_input) {
    const unsigned n_count = std::max(1u, std::min(n_host, 16u));
    for (unsigned t = 0; t < n_count; ++t) threads.emplace_back(counter);
    threads.emplace_back(cutter_plain);
  }
  else threads.emplace_back(cutter_stream);
  for (unsigned t = 0; t < n_host; ++t) threads.emplace_back(parser);
  for (int slot = 0; slot < n_gpus * per_gpu; ++slot) threads.emplace_back(mapper, slot);
  for (unsigned t = 0; t < n_host; ++t) threads.emplace_back(worker);
  threads.emplace_back(writer);
  {
    size_t k = 0;
    for (auto &t : threads) {  // (a join that has to wait shows in the trace: which thread the run's end hung on)
      const auto tj = now();
      t.join();
      if (since(tj) > 2e-3) trace("waited on join", k, static_cast<uint64_t>(since(tj) * 1e6));
      ++k;
    }
  }
  trace("threads joined", threads.size(), 0);
  for (LineFile &F : lf) if (F.fd >= 0) ::close(F.fd);
  if (failure) std::rethrow_exception(failure);
  if (opt.bam) {  // BGZF end-of-file marker
    static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    write_all(reinterpret_cast<const char *>(eof_block), 28, file_offset);
    file_offset += 28;
  }
  out_closer.fd = -1;
  if (::close(out_fd) != 0) throw std::runtime_error("failed writing output file: " + opt.out);
  trace("output closed", file_offset, 0);
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
       << ", \"index_load_s\": " << index_load_s << ", \"host_prepare_s\": " << host_prepare_s << ", \"gpus\": " << n_gpus << ", \"mappers_per_gpu\": " << per_gpu
       << ", \"host_threads\": " << n_host << ", \"batch_reads\": " << batch_reads << ", \"host_ceiling\": " << (opt.host_ceiling ? "true" : "false")
       << ", \"batches_per_gpu\": [";
    for (int g = 0; g < n_gpus; ++g) tj << (g ? ", " : "") << gpu_batches[g];
    tj << "], \"reads_per_gpu\": [";
    for (int g = 0; g < n_gpus; ++g) tj << (g ? ", " : "") << gpu_reads[g];
    tj << "], \"busy_s\": {\"split\": " << busy_split
       << ", \"parse\": " << busy_parse << ", \"map\": " << busy_map << ", \"format\": " << busy_format << ", \"write\": "
       << busy_write << "}}\n";
  }
  if (opt.verbose)
    std::cerr << "[abismal-amd] " << total_records << (paired ? " pairs" : " reads") << " on " << n_gpus << " GPU(s) in "
              << secs << " s (" << (paired ? 2 : 1) * total_records / secs << " reads/s incl. host I/O)\n"
              << "[abismal-amd] busy seconds: split " << busy_split << ", parse " << busy_parse << " (" << n_host
              << " threads), map " << busy_map << " (" << n_gpus * per_gpu << " threads), format " << busy_format << " ("
              << n_host << " threads), write " << busy_write << "\n";
  if (opt.verbose)
    for (int g = 0; g < n_gpus; ++g)
      std::cerr << "[abismal-amd] GPU " << g << ": " << gpu_batches[g] << " batches, " << gpu_reads[g] << (paired ? " pairs\n" : " reads\n");
  // Single-end reads of any length the reference takes are mapped (longer ones stop the run while the input is parsed,
  // with the reference's message).  Pairs: the paired-end kernels take ends of up to 1024 bases; a pair with a longer end
  // was written unmapped, which the reference would not have done -- so the run fails unless -skip-long accepts it.
  uint64_t too_long = 0;
  for (abm_ctx *c : ctxs) too_long += abm_ctx_reads_too_long(c);
  if (too_long)
    std::cerr << "[abismal-amd] " << (opt.skip_long ? "warning: " : "error: ") << too_long << (paired ? " pairs with an end longer than 1024 bases"
                                                                                                       : " reads beyond the supported length")
              << " were not mapped (written as unmapped" << (opt.skip_long ? ")\n" : "); -skip-long accepts this\n");
  for (abm_ctx *c : ctxs) abm_ctx_destroy(c);
  abm_index_close(ix);
  return too_long && !opt.skip_long ? EXIT_FAILURE : EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char **argv) {
  // batches of several mapper threads overlap on a GPU, each on its own stream; the HIP runtime folds streams onto 4
  // hardware queues unless told otherwise before it starts (paired-end: 1.8 -> 3.0 M reads/s with 16, bench.py --pe)
  ::setenv("GPU_MAX_HW_QUEUES", "16", 0);
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
