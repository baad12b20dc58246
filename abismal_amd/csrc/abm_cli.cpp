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
#include <sched.h>
#include <signal.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <malloc.h>
#include <sys/mman.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
std::atomic<uint64_t> g_pinned_bytes{0};  // page-locked memory the run has asked the library for (batches' read buffers)
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
      g_pinned_bytes += want - cap;
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
  uint64_t g = 0;                    // slice number within its region = output order
  int region = 0;                    // which contiguous share of the input (= which output file) it belongs to
  int node = 0;                      // NUMA node its buffers were first touched on: where it is parsed and formatted
  uint64_t place = 0;                // its text's offset in the region's file, once every earlier slice's size is known
  std::vector<uint32_t> tail;        // records that can still be a ghost-bit source for later reads (ghost_tail)
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
  bool virt = false;                 // virtual GPUs: own_* are filled in by the formatter (made-up hits)
  PodVec<abm_hit> own_se;
  PodVec<uint32_t> own_cig;
  PodVec<uint64_t> own_cig_off;
  // ... and, when the kernel wrote the reads' SAM text itself (abm_ctx_set_sam_tails), every read's line after QNAME:
  // lengths (0 = no record, 0xFFFFFFFF = format it here) and the text, one after the other
  bool has_tails = false;
  PodVec<uint32_t> tail_len;
  RawBuf tail_text;
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
    s->virt = false;
    s->tail.clear();
    std::lock_guard<std::mutex> lk(mu);
    if (free_list.size() < 1024) free_list.push_back(std::move(s));
  }
};

bool g_pin_batches = true;  // batch blobs in page-locked memory (not with virtual GPUs: it comes from the HIP runtime)
struct Batch {
  Batch() { for (int e = 0; e < 2; ++e) blob[e].pinned = off_bytes[e].pinned = g_pin_batches; }
  uint64_t seq = 0;
  int gpu = 0;
  int node = 0;                      // NUMA node of its GPU: the pool it returns to
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

// What a read of 44-46 bases finds past its end comes, position by position, from the nearest EARLIER
// read that is longer than that position -- up to 64 positions out, so a read of kGhostReach = 46 + 64 bases hides
// everything before it, and reads the reference never preps (shorter than the index's minimum) leave nothing.
// ghost_tail: of n records (off[e][k], off[e][k + 1]: read k of end e), scanning backwards, those that are longer in
// some end than every record after them, until all ends have reached kGhostReach -- the only records of this input
// that can still be such a source for reads that come later.  Indices in descending order; at most 67 per end.
constexpr uint32_t kGhostReach = 110;
inline uint32_t ghost_len(uint64_t len) { return len < g_min_read_len ? 0u : static_cast<uint32_t>(std::min<uint64_t>(len, kGhostReach)); }
// (reach: how far each end is covered by the records after these n -- a scan that continues further back in the input
// passes the same array on; a fresh scan starts from ghost_reach_start)
inline void ghost_reach_start(uint32_t reach[2], int ends) { reach[0] = 0; reach[1] = ends == 2 ? 0u : kGhostReach; }
inline bool ghost_closed(const uint32_t reach[2]) { return reach[0] >= kGhostReach && reach[1] >= kGhostReach; }
std::vector<uint32_t> ghost_tail(const std::vector<uint64_t> *off, size_t n, int ends, uint32_t reach[2]) {
  std::vector<uint32_t> out;
  for (size_t k = n; k-- > 0 && (reach[0] < kGhostReach || reach[1] < kGhostReach);) {
    bool raises = false;
    for (int e = 0; e < ends; ++e) {
      const uint32_t len = ghost_len(off[e][k + 1] - off[e][k]);
      if (len > reach[e]) { raises = true; reach[e] = len; }
    }
    if (raises) out.push_back(static_cast<uint32_t>(k));
  }
  return out;
}
std::vector<uint32_t> ghost_tail(const std::vector<uint64_t> *off, size_t n, int ends) {
  uint32_t reach[2];
  ghost_reach_start(reach, ends);
  return ghost_tail(off, n, ends, reach);
}

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

void parse_raw(const char *text, size_t text_n, uint64_t first_line, const std::string &path, std::vector<NameRef> &names,
               RawBuf &blob, std::vector<uint64_t> &off) {
  names.clear(); blob.clear(); off.assign(1, 0);
  blob.reserve(text_n / 2);
  names.reserve(text_n / 200 + 16);
  off.reserve(text_n / 200 + 16);
  const char *p = text, *end = p + text_n;
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
void parse_raw(const RawBuf &raw, uint64_t first_line, const std::string &path, std::vector<NameRef> &names,
               RawBuf &blob, std::vector<uint64_t> &off) {
  parse_raw(raw.p, raw.n, first_line, path, names, blob, off);
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
// SEQ of a record: the tables above applied to a whole read.  A read is almost always upper-case A, C, G, T, N, so the
// forward form is a copy wherever 32 bytes at a time are nothing else, and the reverse-complement form -- A <-> T,
// C <-> G, everything else N, back to front -- is four compares and blends per 32 bytes; both fall back to the tables
// for a read's last bytes and for anything unusual, and are the tables themselves on a CPU without AVX2.  (SEQ was the
// costliest field of a line: a table look-up per base.)
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void seq_forward_avx2(char *w, const char *s, size_t n) {
  size_t i = 0;
  const __m256i a = _mm256_set1_epi8('A'), c = _mm256_set1_epi8('C'), g = _mm256_set1_epi8('G'), t = _mm256_set1_epi8('T'), nn = _mm256_set1_epi8('N');
  for (; i + 32 <= n; i += 32) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, a), _mm256_cmpeq_epi8(v, c)),
                                                       _mm256_or_si256(_mm256_cmpeq_epi8(v, g), _mm256_cmpeq_epi8(v, t))), _mm256_cmpeq_epi8(v, nn));
    if (static_cast<uint32_t>(_mm256_movemask_epi8(ok)) == 0xFFFFFFFFu) _mm256_storeu_si256(reinterpret_cast<__m256i *>(w + i), v);
    else for (size_t k = i; k < i + 32; ++k) w[k] = kSeq.fwd[static_cast<unsigned char>(s[k])];
  }
  for (; i < n; ++i) w[i] = kSeq.fwd[static_cast<unsigned char>(s[i])];
}
__attribute__((target("avx2"))) static void seq_revcomp_avx2(char *w, const char *s, size_t n) {
  // w[i] = rc[s[n - 1 - i]]: 32 bytes from the back of s at a time, reversed, then mapped
  const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  const __m256i a = _mm256_set1_epi8('A'), c = _mm256_set1_epi8('C'), g = _mm256_set1_epi8('G'), t = _mm256_set1_epi8('T'), nn = _mm256_set1_epi8('N');
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + n - 32 - i));
    v = _mm256_permute2x128_si256(_mm256_shuffle_epi8(v, rev), _mm256_shuffle_epi8(v, rev), 0x01);  // bytes reversed across the whole register
    __m256i o = nn;
    o = _mm256_blendv_epi8(o, t, _mm256_cmpeq_epi8(v, a));
    o = _mm256_blendv_epi8(o, g, _mm256_cmpeq_epi8(v, c));
    o = _mm256_blendv_epi8(o, c, _mm256_cmpeq_epi8(v, g));
    o = _mm256_blendv_epi8(o, a, _mm256_cmpeq_epi8(v, t));
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(w + i), o);
  }
  for (; i < n; ++i) w[i] = kSeq.rc[static_cast<unsigned char>(s[n - 1 - i])];
}
static const bool kHaveAvx2 = __builtin_cpu_supports("avx2");
static const bool g_scalar_seq = std::getenv("ABM_CLI_SCALAR_SEQ") != nullptr;  // (tests: the table form on a CPU that has AVX2)
#else
static const bool kHaveAvx2 = false;
#endif
inline void put_seq(char *w, const char *s, size_t n, bool rc) {
#if defined(__x86_64__)
  if (kHaveAvx2 && !g_scalar_seq) { if (rc) seq_revcomp_avx2(w, s, n); else seq_forward_avx2(w, s, n); return; }
#endif
  if (rc) for (size_t i = 0; i < n; ++i) w[i] = kSeq.rc[static_cast<unsigned char>(s[n - 1 - i])];
  else for (size_t i = 0; i < n; ++i) w[i] = kSeq.fwd[static_cast<unsigned char>(s[i])];
}

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
// digits of v at w, returns one past them
inline char *write_uint(char *w, uint64_t v) {
  char buf[24];
  int k = 24;
  do { buf[--k] = static_cast<char>('0' + v % 10); v /= 10; } while (v);
  std::memcpy(w, buf + k, static_cast<size_t>(24 - k));
  return w + (24 - k);
}
template <class S> void put_record(S &o, const Chroms &ch, const Record &r) {
  if (t_bam) { put_bam_record(o, r); return; }
  // one reservation for the whole line (its longest possible form), then plain pointer writes: a line is a dozen short
  // fields, and appending them one by one through the buffer's capacity checks was a third of the formatting time
  const std::string &chrom = ch.names[r.tid + 1];
  const std::string *mate = r.mtid < 0 || r.mtid == r.tid ? nullptr : &ch.names[r.mtid + 1];
  const size_t at0 = o.size();
  o.resize(at0 + r.name->n + chrom.size() + (mate ? mate->size() : 1) + r.n_cig * 12 + r.n_seq + 128);
  char *w = &o[at0];
  std::memcpy(w, r.name->p, r.name->n); w += r.name->n; *w++ = '\t';
  w = write_uint(w, r.flag); *w++ = '\t';
  std::memcpy(w, chrom.data(), chrom.size()); w += chrom.size(); *w++ = '\t';
  w = write_uint(w, static_cast<uint64_t>(r.pos) + 1);
  std::memcpy(w, "\t255\t", 5); w += 5;
  for (size_t i = 0; i < r.n_cig; ++i) { w = write_uint(w, r.cig[i] >> 4); *w++ = "MIDNSHP=XB"[std::min<uint32_t>(r.cig[i] & 15u, 9)]; }
  *w++ = '\t';
  if (r.mtid < 0) { std::memcpy(w, "*\t0\t", 4); w += 4; }
  else {
    if (!mate) *w++ = '=';
    else { std::memcpy(w, mate->data(), mate->size()); w += mate->size(); }
    *w++ = '\t'; w = write_uint(w, static_cast<uint64_t>(r.mpos) + 1); *w++ = '\t';
  }
  if (r.tlen < 0) { *w++ = '-'; w = write_uint(w, static_cast<uint64_t>(-static_cast<int64_t>(r.tlen))); }
  else w = write_uint(w, static_cast<uint64_t>(r.tlen));
  *w++ = '\t';
  put_seq(w, r.seq, r.n_seq, r.rc);
  w += r.n_seq;
  std::memcpy(w, "\t*\tNM:i:", 8); w += 8;
  if (r.nm < 0) { *w++ = '-'; w = write_uint(w, static_cast<uint64_t>(-static_cast<int64_t>(r.nm))); }
  else w = write_uint(w, static_cast<uint64_t>(r.nm));
  std::memcpy(w, "\tCV:A:", 6); w += 6;
  *w++ = r.cv; *w++ = '\n';
  o.resize(static_cast<size_t>(w - &o[0]));
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
// 4-bit BAM base codes of what SEQ shows (kSeq.fwd / kSeq.rc, then htslib's seq_nt16_table)
struct Seq4Tables {
  unsigned char fwd[256], rc[256];
  Seq4Tables() {
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    for (int c = 0; c < 256; ++c) {
      const char *qf = std::strchr(nt16, kSeq.fwd[c]), *qr = std::strchr(nt16, kSeq.rc[c]);
      fwd[c] = static_cast<unsigned char>(qf && kSeq.fwd[c] ? qf - nt16 : 15);
      rc[c] = static_cast<unsigned char>(qr && kSeq.rc[c] ? qr - nt16 : 15);
    }
  }
};
const Seq4Tables kSeq4;
template <class S> void put_bam_record(S &o, const Record &r) {
  // (one reservation for the record, then pointer writes, as for the SAM line)
  const size_t start = o.size();
  const size_t l_seq = r.n_seq, packed = (l_seq + 1) / 2;
  o.resize(start + 36 + r.name->n + 1 + 4 * r.n_cig + packed + l_seq + 16);
  unsigned char *w = reinterpret_cast<unsigned char *>(&o[start]);
  auto le32w = [&](uint32_t v) { w[0] = static_cast<unsigned char>(v); w[1] = static_cast<unsigned char>(v >> 8); w[2] = static_cast<unsigned char>(v >> 16); w[3] = static_cast<unsigned char>(v >> 24); w += 4; };
  auto le16w = [&](uint32_t v) { w[0] = static_cast<unsigned char>(v); w[1] = static_cast<unsigned char>(v >> 8); w += 2; };
  unsigned char *const size_at = w;
  le32w(0);  // block_size, patched below
  le32w(static_cast<uint32_t>(r.tid));
  le32w(r.pos);
  const uint32_t rl = ref_len(r.cig, r.n_cig);
  *w++ = static_cast<unsigned char>(r.name->n + 1);
  *w++ = 255;
  le16w(static_cast<uint32_t>(reg2bin(r.pos, static_cast<int64_t>(r.pos) + (rl ? rl : 1))));
  le16w(static_cast<uint32_t>(r.n_cig));
  le16w(r.flag);
  le32w(static_cast<uint32_t>(l_seq));
  le32w(static_cast<uint32_t>(r.mtid));
  le32w(r.mtid < 0 ? 0xFFFFFFFFu : r.mpos);
  le32w(static_cast<uint32_t>(r.tlen));
  std::memcpy(w, r.name->p, r.name->n); w += r.name->n; *w++ = 0;
  for (size_t i = 0; i < r.n_cig; ++i) le32w(r.cig[i]);
  const unsigned char *s = reinterpret_cast<const unsigned char *>(r.seq);
  if (r.rc)
    for (size_t i = 0; i < l_seq; i += 2)
      *w++ = static_cast<unsigned char>((kSeq4.rc[s[l_seq - 1 - i]] << 4) | (i + 1 < l_seq ? kSeq4.rc[s[l_seq - 2 - i]] : 0));
  else
    for (size_t i = 0; i < l_seq; i += 2)
      *w++ = static_cast<unsigned char>((kSeq4.fwd[s[i]] << 4) | (i + 1 < l_seq ? kSeq4.fwd[s[i + 1]] : 0));
  std::memset(w, 0xFF, l_seq); w += l_seq;
  *w++ = 'N'; *w++ = 'M';  // bam_aux_update_int: smallest type that holds the value
  if (r.nm >= 0 && r.nm <= 255) { *w++ = 'C'; *w++ = static_cast<unsigned char>(r.nm); }
  else if (r.nm >= 0) { *w++ = 'S'; le16w(static_cast<uint32_t>(r.nm)); }
  else if (r.nm >= -128) { *w++ = 'c'; *w++ = static_cast<unsigned char>(r.nm); }
  else { *w++ = 's'; le16w(static_cast<uint32_t>(static_cast<uint16_t>(static_cast<int16_t>(r.nm)))); }
  *w++ = 'C'; *w++ = 'V'; *w++ = 'A'; *w++ = static_cast<unsigned char>(r.cv);
  const size_t end = static_cast<size_t>(reinterpret_cast<char *>(w) - &o[0]);
  const uint32_t bs = static_cast<uint32_t>(end - start - 4);
  size_at[0] = static_cast<unsigned char>(bs); size_at[1] = static_cast<unsigned char>(bs >> 8); size_at[2] = static_cast<unsigned char>(bs >> 16); size_at[3] = static_cast<unsigned char>(bs >> 24);
  o.resize(end);
}
// raw bytes -> BGZF blocks (each an independent gzip member with the BC extra field)
int g_bgzf_level = 1;  // deflate level of BAM output (-z): 1 = the fast encoder below, 0 = stored, 2..9 = zlib; decoded content is the same at every level
// ---- a fast deflate for BGZF blocks (-z 1, the default) ---------------------------------------------------------------
// zlib at level 1 costs 1.15 us of CPU per 100-base record (BAM through 16 CPUs: 13.6 M reads/s, profiles/r04_host_ceiling.log)
// -- more than everything else the host does per read, six times over.  A BAM record stream is an easy input: runs (the
// 0xFF of absent qualities), fields repeated from the record before, 4-bit sequence that does not compress.  This encoder
// takes one greedy match per position from a single-probe hash of the last occurrence of each 4-byte string and writes
// ONE block with the fixed Huffman code (RFC 1951 3.2.6: no trees to build or ship); whatever inflates it gets the same
// bytes back.  Returns the compressed size, or 0 if `cap` does not suffice (the caller then stores the block).
struct FastDeflate {
  // fixed code, bit-reversed for the LSB-first stream: literal / length symbol -> (code, bits); length -> (symbol, extra)
  uint16_t lit_code[288];
  uint8_t lit_bits[288];
  uint16_t len_sym[259];
  uint8_t len_extra_bits[259];
  uint16_t len_extra_val[259];
  uint8_t dist_sym_small[513];  // distances 1..512 -> symbol; beyond: by the distance's top bits
  static uint32_t rev(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1u); v >>= 1; } return r; }
  FastDeflate() {
    for (int s = 0; s < 288; ++s) {
      uint32_t code; int bits;
      if (s < 144) { code = 0x30 + s; bits = 8; }
      else if (s < 256) { code = 0x190 + (s - 144); bits = 9; }
      else if (s < 280) { code = s - 256; bits = 7; }
      else { code = 0xC0 + (s - 280); bits = 8; }
      lit_code[s] = static_cast<uint16_t>(rev(code, bits));
      lit_bits[s] = static_cast<uint8_t>(bits);
    }
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (int len = 3; len <= 258; ++len) {
      int k = 28;
      while (base[k] > len) --k;
      if (len == 258) k = 28;
      len_sym[len] = static_cast<uint16_t>(257 + k);
      len_extra_bits[len] = extra[k];
      len_extra_val[len] = static_cast<uint16_t>(len - base[k]);
    }
    for (int d = 1; d <= 512; ++d) dist_sym_small[d] = static_cast<uint8_t>(dist_symbol_slow(static_cast<uint32_t>(d)));
  }
  static int dist_symbol_slow(uint32_t d) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    int k = 29;
    while (base[k] > d) --k;
    return k;
  }
  size_t operator()(const unsigned char *src, size_t n, unsigned char *dst, size_t cap, uint16_t *table /*[1 << 13], zeroed by this call*/) const {
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dextra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (n > 0xFFFF || cap < 16) return 0;
    std::memset(table, 0, sizeof(uint16_t) << 13);
    uint64_t acc = 0;
    int nbits = 0;
    unsigned char *out = dst, *const out_end = dst + cap - 16;
    auto put = [&](uint32_t v, int b) {
      acc |= static_cast<uint64_t>(v) << nbits;
      nbits += b;
      if (nbits >= 32) { std::memcpy(out, &acc, 4); out += 4; acc >>= 32; nbits -= 32; }
    };
    put(1, 1);  // BFINAL
    put(1, 2);  // BTYPE = 01, fixed Huffman
    size_t i = 0;
    const size_t last_hashable = n >= 4 ? n - 4 : 0;
    while (i < n) {
      if (out > out_end) return 0;
      size_t mlen = 0, mdist = 0;
      if (n >= 4 && i <= last_hashable) {
        uint32_t w;
        std::memcpy(&w, src + i, 4);
        const uint32_t h = (w * 2654435761u) >> 19;
        const size_t cand = table[h];  // position + 1 of the last string with this hash; 0 = none
        table[h] = static_cast<uint16_t>(i + 1);
        if (cand != 0) {
          const size_t p = cand - 1;
          uint32_t v;
          std::memcpy(&v, src + p, 4);
          if (v == w && i - p <= 32768) {
            const size_t lim = std::min<size_t>(258, n - i);
            size_t l = 4;
            while (l + 8 <= lim) {
              uint64_t a, b;
              std::memcpy(&a, src + p + l, 8);
              std::memcpy(&b, src + i + l, 8);
              if (a != b) { l += static_cast<size_t>(__builtin_ctzll(a ^ b) >> 3); break; }
              l += 8;
            }
            if (l + 8 > lim) while (l < lim && src[p + l] == src[i + l]) ++l;
            mlen = std::min(l, lim);
            mdist = i - p;
          }
        }
      }
      if (mlen >= 4) {
        const uint32_t ls = len_sym[mlen];
        put(lit_code[ls], lit_bits[ls]);
        if (len_extra_bits[mlen]) put(len_extra_val[mlen], len_extra_bits[mlen]);
        const int ds = mdist <= 512 ? dist_sym_small[mdist] : dist_symbol_slow(static_cast<uint32_t>(mdist));
        put(rev(static_cast<uint32_t>(ds), 5), 5);
        if (dextra[ds]) put(static_cast<uint32_t>(mdist - dbase[ds]), dextra[ds]);
        // (the strings inside the match are not entered into the table: the next record repeats this one's fields at
        // the positions where matches begin)
        i += mlen;
      }
      else {
        put(lit_code[src[i]], lit_bits[src[i]]);
        ++i;
      }
    }
    put(lit_code[256], lit_bits[256]);  // end of block
    while (nbits > 0) { if (out >= dst + cap) return 0; *out++ = static_cast<unsigned char>(acc); acc >>= 8; nbits -= 8; }
    return static_cast<size_t>(out - dst);
  }
};
const FastDeflate kFastDeflate;

// One deflate state and one block buffer per thread, reset per block: deflateInit2 allocates a quarter of a megabyte,
// and a hundred formatter threads doing that once per 64 KB block spent six times their compression time waiting on
// the allocator (profiles/r04_host_ceiling.log: -B busy 384 s for 60 s of CPU).
struct BgzfDeflater {
  z_stream zs;
  bool live = false;
  int level = -2;
  std::vector<unsigned char> buf;
  ~BgzfDeflater() { if (live) deflateEnd(&zs); }
  void prepare(int want_level) {
    if (live && level == want_level) { deflateReset(&zs); return; }
    if (live) deflateEnd(&zs);
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, want_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    live = true;
    level = want_level;
  }
};
template <class A, class B> void bgzf_compress(const A &raw, B &out) {
  constexpr size_t kBlock = 0xff00;
  thread_local BgzfDeflater d;
  if (d.buf.empty()) d.buf.resize(compressBound(kBlock) + 64);
  thread_local std::vector<uint16_t> hash_table(size_t(1) << 13);
  for (size_t at = 0; at < raw.size(); at += kBlock) {
    const size_t len = std::min(kBlock, raw.size() - at);
    size_t clen = 0;
    if (g_bgzf_level == 1)  // the fast encoder (a block it cannot fit -- incompressible input -- goes through zlib, stored)
      clen = kFastDeflate(reinterpret_cast<const unsigned char *>(raw.data() + at), len, d.buf.data(), std::min<size_t>(d.buf.size(), 0xFFFF - 26), hash_table.data());
    if (clen == 0) {
      d.prepare(g_bgzf_level == 1 ? 0 : g_bgzf_level);
      z_stream &zs = d.zs;
      zs.next_in = reinterpret_cast<Bytef *>(const_cast<char *>(raw.data() + at));
      zs.avail_in = static_cast<uInt>(len);
      zs.next_out = d.buf.data();
      zs.avail_out = static_cast<uInt>(d.buf.size());
      if (deflate(&zs, Z_FINISH) != Z_STREAM_END) throw std::runtime_error("deflate failed");
      clen = zs.total_out;
    }
    const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(raw.data() + at), static_cast<uInt>(len)));
    static const unsigned char head[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    out.append(reinterpret_cast<const char *>(head), 12);
    out.append("BC", 2); put_le16(out, 2); put_le16(out, static_cast<uint16_t>(clen + 25));
    out.append(reinterpret_cast<const char *>(d.buf.data()), clen);
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
  int window_records = -1;   // -window-records L: the index's window records serve reads of up to L bases (0 = none; default: the
                             // longest of the input's first reads -- abm_index_set_window_records)
  bool skip_long = false;     // -skip-long: reads the library reports as beyond its supported length are written unmapped
                              // (and counted in the warning) instead of failing the run
  bool host_ceiling = false;  // -host-ceiling / -virtual-gpus N (diagnostic): no device and no mapping call; every "GPU" hands
                              // back made-up hits at once, so that count -> cut -> parse -> deal -> format -> write run at the
                              // rate the host can carry around N GPUs (single-end input)
  std::vector<int> devices;   // -devices a,b,...: GPU g of the run is device ordinal devices[g]; an ordinal may repeat
                              // (replicas of the sharding on one device)
  int out_parts = 1;          // -out-parts R: the input's R contiguous shares mapped side by side, each into <out>.partNNN
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
    else if (k == "virtual-gpus") { o.host_ceiling = true; o.gpus = std::stoi(need(i)); }
    else if (k == "devices") {
      const std::string v = need(i);
      for (size_t at = 0; at < v.size();) {
        size_t used = 0;
        o.devices.push_back(std::stoi(v.substr(at), &used));
        at += used;
        if (at < v.size() && v[at] == ',') ++at;
      }
    }
    else if (k == "out-parts") o.out_parts = std::stoi(need(i));
    else if (k == "skip-long") o.skip_long = true;
    else if (k == "window-records") o.window_records = std::max(0, std::stoi(need(i)));
    else if (k == "seed-ext") { const std::string v = need(i); if (std::sscanf(v.c_str(), "%d,%d", &o.ext2, &o.ext3) != 2) throw std::runtime_error("-seed-ext wants two numbers: a,b"); }
    else if (k == "timing") o.timing = need(i);  // JSON: reads, seconds (first batch submitted -> last byte written), stage busy times
    else if (k == "z" || k == "bam-level") g_bgzf_level = std::max(0, std::min(9, std::stoi(need(i))));
    else throw std::runtime_error("unknown option " + a);
  }
  return o;
}

unsigned default_build_threads();  // hardware threads, or what the container's CPU quota pays for (defined with CpuQuota)
int cmd_idx(int argc, char **argv) {
  unsigned threads = default_build_threads();
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

// ---- where threads run and where their memory lives --------------------------------------------------------------
// The NUMA nodes of the box and the CPUs this process may use on each (its affinity mask at start-up), the first SMT
// sibling of every core listed apart: a group of threads that fits on a node's cores is kept off their second siblings.
// ABM_CLI_PIN=0 leaves every thread where the scheduler puts it (round 3's behaviour).
struct Topology {
  std::vector<std::vector<int>> primary, all;  // [node] -> CPUs
  std::vector<int> node_id;                    // [node] -> the system's id of that node (nodes without an allowed CPU are left out)
  bool pinning = true;
  static std::vector<int> parse_list(const std::string &s) {
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size() && std::isdigit(static_cast<unsigned char>(s[i]))) {
      const int a = std::atoi(s.c_str() + i);
      while (i < s.size() && std::isdigit(static_cast<unsigned char>(s[i]))) ++i;
      int b = a;
      if (i < s.size() && s[i] == '-') { ++i; b = std::atoi(s.c_str() + i); while (i < s.size() && std::isdigit(static_cast<unsigned char>(s[i]))) ++i; }
      for (int c = a; c <= b; ++c) out.push_back(c);
      if (i < s.size() && s[i] == ',') ++i;
    }
    return out;
  }
  static std::string first_line(const std::string &path) {
    std::ifstream f(path);
    std::string s;
    std::getline(f, s);
    return s;
  }
  Topology() {
    if (const char *e = std::getenv("ABM_CLI_PIN")) pinning = e[0] != '0';
    cpu_set_t mine;
    CPU_ZERO(&mine);
    const bool have_mask = sched_getaffinity(0, sizeof(mine), &mine) == 0;
    for (int n = 0; n < 64; ++n) {
      const std::vector<int> cpus = parse_list(first_line("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist"));
      if (cpus.empty()) { if (n == 0) continue; else break; }
      std::vector<int> p, a;
      for (int c : cpus) {
        if (have_mask && !CPU_ISSET(c, &mine)) continue;
        a.push_back(c);
        const std::vector<int> sib = parse_list(first_line("/sys/devices/system/cpu/cpu" + std::to_string(c) + "/topology/thread_siblings_list"));
        if (sib.empty() || sib.front() == c) p.push_back(c);
      }
      if (a.empty()) continue;
      if (p.empty()) p = a;
      primary.push_back(p);
      all.push_back(a);
      node_id.push_back(n);
    }
    if (all.empty()) {  // no sysfs: one node holding whatever the mask allows
      std::vector<int> a;
      for (int c = 0; c < CPU_SETSIZE; ++c) if (!have_mask || CPU_ISSET(c, &mine)) { if (have_mask || c < static_cast<int>(std::thread::hardware_concurrency())) a.push_back(c); }
      primary.push_back(a);
      all.push_back(a);
      node_id.push_back(0);
      pinning = false;
    }
  }
  int n_nodes() const { return static_cast<int>(all.size()); }
  // the index here of the system's node `id` (sysfs numbering), or -1 if this process may not run there
  int index_of(int id) const {
    for (size_t k = 0; k < node_id.size(); ++k) if (node_id[k] == id) return static_cast<int>(k);
    return -1;
  }
  size_t n_cores() const { size_t k = 0; for (const auto &p : primary) k += p.size(); return k; }
  // the calling thread onto `node`: onto its cores' first siblings while the `group` threads that share the node fit there
  void pin(int node, size_t group) const {
    if (!pinning) return;
    const std::vector<int> &cpus = group <= primary[node].size() ? primary[node] : all[node];
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)sched_setaffinity(0, sizeof(set), &set);
  }
};

// ---- BGZF input (bgzip-compressed FASTQ): blocks are independent gzip members that say how long they are ----------
// header: 1f 8b 08 04 | mtime(4) xfl os | xlen(2) | subfields ... 'B' 'C' 02 00 BSIZE(2) ... | deflate data | crc32 isize
// (SAM spec 4.1).  bsize_at returns the block's whole length (BSIZE + 1) or 0 if `p` does not start a BGZF block.
inline uint32_t le16(const unsigned char *p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8); }
inline uint32_t le32(const unsigned char *p) { return le16(p) | (le16(p + 2) << 16); }
uint32_t bgzf_block_length(const unsigned char *p, uint64_t avail, uint32_t &data_off) {
  if (avail < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
  const uint32_t xlen = le16(p + 10);
  if (12ull + xlen > avail) return 0;
  for (uint32_t at = 0; at + 4 <= xlen;) {
    const unsigned char *sf = p + 12 + at;
    const uint32_t slen = le16(sf + 2);
    if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && at + 6 <= xlen) {
      const uint32_t total = le16(sf + 4) + 1;
      data_off = 12 + xlen;
      return total >= data_off + 8 && total <= avail ? total : 0;
    }
    at += 4 + slen;
  }
  return 0;
}
bool looks_like_bgzf(const std::string &path) {
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  unsigned char head[512];
  const ssize_t got = ::pread(fd, head, sizeof(head), 0);
  struct stat sb;
  const bool regular = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
  ::close(fd);
  uint32_t data_off = 0;
  if (!regular || got < 18) return false;
  // (the first block's length may exceed what was read: only its header is checked here, the scan checks every block)
  if (head[0] != 0x1f || head[1] != 0x8b || head[2] != 8 || !(head[3] & 4)) return false;
  const uint32_t xlen = le16(head + 10);
  for (uint32_t at = 0; at + 6 <= xlen && 12 + at + 6 <= static_cast<uint32_t>(got);) {
    const unsigned char *sf = head + 12 + at;
    if (sf[0] == 'B' && sf[1] == 'C' && le16(sf + 2) == 2) return true;
    at += 4 + le16(sf + 2);
  }
  (void)data_off;
  return false;
}
// one thread's inflate state, reset per block
struct BgzfInflater {
  z_stream zs;
  bool live = false;
  ~BgzfInflater() { if (live) inflateEnd(&zs); }
  // block at `p` (whole length `len`, deflate data from `data_off`) -> dst (room for isize bytes); checks size and CRC
  void block(const unsigned char *p, uint32_t len, uint32_t data_off, char *dst, uint32_t isize) {
    if (!live) {
      std::memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("inflateInit2 failed");
      live = true;
    }
    else inflateReset(&zs);
    zs.next_in = const_cast<Bytef *>(p + data_off);
    zs.avail_in = len - data_off - 8;
    zs.next_out = reinterpret_cast<Bytef *>(dst);
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END || zs.total_out != isize) throw std::runtime_error("corrupt BGZF block in the reads file");
    if (static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), reinterpret_cast<const Bytef *>(dst), isize)) != le32(p + len - 8))
      throw std::runtime_error("BGZF block with a wrong checksum in the reads file");
  }
};

// a mapped input file that shrinks under the run (truncated, a network file system losing it) faults with SIGBUS
void install_sigbus_handler() {
  struct sigaction sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.sa_handler = [](int) {
    static const char msg[] = "abismal-amd: an input file changed or became unreadable while it was being read (SIGBUS on its mapping)\n";
    (void)!::write(2, msg, sizeof(msg) - 1);
    ::_exit(EXIT_FAILURE);
  };
  ::sigaction(SIGBUS, &sa, nullptr);
}

// The CPU time the container gives this process (CFS bandwidth control: cgroup v2 cpu.max, v1 cpu.cfs_quota_us): a pod
// of an 8-GPU node typically gets its share of the cores (16 of 128 on the box this was measured on) although it sees
// all 256 hardware threads.  More runnable threads than that do not run more: they burn the period's quota in its first
// milliseconds and the whole process is frozen for the rest of it (profiles/r04_trace_parts8_t64.log: every thread
// stalled 77 of every 100 ms) -- which is what made round 3's host pipeline "anti-scale" with its thread count.
struct CpuQuota {
  double cpus = 0;          // 0 = unlimited / unknown
  std::string stat_path;    // cpu.stat of the same cgroup
  bool v2 = false;
  static bool read_file(const std::string &path, std::string &out) {
    std::ifstream f(path);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
  }
  CpuQuota() {
    std::string own, v1_path, v2_path;
    if (read_file("/proc/self/cgroup", own)) {
      std::istringstream is(own);
      std::string line;
      while (std::getline(is, line)) {
        const size_t a = line.find(':'), b = line.find(':', a + 1);
        if (a == std::string::npos || b == std::string::npos) continue;
        const std::string ctl = line.substr(a + 1, b - a - 1), path = line.substr(b + 1);
        if (ctl.empty()) v2_path = path;
        else if (("," + ctl + ",").find(",cpu,") != std::string::npos) v1_path = path;
      }
    }
    std::string s;
    for (const std::string &dir : {std::string("/sys/fs/cgroup") + v2_path, std::string("/sys/fs/cgroup")})
      if (cpus == 0 && read_file(dir + "/cpu.max", s)) {
        long long q = 0, per = 0;
        if (std::sscanf(s.c_str(), "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) { cpus = static_cast<double>(q) / per; stat_path = dir + "/cpu.stat"; v2 = true; }
        else if (s.compare(0, 3, "max") == 0) { stat_path = dir + "/cpu.stat"; v2 = true; break; }
      }
    if (stat_path.empty())
      for (const std::string &dir : {std::string("/sys/fs/cgroup/cpu") + v1_path, std::string("/sys/fs/cgroup/cpu")}) {
        std::string qs, ps;
        if (read_file(dir + "/cpu.cfs_quota_us", qs) && read_file(dir + "/cpu.cfs_period_us", ps)) {
          const long long q = std::atoll(qs.c_str()), per = std::atoll(ps.c_str());
          if (q > 0 && per > 0) cpus = static_cast<double>(q) / per;
          stat_path = dir + "/cpu.stat";
          break;
        }
      }
  }
  // periods in which the cgroup was throttled so far, and for how long (seconds)
  void throttled(uint64_t &periods, double &seconds) const {
    periods = 0; seconds = 0;
    std::string s;
    if (stat_path.empty() || !read_file(stat_path, s)) return;
    std::istringstream is(s);
    std::string key;
    unsigned long long v = 0;
    while (is >> key >> v) {
      if (key == "nr_throttled") periods = v;
      else if (key == "throttled_usec") seconds = static_cast<double>(v) * 1e-6;
      else if (key == "throttled_time") seconds = static_cast<double>(v) * 1e-9;
    }
  }
};

unsigned default_build_threads() {
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const CpuQuota q;
  return q.cpus > 0 ? std::min(hw, std::max(1u, static_cast<unsigned>(q.cpus + 0.5))) : hw;
}

int cmd_map(int argc, char **argv) {
  const Options opt = parse_map(argc, argv);
  if (opt.out.empty()) { std::cerr << "Missing required argument\n-o, -outfile\n"; return EXIT_SUCCESS; }
  if (opt.reads.size() != 1 && opt.reads.size() != 2) { std::cerr << "usage: abismal-amd map -i idx -o out.sam [flags] reads_1.fq [reads_2.fq]\n"; return EXIT_SUCCESS; }
  if (opt.index.empty() == opt.genome.empty()) { std::cerr << "Select one of index file (-i) or genome file (-g)\n"; return EXIT_SUCCESS; }
  const bool paired = opt.reads.size() == 2;
  const int ends = paired ? 2 : 1;
  if (opt.host_ceiling && paired) throw std::runtime_error("-host-ceiling / -virtual-gpus take single-end input");
  const Topology topo;

  std::string index_path = opt.index;
  if (index_path.empty()) {  // -g: index the genome on the fly (src/abismal.cpp:2439-2446)
    index_path = opt.out + ".tmp.idx";
    if (abm_index_build(opt.genome.c_str(), index_path.c_str(), default_build_threads()) != 0) die_abm("indexing genome");
  }
  abm_index *ix = nullptr;
  const auto t_index = std::chrono::steady_clock::now();
  if (abm_index_open(index_path.c_str(), &ix) != 0) die_abm("loading index");
  if (opt.index.empty()) std::remove(index_path.c_str());
  g_min_read_len = 24 + abm_index_window(ix);
  Chroms ch;
  for (uint32_t i = 0; i < abm_index_n_chroms(ix); ++i) ch.names.push_back(abm_index_chrom_name(ix, i));
  ch.starts.assign(abm_index_chrom_starts(ix), abm_index_chrom_starts(ix) + ch.names.size() + 1);

  // GPUs.  "GPU g" of the run is device dev_of[g]: the visible devices in order, the first -gpus of them, or the list
  // -devices gives -- in which a device may appear more than once (replicas of the read sharding on one GPU: how a
  // one-GPU box runs the multi-GPU code path).  -host-ceiling / -virtual-gpus N: no device at all; every "GPU" hands
  // back made-up hits at once, so that what is measured is the host pipeline around N GPUs.
  // One replica of the index in each device's HBM, shared by that device's contexts; a context is one mapper thread's
  // workspaces + stream, and two per GPU keep the device busy while the other thread's batch is in transit over PCIe.
  std::vector<int> dev_of = opt.devices;
  int n_gpus = dev_of.empty() ? opt.gpus : static_cast<int>(dev_of.size());
  const bool virtual_gpus = opt.host_ceiling;
  // (paired-end: a batch ends in a tail of a few pairs with huge candidate sets that keep single waves busy for a second or
  // two after the rest is done -- DESIGN 4.3 -- so many smaller batches are kept in flight, each on a context of its own,
  // and their tails overlap: 16, as in bench.py's kernel loop.  8 M pairs 2x150 end to end on one GPU, kernels' own rate
  // 4.3 M reads/s: 3 contexts x 2 M pairs 2.55 M reads/s, 4 x 1 M 2.83, 8 x 1 M 3.3-3.4, 12 x 512 k 3.66, 16 x 1 M 3.72
  // (profiles/r04_pe_e2e_variants.log); a context's tier-2 workspaces are ~8 GB of HBM)
  // (GPUs of the run that are replicas on ONE device share that device's 16)
  int most_shared = 1;
  for (size_t g = 0; g < dev_of.size(); ++g)
    most_shared = std::max<int>(most_shared, static_cast<int>(std::count(dev_of.begin(), dev_of.end(), dev_of[g])));
  // (round 5, the pair kernels split by phase: on 8 M pairs 8 contexts carry what 16 do -- 3.80-3.91 M reads/s end to end
  // against 3.15-3.93, profiles/r05_pe_e2e_after_reserve.log -- but on 50 M pairs 16 give 5.57 M reads/s and 8 give 4.80,
  // profiles/r05_full_size.log: 16 it stays, as many as the device's free memory holds -- see below)
  // (round 5's final kernels, no seed-extension tables in paired runs: 24 slots of the bench loop 6.4-6.6 M reads/s against
  // 6.2 M with 16, profiles/r05_exp_pe_slots.log -- 24, as many as the device's free memory holds)
  int per_gpu = opt.mappers > 0 ? opt.mappers : (paired ? std::max(2, 24 / most_shared) : 2);
  if (!virtual_gpus) {
    if (opt.ext2 >= 0 && abm_index_set_seed_extension(ix, opt.ext2, opt.ext3) != 0) die_abm("seed extension");
    // (pairs: no tables unless asked for -- since the pair kernels narrow every range beyond max_candidates directly they are
    // as fast without them, 6.24 against 6.21 M reads/s, and 36 GB of device memory lighter: profiles/r05_exp_e2e_and_tables.log)
    if (opt.ext2 < 0 && paired && abm_index_set_seed_extension_cap(ix, 0, 0) != 0) die_abm("seed extension");
    if (opt.max_candidates && abm_index_set_max_candidates(ix, opt.max_candidates) != 0) die_abm("max candidates");
    {
      // window records for reads as long as the input's: the longest of each file's first 256 (a batch with a longer read
      // than the records serve is filtered from the bit planes, as without them)
      int want = opt.window_records;
      if (want < 0) {
        want = 0;
        for (const std::string &path : opt.reads)
          if (gzFile zf = gzopen(path.c_str(), "rb")) {
            std::vector<char> line(1 << 20);
            for (int rec = 0; rec < 256; ++rec) {
              if (!gzgets(zf, line.data(), static_cast<int>(line.size())) || !gzgets(zf, line.data(), static_cast<int>(line.size()))) break;
              want = std::max<int>(want, static_cast<int>(std::strcspn(line.data(), "\r\n")));
              if (!gzgets(zf, line.data(), static_cast<int>(line.size())) || !gzgets(zf, line.data(), static_cast<int>(line.size()))) break;
            }
            gzclose(zf);
          }
      }
      if (abm_index_set_window_records(ix, want) != 0) die_abm("window records");
    }
    const int visible = abm_device_count();
    if (visible <= 0) { std::cerr << "creating GPU context: no HIP device present (the mapping path has no CPU fallback)\n"; return EXIT_FAILURE; }
    if (n_gpus <= 0) n_gpus = visible;  // all that are visible
    if (dev_of.empty()) for (int g = 0; g < n_gpus; ++g) dev_of.push_back(g);
  }
  else {
    if (n_gpus <= 0) n_gpus = 1;
    dev_of.assign(n_gpus, -1);
    g_pin_batches = false;  // (page-locked memory comes from the HIP runtime)
  }
  bool shared_device = false;
  for (int g = 0; g < n_gpus; ++g) for (int h = 0; h < g; ++h) shared_device |= dev_of[g] >= 0 && dev_of[g] == dev_of[h];
  bool device_sam = false;  // decided below, once the number of host workers is known
  std::vector<abm_ctx *> ctxs;  // [g * per_gpu + k]
  if (!virtual_gpus) {
    // the first context on a device uploads the index and derives its tables there: every device's at the same time
    std::vector<abm_ctx *> first(n_gpus, nullptr);
    std::vector<std::thread> th;
    std::mutex emu;
    std::string err;
    for (int g = 0; g < n_gpus; ++g) {
      bool seen = false;
      for (int h = 0; h < g; ++h) seen |= dev_of[h] == dev_of[g];
      if (!seen) th.emplace_back([&, g] {
        if (abm_ctx_create(ix, dev_of[g], &first[g]) != 0) { std::lock_guard<std::mutex> lk(emu); err = abm_last_error(); }
      });
    }
    for (auto &t : th) t.join();
    if (!err.empty()) { std::cerr << "creating GPU context: " << err << "\n"; return EXIT_FAILURE; }
    if (paired) {
      // A paired-end context's workspaces are ~9 GB of device memory for batches of a million pairs (tier 2's per-wave lists
      // above all): with the index and its tables resident, sixteen of them fit the 288 GB part, not every device and not
      // every -mappers.  The default follows what is free; an explicit -mappers that cannot fit is an error here,
      // with the figures, rather than a failing allocation inside the first batch.
      const uint64_t pairs = opt.batch ? opt.batch : (1u << 20);
      int fit = 1 << 20;
      uint64_t need = 0, least_free = 0;
      for (int g = 0; g < n_gpus; ++g) {
        uint64_t free_b = 0, total_b = 0, per_ctx = 0;
        if (!first[g] || abm_device_memory(dev_of[g], &free_b, &total_b) != 0 || abm_ctx_pe_footprint(first[g], pairs, 150, &per_ctx) != 0 || per_ctx == 0) continue;
        const uint64_t share = static_cast<uint64_t>(std::count(dev_of.begin(), dev_of.end(), dev_of[g]));  // replicas of the run on this device
        const uint64_t usable = free_b - std::min<uint64_t>(free_b, uint64_t(2) << 30);
        const int f = static_cast<int>(std::min<uint64_t>(1 << 20, usable / per_ctx / share));
        if (f < fit) { fit = f; need = per_ctx; least_free = free_b; }
      }
      if (fit < per_gpu) {
        if (opt.mappers > 0 || fit < 1) {
          std::cerr << "creating GPU contexts: " << per_gpu << " paired-end mapper contexts per GPU need " << (need >> 20) << " MB of device memory each ("
                    << pairs << " pairs per batch), " << (least_free >> 20) << " MB are free beside the index: at most " << fit << " fit (-mappers, -batch)\n";
          return EXIT_FAILURE;
        }
        if (opt.verbose) std::cerr << "[abismal-amd] " << fit << " mapper contexts per GPU instead of " << per_gpu << ": " << (need >> 20) << " MB each, " << (least_free >> 20) << " MB free\n";
        per_gpu = fit;
      }
    }
    for (int g = 0; g < n_gpus; ++g)
      for (int k = 0; k < per_gpu; ++k) {
        abm_ctx *c = k == 0 ? first[g] : nullptr;
        if (!c && abm_ctx_create(ix, dev_of[g], &c) != 0) die_abm("creating GPU context");
        ctxs.push_back(c);
      }
  }
  // where each GPU's mapper threads and batch buffers live: the NUMA node its PCIe root hangs off (virtual GPUs: in
  // blocks, as the GPUs of a real node are wired)
  std::vector<int> gpu_node(n_gpus, 0);
  for (int g = 0; g < n_gpus; ++g) {
    // (abm_device_numa_node is the system's node id: under a cpuset that leaves a node out the nodes here are renumbered)
    int node = virtual_gpus ? g * topo.n_nodes() / n_gpus : topo.index_of(abm_device_numa_node(dev_of[g]));
    if (node < 0 || node >= topo.n_nodes()) node = g % topo.n_nodes();
    gpu_node[g] = node;
  }
  auto env_reads = [](const char *name, size_t dflt) { const char *e = std::getenv(name); return e && std::atoll(e) > 0 ? static_cast<size_t>(std::atoll(e)) : dflt; };
  if (!virtual_gpus) {
    // set-up, like the index upload: workspaces for full batches of reads as long as the input's first one, and the
    // kernels' code loaded, before the clock of the run starts (a longer read later only makes the buffers grow)
    uint32_t first_len = 100;
    if (gzFile zf = gzopen(opt.reads[0].c_str(), "rb")) {
      char line[65536];
      if (gzgets(zf, line, sizeof(line)) && gzgets(zf, line, sizeof(line))) first_len = static_cast<uint32_t>(std::strcspn(line, "\r\n"));
      gzclose(zf);
    }
    // (the same expression the mappers use for a full batch, rounded up to whole slices as they do)
    const size_t slice_for_reserve = env_reads("ABM_CLI_SLICE_READS", 1u << 15);
    size_t reserve_reads = opt.batch ? opt.batch : env_reads("ABM_CLI_BATCH_READS", paired ? (1u << 20) : (1u << 23));
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
    // single-end SAM text: the kernel writes every read's line after QNAME itself (abm_ctx_set_sam_tails), the formatters
    // put names in front -- half of the host's CPU time per read was building that text base by base.  BAM records are
    // built from the fields as before; ABM_CLI_HOST_FORMAT=1: SAM text too (same-box comparisons, tests)
    // Where it pays: with few host workers per GPU.  40 M reads on one GPU, end to end (profiles/r05_exp_device_sam_text.log):
    // -t 2 12.2-12.4 M reads/s against 8.4 with host formatting, -t 4 13.0-13.5 against 12.9-13.0, -t 16 13.8-13.9 against
    // 14.2-14.3 (the kernel writes 160 bytes more per read across PCIe); process CPU 8.5 s against 12.2 s.  So the device
    // writes the text when a GPU has fewer than twelve workers to itself (ABM_CLI_DEVICE_SAM=0 / 1 decides otherwise).
    {
      const CpuQuota q0;
      unsigned workers = opt.threads ? std::max(1u, opt.threads) : static_cast<unsigned>(std::min<size_t>(std::max<size_t>(topo.n_cores(), 1), 8u + 8u * static_cast<unsigned>(n_gpus)));
      if (q0.cpus > 0 && !std::getenv("ABM_CLI_NO_QUOTA_CLAMP")) workers = std::min(workers, std::max(1u, static_cast<unsigned>(q0.cpus + 0.5)));
      device_sam = !paired && !opt.bam && !std::getenv("ABM_CLI_NO_STREAM") && !std::getenv("ABM_CLI_HOST_FORMAT") && workers < 12u * static_cast<unsigned>(n_gpus);
      if (const char *e = std::getenv("ABM_CLI_DEVICE_SAM")) device_sam = !paired && !opt.bam && !std::getenv("ABM_CLI_NO_STREAM") && e[0] != '0';
    }
    if (device_sam)
      for (abm_ctx *c : ctxs) if (abm_ctx_set_sam_tails(c, 1, opt.ambig ? 1 : 0) != 0) die_abm("SAM text on the device");
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

  mallopt(M_MMAP_THRESHOLD, 32 << 20);  // (the largest value the library takes: blocks below it come from its arenas ...)
  mallopt(M_TRIM_THRESHOLD, 1 << 30);   // (... and stay there)

  abm_params par;
  abm_default_params(&par);
  par.max_candidates = opt.max_candidates;
  par.valid_frac = opt.max_distance;
  par.min_frag = opt.min_frag;
  par.max_frag = opt.max_frag;
  par.allow_ambig = opt.ambig;
  const int se_mode = opt.rpbat ? ABM_SE_RANDOM : ((opt.arich || opt.pbat) ? ABM_SE_A_RICH : ABM_SE_T_RICH);
  const int pe_mode = opt.rpbat ? ABM_PE_RANDOM : (opt.pbat ? ABM_PE_PBAT : ABM_PE_NORMAL);

  // Pipeline.  The unit of host work is a SLICE (32 k records, in file order); every stage runs on many slices at
  // once and only the assignment of output offsets looks at their order:
  //   count    plain files: host workers count newlines chunk by chunk (pread) -- record j starts at line 4j, so there
  //            is no guessing at record boundaries;
  //   cut      one thread per REGION turns the counts into slice byte ranges (gzip input: one thread inflates and cuts)
  //   parse    host workers pread the slice's text, apply ReadLoader's rules, lay the reads out for the C ABI
  //   map      (-mappers per GPU) a mapper takes a run of consecutive parsed slices of its region, up to its batch size
  //   format   host workers: SAM text / BAM blocks and statistics per slice
  //   write    one writer per region: slices in order, one pwrite each
  // A REGION is a contiguous share of the input with an output file of its own (-out-parts R: <out>.part000 ...; `cat`
  // of the parts in order is byte for byte the file a run without parts writes).  One region (the default) is the
  // reference's contract: one output file; its cap is what ONE file takes -- a tmpfs file 6.5 GB/s from any number of
  // writers (they queue for its lock, profiles/r04_sink_probe.log) = 40 M reads/s of SAM text -- while files of their
  // own scale with their number (78 GB/s from 16).  Regions share nothing but the host workers: each has its GPUs, its
  // cutter, its queue of parsed slices, its lead-in, its writer, all on the NUMA node of its GPUs.
  // Host workers are ONE pool (round 3 ran -t parsers plus -t formatters plus 16 counters: 275 threads on 256 at -t 128):
  // -t threads, spread over the NUMA nodes and pinned there; a worker takes the most urgent task of its own node
  // (format, then parse, then count) and another node's only when its own has none.  A slice's buffers are first
  // touched, parsed and formatted on one node.
  std::mutex mu;
  auto env_or = [](const char *name, uint64_t dflt) { const char *e = std::getenv(name); return e && std::atoll(e) > 0 ? static_cast<uint64_t>(std::atoll(e)) : dflt; };
  // (test hooks: ABM_CLI_SLICE_READS / ABM_CLI_CHUNK_BYTES / ABM_CLI_MARK_LINES shrink the units so that small
  // fixtures cross many slice, chunk and mark boundaries)
  const size_t slice_reads = static_cast<size_t>(env_or("ABM_CLI_SLICE_READS", 1u << 15));
  // single-end results leave the library slice by slice while the kernel runs (ABM_CLI_NO_STREAM=1: whole batches, as
  // the paired-end path takes them); virtual GPUs hand their made-up hits over the same way
  const bool stream_slices = !paired && !std::getenv("ABM_CLI_NO_STREAM");
  // size of a GPU's first batch (see the mapper's target()); ABM_CLI_FIRST_BATCH=n overrides, a huge n = no special first batch
  // paired-end batches near the end of the input at most 1 / ABM_CLI_PE_TAPER of what is left (0 = off, the default: see target())
  const size_t pe_taper = [] { const char *e = std::getenv("ABM_CLI_PE_TAPER"); return e ? static_cast<size_t>(std::max<long long>(0, std::atoll(e))) : size_t(0); }();
  const size_t first_batch_reads = static_cast<size_t>(env_or("ABM_CLI_FIRST_BATCH", stream_slices ? 1u << 19 : (paired ? 1u << 17 : 1u << 21)));
  // BGZF-compressed input (bgzip): its blocks are independent, so the workers inflate them side by side into one
  // anonymous mapping that then IS the input as far as counting, cutting and parsing go (a single-member .gz, what plain
  // gzip writes, has no such structure: one thread inflates it, as in the reference, src/abismal.cpp:150-209)
  const bool bgzf_input = !std::getenv("ABM_CLI_NO_BGZF") && std::all_of(opt.reads.begin(), opt.reads.end(), looks_like_bgzf);
  const bool plain_input = bgzf_input || [&] {
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
  const int n_regions = std::max(1, opt.out_parts);
  if (n_regions > 1 && (!plain_input || bgzf_input)) throw std::runtime_error("-out-parts needs plain (seekable, uncompressed) FASTQ input");
  if (n_regions > n_gpus * per_gpu) throw std::runtime_error("-out-parts: more parts than mapper threads");
  const int n_nodes = topo.n_nodes();
  // Batches are FULL (-batch reads) except at the end of the input: the mapping kernel's time has a floor set
  // by its costliest reads (a quarter of a second, whatever the batch), so small batches waste the GPU;
  // cutting and parsing run far ahead of it, so a full batch is ready within a fraction of a kernel's time.
  const size_t batch_reads = opt.batch ? opt.batch : static_cast<size_t>(env_or("ABM_CLI_BATCH_READS", paired ? (1u << 20) : (1u << 23)));
  // host workers: -t, else 8 plus 8 per GPU (what 14 M reads/s per GPU of counting, parsing and formatting take, twice
  // over: 0.2 us of CPU per read), never more than the box has cores (second SMT siblings add little, and the mapper
  // threads, the writers and the HIP runtime's own threads need somewhere to run) -- and never more than the CPU time
  // the container's quota gives the process (see CpuQuota), whatever -t says: threads beyond it do not run more, they
  // get the whole process frozen for most of every scheduling period (ABM_CLI_NO_QUOTA_CLAMP=1 to measure just that).
  const CpuQuota quota;
  unsigned n_host = opt.threads ? std::max(1u, opt.threads)
                                : static_cast<unsigned>(std::min<size_t>(std::max<size_t>(topo.n_cores(), 1), 8u + 8u * static_cast<unsigned>(n_gpus)));
  n_host = std::min<unsigned>(n_host, static_cast<unsigned>(std::max<size_t>(1, topo.n_cores() * 2)));
  if (quota.cpus > 0 && !std::getenv("ABM_CLI_NO_QUOTA_CLAMP")) {
    const unsigned cap = std::max(1u, static_cast<unsigned>(quota.cpus + 0.5));
    if (n_host > cap) {
      if (opt.verbose || opt.threads) std::cerr << "[abismal-amd] " << n_host << " host workers asked for, " << cap << " started: the container's CPU quota is " << quota.cpus << " CPUs\n";
      n_host = cap;
    }
  }
  uint64_t throttled0 = 0; double throttled_s0 = 0;
  quota.throttled(throttled0, throttled_s0);
  const size_t max_reads_in_flight = (static_cast<size_t>(n_gpus) * per_gpu + 2) * batch_reads + 4 * slice_reads * n_host;

  // which region a mapper thread serves: regions are dealt to the GPUs in blocks, a GPU's mappers take its regions in turn
  auto region_of = [&](int g, int k) -> int {
    if (n_regions >= n_gpus) {
      const int lo = g * n_regions / n_gpus, hi = (g + 1) * n_regions / n_gpus;
      return lo + k % std::max(1, hi - lo);
    }
    return g * n_regions / n_gpus;
  };

  struct Region {
    int id = 0;
    std::vector<int> nodes;                                   // NUMA nodes of its GPUs: its slices are striped over them
    int writer_node = 0;
    uint64_t n_slices = 0, next_to_map = 0, run_end = 0, n_parsed = 0;  // parsed[next_to_map .. run_end) are all there
    size_t run_reads = 0;                                     // reads in that run
    std::map<uint64_t, std::unique_ptr<Slice>> parsed;        // parsed, waiting for a mapper (by slice number)
    bool cut_done = false;
    int mappers_live = 0;
    std::vector<std::string> carry[2];                        // lead-in of the region's next batch (see the mapper)
    size_t reads_in_flight = 0, max_in_flight = 0;
    // output
    std::string path;
    int fd = -1;
    bool seekable = true;
    uint64_t file_offset = 0;                                 // bytes of output whose place is fixed
    std::map<uint64_t, Slice *> formatted;                    // formatted, place not yet known
    std::deque<Slice *> q_write;                              // formatted and placed, waiting to be written
    uint64_t next_to_place = 0, slices_written = 0;
    std::condition_variable cv_flow,                          // its cutter: room for more reads in flight
                            cv_map,                           // its mappers: a slice has been parsed
                            cv_write;                         // its writer: a slice's place in the file is fixed
    uint64_t records = 0;
  };
  std::vector<Region> regions(n_regions);
  for (int r = 0; r < n_regions; ++r) {
    Region &R = regions[r];
    R.id = r;
    for (int g = 0; g < n_gpus; ++g)
      for (int k = 0; k < per_gpu; ++k)
        if (region_of(g, k) == r) {
          ++R.mappers_live;
          if (std::find(R.nodes.begin(), R.nodes.end(), gpu_node[g]) == R.nodes.end()) R.nodes.push_back(gpu_node[g]);
        }
    if (n_regions == 1) { R.nodes.clear(); for (int n = 0; n < n_nodes; ++n) R.nodes.push_back(n); }  // one region: every node works on it
    R.writer_node = R.nodes[0];
    R.max_in_flight = std::max<size_t>(max_reads_in_flight / n_regions, 2 * batch_reads);
  }
  const uint64_t kStripe = env_or("ABM_CLI_STRIPE_SLICES", 4);  // consecutive slices of a region that share a node
  auto node_of_slice = [&](const Region &R, uint64_t g) { return R.nodes[(g / kStripe) % R.nodes.size()]; };

  // ---- output files.  A slice's place is fixed in slice order; its region's writer pwrite()s it.
  // (Writes to one file take its inode lock, so they run one at a time; copying into a shared mapping of the file
  // from all threads instead was measured 3x SLOWER -- page faults on the mapping contend far worse than the lock.)
  struct FdCloser { std::vector<Region> *rs; ~FdCloser() { for (Region &R : *rs) if (R.fd >= 0) ::close(R.fd); } } out_closer{&regions};
  for (Region &R : regions) {
    char suffix[32];
    std::snprintf(suffix, sizeof(suffix), ".part%03d", R.id);
    R.path = (n_regions == 1 || opt.out == "/dev/null") ? opt.out : opt.out + suffix;
    R.fd = ::open(R.path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (R.fd < 0) throw std::runtime_error("failed to open output file: " + R.path);
    R.seekable = ::lseek(R.fd, 0, SEEK_CUR) >= 0;
  }
  auto write_all = [&](Region &R, const char *p, size_t len, uint64_t at) {
    while (len) {
      const ssize_t w = R.seekable ? ::pwrite(R.fd, p, len, static_cast<off_t>(at)) : ::write(R.fd, p, len);
      if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error("failed writing output file: " + R.path); }
      p += w; at += static_cast<uint64_t>(w); len -= static_cast<size_t>(w);
    }
  };
  {  // header, src/abismal.cpp:2265-2293 (the first part's)
    std::ostringstream h;
    h << "@HD\tVN:1.0\n";
    for (size_t i = 1; i + 1 < ch.names.size(); ++i) h << "@SQ\tSN:" << ch.names[i] << "\tLN:" << (ch.starts[i + 1] - ch.starts[i]) << '\n';
    h << "@PG\tID:ABISMAL\tVN:" << kVersion << "\tCL:\"";
    for (int i = 0; i < argc; ++i) h << argv[i] << ' ';
    h << "\"\n";
    std::string z;
    if (!opt.bam) z = h.str();
    else bgzf_compress(bam_header_bytes(h.str(), ch), z);
    write_all(regions[0], z.data(), z.size(), 0);
    regions[0].file_offset = z.size();
  }

  // one condition variable per kind of waiter (and per node / region): an event wakes the threads it concerns
  std::condition_variable cv_chunk;   // cutters of plain files: a chunk's newline counts are there
  struct NodeQueues {
    std::deque<std::unique_ptr<Slice>> parse;   // cut, waiting for a worker
    std::deque<Slice *> format;                 // mapped, waiting for a worker
    std::condition_variable cv;                 // this node's idle workers
    int idle = 0;
    int waking = 0;                             // wake-ups sent to idle workers that have not taken mu yet
  };
  std::vector<NodeQueues> nq(n_nodes);
  auto wake_everyone = [&] {
    cv_chunk.notify_all();
    for (NodeQueues &q : nq) q.cv.notify_all();
    for (Region &R : regions) { R.cv_flow.notify_all(); R.cv_map.notify_all(); R.cv_write.notify_all(); }
  };
  // a task for `node` has been queued (mu held): one idle worker there, or failing that anywhere, wakes up
  // (a woken worker only leaves `idle` once it holds mu again: `waking` counts the wake-ups already sent, so that two tasks
  // queued under one hold of mu wake two workers -- the second one of another node if this node has one idler)
  auto wake_worker = [&](int node) {
    if (nq[node].idle > nq[node].waking) { ++nq[node].waking; nq[node].cv.notify_one(); return; }
    for (int n = 0; n < n_nodes; ++n) if (nq[n].idle > nq[n].waking) { ++nq[n].waking; nq[n].cv.notify_one(); return; }
  };
  std::vector<std::unique_ptr<Batch>> live_batches;
  uint64_t n_batches = 0;
  size_t max_lead = 0;  // the longest lead-in a batch carried (records)
  uint64_t region_lead_scanned = 0, region_lead_records = 0;  // -out-parts: records before a region scanned for its lead-in / taken into it (largest over the regions)
  int mappers_live = n_gpus * per_gpu;
  std::vector<SlicePool> slice_pool(n_nodes);
  std::vector<BatchPool> batch_pool(n_nodes);
  std::exception_ptr failure;
  std::vector<Stats3> gpu_stats(n_gpus);
  std::vector<uint64_t> gpu_batches(n_gpus, 0), gpu_reads(n_gpus, 0);  // what each GPU was handed
  std::vector<unsigned> workers_on(n_nodes, 0);
  for (unsigned t = 0; t < n_host; ++t) ++workers_on[t % n_nodes];

  // Set-up, like the index upload and abm_ctx_reserve: the slices and batches the run will have in flight, with their
  // buffers sized from the input's first records and their pages touched -- by threads of the node that will use them.
  // A run's first second otherwise touches gigabytes of fresh memory from a hundred threads that share one address
  // space: page faults and the allocator's calls for more memory, which stall one another.
  double host_prepare_s = 0;
  if (plain_input && !bgzf_input && !std::getenv("ABM_CLI_NO_PREWARM")) {
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
      const size_t want_slices = static_cast<size_t>(std::min<uint64_t>((in_flight + slice_reads - 1) / slice_reads + n_host / 4, 4096));
      const size_t batch_cap = static_cast<size_t>(std::min<uint64_t>(batch_reads, n_recs)) + 512;
      // batches per GPU: its mappers' plus one, but no more than its share of the input makes
      const size_t per_gpu_batches = static_cast<size_t>(std::min<uint64_t>(static_cast<uint64_t>(per_gpu) + 1, (n_recs / n_gpus + batch_cap - 1) / batch_cap + 1));
      // tasks per node: slices for the regions whose slices live there (in proportion), batch pieces for the GPUs there
      struct Task { int kind; size_t a, b; };  // 0: a slice; 1: piece b of batch a
      std::vector<std::vector<Task>> tasks(n_nodes);
      std::vector<std::vector<std::unique_ptr<Batch>>> bt(n_nodes);
      {
        std::vector<size_t> share(n_nodes, 0);
        size_t total = 0;
        for (const Region &R : regions) for (int n : R.nodes) { ++share[n]; ++total; }
        for (int n = 0; n < n_nodes; ++n) {
          const size_t k = total ? (want_slices * share[n] + total - 1) / total : 0;
          for (size_t i = 0; i < k; ++i) tasks[n].push_back(Task{0, i, 0});
        }
        for (int g = 0; g < n_gpus; ++g)
          for (size_t i = 0; i < per_gpu_batches; ++i) {
            const int n = gpu_node[g];
            bt[n].emplace_back(new Batch);
            for (size_t piece = 0; piece < 8; ++piece) tasks[n].push_back(Task{1, bt[n].size() - 1, piece});
          }
      }
      std::vector<std::atomic<size_t>> next(n_nodes);
      for (auto &x : next) x = 0;
      std::vector<std::vector<std::unique_ptr<Slice>>> made(n_nodes);
      for (int n = 0; n < n_nodes; ++n) made[n].resize(tasks[n].size());
      auto touch = [](char *q, size_t bytes) { for (size_t i = 0; i < bytes; i += 4096) q[i] = 0; };
      std::vector<std::thread> th;
      for (unsigned t = 0; t < n_host; ++t)
        th.emplace_back([&, t] {
          const int node = static_cast<int>(t % n_nodes);
          topo.pin(node, workers_on[node]);
          for (;;) {
            const size_t k = next[node].fetch_add(1);
            if (k >= tasks[node].size()) break;
            const Task &tk = tasks[node][k];
            if (tk.kind == 0) {
              std::unique_ptr<Slice> x(new Slice);
              x->node = node;
              for (int e = 0; e < ends; ++e) {
                if (std::getenv("ABM_CLI_NO_MMAP")) { x->raw[e].reserve(slice_reads * rec_bytes + (1u << 16)); touch(x->raw[e].p, x->raw[e].cap); }  // (mapped input is parsed in place)
                x->blob[e].reserve(slice_reads * (read_len + 2)); touch(x->blob[e].p, x->blob[e].cap);
                x->names[e].reserve(slice_reads + 16);
                x->off[e].reserve(slice_reads + 16);
                touch(reinterpret_cast<char *>(x->names[e].data()), (slice_reads + 16) * sizeof(NameRef));
                touch(reinterpret_cast<char *>(x->off[e].data()), (slice_reads + 16) * sizeof(uint64_t));
              }
              x->text.reserve(slice_reads * ends * 330); touch(x->text.p, x->text.cap);
              if (stream_slices) {
                x->own_se.b.reserve(slice_reads * sizeof(abm_hit)); touch(x->own_se.b.p, x->own_se.b.cap);
                x->own_cig.b.reserve((4 * slice_reads + 64) * 4); touch(x->own_cig.b.p, x->own_cig.b.cap);
                x->own_cig_off.b.reserve((slice_reads + 1) * 8); touch(x->own_cig_off.b.p, x->own_cig_off.b.cap);
              }
              made[node][k] = std::move(x);
            }
            else {  // a batch's arrays, one piece per task
              Batch &b = *bt[node][tk.a];
              const size_t n = batch_cap, piece = tk.b;
              const int e = static_cast<int>(piece & 1);
              if (e >= ends) continue;
              switch (piece >> 1) {
                case 0: b.blob[e].reserve(n * (read_len + 2)); touch(b.blob[e].p, b.blob[e].cap); break;
                case 1: b.off_bytes[e].reserve((n + 1) * 8); touch(b.off_bytes[e].p, b.off_bytes[e].cap);
                        if (!stream_slices) { b.se[e].b.reserve(n * sizeof(abm_hit)); touch(b.se[e].b.p, b.se[e].b.cap); }
                        break;
                case 2: if (!stream_slices) { b.cig[e].b.reserve((4 * n + 1024) * 4); touch(b.cig[e].b.p, b.cig[e].b.cap); } break;
                default: if (!stream_slices) { b.cig_off[e].b.reserve((n + 1) * 8); touch(b.cig_off[e].b.p, b.cig_off[e].b.cap); }
                         if (paired && e == 0) { b.pairs.b.reserve(n * sizeof(abm_pair)); touch(b.pairs.b.p, b.pairs.b.cap); }
              }
            }
          }
        });
      for (auto &t : th) t.join();
      for (int n = 0; n < n_nodes; ++n) {
        for (auto &x : made[n]) if (x) slice_pool[n].put(std::move(x));
        for (auto &x : bt[n]) if (x) batch_pool[n].put(std::move(x));
      }
    }
    host_prepare_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - tp).count();
  }
  struct rusage ru0;
  ::getrusage(RUSAGE_SELF, &ru0);
  const auto t_start = std::chrono::steady_clock::now();
  double count_done_s = 0;                           // when the last chunk's newlines were counted (plain input)
  std::vector<double> region_first_batch_s(n_regions, -1.0);  // when each region's first batch was handed to a mapper

  // seconds of work, summed over threads (the wait for the pipeline's lock is not in them: round 3's figures included it)
  double busy_split = 0, busy_parse = 0, busy_map = 0, busy_format = 0, busy_write = 0, lock_wait = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  std::vector<RawPool> raw_pool(n_nodes);
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
  // a fresh (recycled) slice for slice number g of region R, from the pool of the node it will live on
  auto new_slice = [&](const Region &R, uint64_t g) {
    const int node = node_of_slice(R, g);
    std::unique_ptr<Slice> sl = slice_pool[node].get();
    sl->node = node;
    sl->region = R.id;
    return sl;
  };
  // hands a cut slice to the workers (blocks while too much of its region is in flight)
  auto emit_slice = [&](Region &R, std::unique_ptr<Slice> sl, size_t records) -> bool {
    std::unique_lock<std::mutex> lk(mu);
    R.cv_flow.wait(lk, [&] { return failure || R.reads_in_flight < R.max_in_flight; });
    if (failure) return false;
    sl->g = R.n_slices++;
    trace("cut", sl->g, records);
    R.reads_in_flight += records;
    const int node = sl->node;
    nq[node].parse.push_back(std::move(sl));
    wake_worker(node);
    return true;
  };

  // ---- cut, gzip (or non-regular) input: one inflating reader, slices carry their text
  auto cutter_stream = [&]() {
    Region &R = regions[0];
    try {
      RawSplitter s1(opt.reads[0]);
      std::unique_ptr<RawSplitter> s2;
      if (paired) s2.reset(new RawSplitter(opt.reads[1]));
      for (uint64_t g = 0;; ++g) {
        std::unique_ptr<Slice> sl = new_slice(R, g);
        const auto t0 = now();
        if (!sl->raw[0].p) sl->raw[0] = raw_pool[sl->node].get();
        if (paired && !sl->raw[1].p) sl->raw[1] = raw_pool[sl->node].get();
        const uint64_t l1 = s1.next(slice_reads, sl->raw[0], sl->first_line[0]);
        uint64_t l2 = 0;
        if (paired) l2 = s2->next(slice_reads, sl->raw[1], sl->first_line[1]);
        const bool last = s1.exhausted() || (paired && s2->exhausted());
        if (l1 == 0 && (!paired || l2 == 0)) break;
        const double dt = since(t0);
        { std::lock_guard<std::mutex> lk(mu); busy_split += dt; }
        if (!emit_slice(R, std::move(sl), (l1 + 3) / 4)) break;
        if (last) break;
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    R.cut_done = true;
    wake_everyone();
  };

  // ---- cut, plain files: newline counts per chunk (host workers), then slice byte ranges (serial per region, cheap)
  struct ChunkInfo { uint64_t lines = 0; std::vector<uint32_t> marks; bool ready = false; };  // marks: offset just past every kMark-th newline
  const uint64_t kChunk = env_or("ABM_CLI_CHUNK_BYTES", 8u << 20), kMark = env_or("ABM_CLI_MARK_LINES", 1024);
  struct LineFile {
    int fd = -1;
    uint64_t size = 0, n_chunks = 0;
    std::vector<ChunkInfo> chunks;
    uint64_t next_chunk = 0;  // next chunk a worker counts
    uint64_t n_ready = 0;
    bool ends_with_newline = true;
    // The file mapped read-only: counting and parsing read the page cache in place (names stay views into the mapping)
    // instead of copying the whole input out of it twice with pread -- a fifth of the pipeline's CPU time per read
    // (profiles/r04_host_ceiling.log).  nullptr (mapping refused, ABM_CLI_NO_MMAP=1): pread into per-slice buffers.
    const char *map = nullptr;
    std::vector<uint64_t> chunk_begin;  // [n_chunks + 1] where each chunk begins in the text (plain files: multiples of kChunk)
    // BGZF: the compressed file mapped, its blocks (offset and length in the file, offset in the text), the blocks of
    // each chunk, and how far the text has been handed back to the system (everything before the oldest unwritten slice)
    const unsigned char *cmap = nullptr;
    uint64_t csize = 0;
    std::vector<uint64_t> block_at, block_text;
    std::vector<uint32_t> chunk_first_block;
    uint64_t released_upto = 0;
  };
  // BGZF input: how much inflated text may exist beyond the oldest unwritten slice (what the pipeline may hold in flight,
  // generously: a record of 100-base reads is ~250 bytes, twice that for long names and reads)
  const uint64_t inflate_ahead_bytes = env_or("ABM_CLI_INFLATE_AHEAD", std::max<uint64_t>(4ull << 30, static_cast<uint64_t>(max_reads_in_flight) * 512));
  bool count_gated = false;  // BGZF: inflating further ahead is on hold until slices have been written
  std::vector<LineFile> lf(plain_input ? opt.reads.size() : 0);
  for (size_t e = 0; e < lf.size(); ++e) {
    lf[e].fd = ::open(opt.reads[e].c_str(), O_RDONLY);
    if (lf[e].fd < 0) throw std::runtime_error("cannot open reads file: " + opt.reads[e]);
    struct stat sb;
    if (::fstat(lf[e].fd, &sb) != 0) throw std::runtime_error("cannot stat reads file: " + opt.reads[e]);
    lf[e].size = static_cast<uint64_t>(sb.st_size);
    if (bgzf_input) {
      // walk the block headers (each says how long its block is; its last four bytes how long its text): the text's
      // extent and every block's place in it are known before a single block is inflated
      LineFile &F = lf[e];
      F.csize = F.size;
      void *cm = F.csize ? ::mmap(nullptr, F.csize, PROT_READ, MAP_SHARED, F.fd, 0) : nullptr;
      if (F.csize && cm == MAP_FAILED) throw std::runtime_error("cannot map reads file: " + opt.reads[e]);
      F.cmap = static_cast<const unsigned char *>(cm);
      uint64_t at = 0, text = 0;
      while (at < F.csize) {
        uint32_t data_off = 0;
        const uint32_t len = bgzf_block_length(F.cmap + at, F.csize - at, data_off);
        if (!len) throw std::runtime_error("reads file is not BGZF all the way through (block at byte " + std::to_string(at) + "): " + opt.reads[e]);
        const uint32_t isize = le32(F.cmap + at + len - 4);
        if (isize) { F.block_at.push_back(at); F.block_text.push_back(text); }
        text += isize;
        at += len;
      }
      F.block_at.push_back(at);
      F.block_text.push_back(text);
      F.size = text;
      // chunks = runs of blocks holding up to kChunk of text
      for (size_t b = 0; b + 1 < F.block_at.size();) {
        F.chunk_begin.push_back(F.block_text[b]);
        F.chunk_first_block.push_back(static_cast<uint32_t>(b));
        size_t b1 = b + 1;
        while (b1 + 1 < F.block_at.size() && F.block_text[b1 + 1] - F.block_text[b] <= kChunk) ++b1;
        b = b1;
      }
      F.chunk_first_block.push_back(static_cast<uint32_t>(F.block_at.size() - 1));
      F.chunk_begin.push_back(F.size);
      F.n_chunks = F.chunk_begin.size() - 1;
      F.chunks.resize(F.n_chunks);
      if (F.size) {
        void *m = ::mmap(nullptr, F.size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) throw std::runtime_error("cannot reserve address space for the inflated reads of " + opt.reads[e]);
        F.map = static_cast<const char *>(m);
        // the text's last byte (does the file end with a newline?): its last block, inflated here
        const size_t lb = F.block_at.size() - 2;
        uint32_t data_off = 0;
        const uint32_t len = bgzf_block_length(F.cmap + F.block_at[lb], F.csize - F.block_at[lb], data_off);
        std::vector<char> tail(F.block_text[lb + 1] - F.block_text[lb]);
        BgzfInflater inf;
        inf.block(F.cmap + F.block_at[lb], len, data_off, tail.data(), static_cast<uint32_t>(tail.size()));
        F.ends_with_newline = tail.back() == '\n';
      }
      continue;
    }
    lf[e].n_chunks = (lf[e].size + kChunk - 1) / kChunk;
    lf[e].chunks.resize(lf[e].n_chunks);
    for (uint64_t k = 0; k <= lf[e].n_chunks; ++k) lf[e].chunk_begin.push_back(std::min(lf[e].size, k * kChunk));
    if (lf[e].size) { char c = 0; if (::pread(lf[e].fd, &c, 1, static_cast<off_t>(lf[e].size - 1)) == 1) lf[e].ends_with_newline = c == '\n'; }
    if (lf[e].size && !std::getenv("ABM_CLI_NO_MMAP")) {
      void *m = ::mmap(nullptr, lf[e].size, PROT_READ, MAP_SHARED, lf[e].fd, 0);
      if (m != MAP_FAILED) lf[e].map = static_cast<const char *>(m);
    }
  }
  struct Unmapper { std::vector<LineFile> *v; ~Unmapper() { for (LineFile &F : *v) { if (F.map) ::munmap(const_cast<char *>(F.map), F.size); if (F.cmap) ::munmap(const_cast<unsigned char *>(F.cmap), F.csize); } } } unmapper{&lf};
  const bool mapped_input = !lf.empty() && std::all_of(lf.begin(), lf.end(), [](const LineFile &F) { return F.map != nullptr || F.size == 0; });
  if (mapped_input) install_sigbus_handler();
  auto read_range = [&](int fd, const std::string &path, char *dst, uint64_t lo, uint64_t hi) {
    while (lo < hi) {
      const ssize_t got = ::pread(fd, dst, hi - lo, static_cast<off_t>(lo));
      if (got < 0) { if (errno == EINTR) continue; throw std::runtime_error("error reading " + path); }
      if (got == 0) throw std::runtime_error("unexpected end of " + path);
      dst += got; lo += static_cast<uint64_t>(got);
    }
  };
  // (mu held) the next chunk to count: of the file whose counting is least advanced (both files of a pair are cut in step)
  auto take_chunk = [&](size_t &e, uint64_t &k) -> bool {
    size_t best = lf.size();
    for (size_t f = 0; f < lf.size(); ++f)
      if (lf[f].next_chunk < lf[f].n_chunks && (best == lf.size() || lf[f].next_chunk < lf[best].next_chunk)) best = f;
    if (best == lf.size()) return false;
    // (BGZF: the text is inflated into memory and handed back as slices are written; no further ahead of that than what
    // may be in flight anyway)
    if (lf[best].cmap && lf[best].chunk_begin[lf[best].next_chunk] > lf[best].released_upto + inflate_ahead_bytes) {
      // (the mappers must not wait for more reads than this lets through: they take what is parsed)
      if (!count_gated) { count_gated = true; for (Region &R : regions) R.cv_map.notify_all(); }
      return false;
    }
    count_gated = false;
    e = best; k = lf[e].next_chunk++;
    return true;
  };
  auto count_chunk = [&](size_t e, uint64_t k, std::vector<char> &buf) {  // the newlines of one chunk
    const auto t0 = now();
    const uint64_t lo = lf[e].chunk_begin[k], hi = lf[e].chunk_begin[k + 1];
    if (lf[e].cmap) {  // BGZF: this chunk's blocks are inflated into their place in the text first
      thread_local BgzfInflater inf;
      LineFile &F = lf[e];
      for (uint32_t b = F.chunk_first_block[k]; b < F.chunk_first_block[k + 1]; ++b) {
        uint32_t data_off = 0;
        const uint32_t len = bgzf_block_length(F.cmap + F.block_at[b], F.csize - F.block_at[b], data_off);
        if (!len) throw std::runtime_error("corrupt BGZF block in " + opt.reads[e]);
        // (blocks without text were left out of the list: the next listed block begins where this one's text ends)
        inf.block(F.cmap + F.block_at[b], len, data_off, const_cast<char *>(F.map) + F.block_text[b], static_cast<uint32_t>(F.block_text[b + 1] - F.block_text[b]));
      }
    }
    const char *base = lf[e].map ? lf[e].map + lo : nullptr;
    if (!base) {
      buf.resize(kChunk);
      read_range(lf[e].fd, opt.reads[e], buf.data(), lo, hi);
      base = buf.data();
    }
    ChunkInfo ci;
    const char *p = base, *end = p + (hi - lo);
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
        if (--until_mark == 0) { ci.marks.push_back(static_cast<uint32_t>(q - base)); until_mark = kMark; }
      }
      p = bend;
    }
    ci.ready = true;
    const double dt = since(t0);
    {
      std::lock_guard<std::mutex> lk(mu);
      lf[e].chunks[k] = std::move(ci);
      ++lf[e].n_ready;
      busy_split += dt;
      if (lf[e].n_ready == lf[e].n_chunks) count_done_s = std::max(count_done_s, since(t_start));
    }
    cv_chunk.notify_all();
  };
  // byte offset just past newline number `line` (1-based count of newlines) of file e; chunks up to the one
  // holding it must be ready.  cum[k] = newlines before chunk k.
  struct Cursor { uint64_t chunk = 0, cum = 0; };  // first chunk not yet passed, newlines before it
  auto offset_after_line = [&](size_t e, Cursor &cur, uint64_t line, uint64_t &off_out) -> bool {
    // returns false if the file has fewer newlines
    LineFile &F = lf[e];
    if (line == 0) { off_out = 0; return true; }
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
    const uint64_t base = F.chunk_begin[cur.chunk], hi = F.chunk_begin[cur.chunk + 1];
    std::vector<char> buf;
    while (seen < local) {  // walk the <= kMark lines after the mark
      const uint64_t want = std::min<uint64_t>(hi - (base + at), 1u << 16);
      if (want == 0) return false;
      const char *text = F.map ? F.map + base + at : nullptr;
      if (!text) {
        buf.resize(want);
        read_range(F.fd, opt.reads[e], buf.data(), base + at, base + at + want);
        text = buf.data();
      }
      const char *p = text, *end = p + want;
      while (p < end && seen < local) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        if (!nl) { p = end; break; }
        ++seen;
        p = nl + 1;
      }
      at += static_cast<uint64_t>(p - text);
    }
    off_out = base + at;
    return true;
  };
  // Several regions: where each begins is known once the first file's lines have all been counted (counting is the
  // workers' first job then, and takes a fraction of a second for tens of gigabytes): region r takes the records
  // [records * r / R, records * (r + 1) / R), rounded down to whole slices.
  auto region_start_record = [&](int r) -> uint64_t {
    if (r <= 0) return 0;
    if (r >= n_regions) return ~0ull >> 3;
    uint64_t lines = 0;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_chunk.wait(lk, [&] { return failure || lf[0].n_ready == lf[0].n_chunks; });
      if (failure) throw std::runtime_error("aborted");
      for (const ChunkInfo &ci : lf[0].chunks) lines += ci.lines;
    }
    if (!lf[0].ends_with_newline) ++lines;
    const uint64_t records = (lines + 3) / 4;
    return records * static_cast<uint64_t>(r) / static_cast<uint64_t>(n_regions) / slice_reads * slice_reads;
  };
  // the input before a region is parsed for the region's lead-in kLeadRecords records at a time, backwards, until the
  // records found cover every position a 44-46-base read can look at (ghost_closed), the input begins, or kLeadBudget
  // records have been looked at: the records a one-file run would still be carrying at that point (round 4
  // looked at the last 4096 only).  A library of one read length below 110 bases never closes the scan -- its lead-in
  // is the last record, whatever lies before -- hence the budget: a quarter of a million records are 60 ms of parsing.
  static const uint64_t kLeadRecords = [] { const char *e = std::getenv("ABM_CLI_LEAD_RECORDS"); return e ? std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) : uint64_t(4096); }();
  static const uint64_t kLeadBudget = [] { const char *e = std::getenv("ABM_CLI_LEAD_BUDGET"); return e ? std::max<uint64_t>(1, std::strtoull(e, nullptr, 10)) : uint64_t(1) << 18; }();
  auto cutter_plain = [&](int r) {
    Region &R = regions[r];
    try {
      Cursor cur[2];
      uint64_t lo[2] = {0, 0};
      bool done[2] = {false, false};
      const size_t nf = lf.size();
      const uint64_t first_rec = region_start_record(r), end_rec = region_start_record(r + 1);
      uint64_t line = 4 * first_rec;
      const uint64_t end_line = 4 * end_rec;
      if (first_rec > 0) {
        // lead-in: what a 44-46-base read at the region's start finds past its end comes from the reads before it
        std::vector<std::string> lead[2];  // (nearest record first while it is collected)
        uint32_t reach[2];
        ghost_reach_start(reach, ends);
        bool input_short = false;
        uint64_t scanned = 0;
        for (uint64_t upto = first_rec; upto > 0 && !ghost_closed(reach) && !input_short && scanned < kLeadBudget;) {
          const uint64_t lead_rec = upto > kLeadRecords ? upto - kLeadRecords : 0;
          std::vector<NameRef> names; RawBuf blob[2], raw; std::vector<uint64_t> off[2];
          for (size_t e = 0; e < nf; ++e) {
            Cursor c;  // (a window further back than the last one: its own cursor, from the file's first chunk)
            uint64_t a = 0, b = 0;
            if (!offset_after_line(e, c, 4 * lead_rec, a) || !offset_after_line(e, c, 4 * upto, b)) {
              if (upto == first_rec) { done[e] = true; lo[e] = lf[e].size; }
              input_short = true;
              continue;
            }
            if (lf[e].map) parse_raw(lf[e].map + a, b - a, 4 * lead_rec, opt.reads[e], names, blob[e], off[e]);
            else {
              raw.resize(b - a);
              read_range(lf[e].fd, opt.reads[e], raw.p, a, b);
              parse_raw(raw, 4 * lead_rec, opt.reads[e], names, blob[e], off[e]);
            }
            if (upto == first_rec) lo[e] = b;
          }
          const size_t m = off[0].empty() ? 0 : off[0].size() - 1;
          if (!input_short && (nf == 1 || (off[1].size() == off[0].size())))
            for (uint32_t k : ghost_tail(off, m, ends, reach)) for (int e = 0; e < ends; ++e) lead[e].emplace_back(blob[e].data() + off[e][k], off[e][k + 1] - off[e][k]);
          scanned += upto - lead_rec;
          upto = lead_rec;
        }
        for (int e = 0; e < ends; ++e) std::reverse(lead[e].begin(), lead[e].end());
        std::lock_guard<std::mutex> lk(mu);
        region_lead_scanned = std::max(region_lead_scanned, scanned);
        region_lead_records = std::max<uint64_t>(region_lead_records, lead[0].size());
        R.carry[0] = std::move(lead[0]);
        R.carry[1] = std::move(lead[1]);
      }
      for (uint64_t g = 0; line < end_line; ++g) {
        std::unique_ptr<Slice> sl = new_slice(R, g);
        const uint64_t target = std::min(line + 4 * slice_reads, end_line);  // newlines before the next slice
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
        if (!any) { slice_pool[sl->node].put(std::move(sl)); break; }
        // (the last slice may hold fewer records; the figure only bounds memory in flight)
        const bool last = done[0] || (nf == 2 && done[1]);
        if (!emit_slice(R, std::move(sl), slice_reads)) break;
        line = target;
        if (last) break;
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    R.cut_done = true;
    wake_everyone();
  };

  auto parse_slice = [&](std::unique_ptr<Slice> sl) {
    const auto t0 = now();
    Region &R = regions[sl->region];
    for (int e = 0; e < ends; ++e) {
      if (plain_input && lf[e].map) {  // parsed in place: the names are views into the mapping
        parse_raw(lf[e].map + sl->byte_lo[e], sl->byte_hi[e] - sl->byte_lo[e], sl->first_line[e], opt.reads[e], sl->names[e], sl->blob[e], sl->off[e]);
        continue;
      }
      if (plain_input) {
        if (!sl->raw[e].p) sl->raw[e] = raw_pool[sl->node].get();
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
    sl->tail = ghost_tail(sl->off, sl->n(), ends);
    const double dt = since(t0);
    const auto tl = now();
    {
      std::lock_guard<std::mutex> lk(mu);
      lock_wait += since(tl);
      busy_parse += dt;
      const uint64_t g = sl->g;
      trace("parsed", g, static_cast<uint64_t>(dt * 1e6));
      R.parsed[g] = std::move(sl);
      ++R.n_parsed;
      for (auto it = R.parsed.find(R.run_end); it != R.parsed.end(); it = R.parsed.find(R.run_end)) { R.run_reads += it->second->n(); ++R.run_end; }
    }
    R.cv_map.notify_all();
  };

  auto mapper = [&](int slot) {
    const int g = slot / per_gpu;
    const int node = gpu_node[g];
    topo.pin(node, 1);  // (a handful of mapper threads per node: they sleep in the library while their kernel runs)
    abm_ctx *ctx = virtual_gpus ? nullptr : ctxs[slot];
    Region &R = regions[region_of(g, slot % per_gpu)];
    try {
      for (;;) {
        std::unique_ptr<Batch> owned = batch_pool[node].get();
        Batch *b = owned.get();
        {
          std::unique_lock<std::mutex> lk(mu);
          auto all_parsed = [&] { return R.cut_done && R.n_parsed == R.n_slices; };
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
          // profiles/r03_exp_e2e_first_batch.log).  Several GPUs on one region: what is left is shared among them.
          auto target = [&]() -> size_t {
            size_t cap = batch_reads;
            const uint64_t grown = std::min<uint64_t>(gpu_batches[g], 10) * (stream_slices ? 2 : 1);
            if (first_batch_reads)  // (saturating: a huge first batch shifted left would wrap)
              cap = std::min<size_t>(batch_reads, std::max<size_t>(slice_reads, first_batch_reads >= (batch_reads >> grown) ? batch_reads : first_batch_reads << grown));
            if (!R.cut_done) return cap;
            const size_t left = static_cast<size_t>(R.n_slices - R.next_to_map) * slice_reads;
            // (pairs: a batch's launch is as long as its costliest pair -- a second for a pair with two 32768-entry sets, of
            // which a million pairs hold some five hundred -- and nothing hides the LAST batch's.  Tapering the batches off
            // towards the end of the input (each at most 1 / ABM_CLI_PE_TAPER of what is left) was tried and COSTS: 8 M pairs
            // 3.80-3.86 M reads/s without, 3.45-3.73 with a quarter, 3.2 with an eighth -- more batches, the same worst pair;
            // profiles/r05_pe_e2e_taper.log.  Off by default.)
            if (paired && pe_taper) cap = std::min(cap, std::max<size_t>(slice_reads, (left / pe_taper + slice_reads - 1) / slice_reads * slice_reads));
            size_t k = (left + cap - 1) / cap;
            if (k <= 1) k = (!stream_slices && left >= (1u << 22)) ? 2 : 1;
            k = std::max<size_t>(k, std::min<size_t>(static_cast<size_t>(R.mappers_live + per_gpu - 1) / per_gpu, (left + (1u << 20) - 1) >> 20));
            // (pairs: what is left is shared among the region's mapper threads, so that their batches' tails overlap)
            if (paired) k = std::max<size_t>(k, std::min<size_t>(static_cast<size_t>(R.mappers_live), (left + (1u << 18) - 1) >> 18));
            return std::max<size_t>(slice_reads, (left + k - 1) / std::max<size_t>(k, 1));
          };
          R.cv_map.wait(lk, [&] { return failure || R.run_reads >= target() || (count_gated && R.run_end > R.next_to_map && R.n_parsed == R.n_slices) ||
                                         (all_parsed() && (R.run_end > R.next_to_map || R.parsed.empty())); });
          if (failure || R.run_end == R.next_to_map) {
            // (the unused batch goes back to the pool: destroying it here would free its page-locked buffers -- a
            // device-wide wait and 0.1-0.2 s of unpinning -- inside the run's clock; the trace showed the run's end
            // waiting on exactly these two threads)
            lk.unlock();
            batch_pool[node].put(std::move(owned));
            break;
          }
          const size_t want = std::min(target(), std::max<size_t>(R.run_reads, 1));
          while (R.next_to_map < R.run_end) {
            auto it = R.parsed.find(R.next_to_map);
            if (!b->slices.empty() && b->n + it->second->n() > want) break;
            b->n += it->second->n();
            R.run_reads -= it->second->n();
            b->slices.push_back(std::move(it->second));
            R.parsed.erase(it);
            ++R.next_to_map;
          }
          b->seq = n_batches++;
          { const int rr = region_of(g, slot % per_gpu); if (region_first_batch_s[rr] < 0) region_first_batch_s[rr] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); }
          b->gpu = g;
          b->node = node;
          ++gpu_batches[g];
          gpu_reads[g] += b->n;
          b->slices_left = static_cast<int>(b->slices.size());
          // Reads of 44-46 bases see what earlier reads left in the reference's reused buffers: for each
          // position past its end (up to 64 of them: 110 bases out) the nearest EARLIER read that is longer.  The mapper
          // looks for that among the reads handed over in the same call, so a batch is led by the records of the input
          // before it that can still be such a source: scanning backwards, every record that is longer (in either end)
          // than everything after it, until one reaches 110 bases -- for a library of one read length just the last
          // record, and never more than 2 x 67.  Each slice's own list was made when it was parsed (ghost_tail); here
          // the lists are merged, newest slice first, then what the previous lead-in still contributes.
          b->carry[0] = R.carry[0];
          b->carry[1] = R.carry[1];
          {
            std::vector<std::string> next[2];
            uint32_t reach[2] = {0, 0};
            auto closed = [&] { return reach[0] >= kGhostReach && (!paired || reach[1] >= kGhostReach); };
            for (size_t si = b->slices.size(); si-- > 0 && !closed();) {
              const Slice &sl = *b->slices[si];
              for (uint32_t k : sl.tail) {
                if (closed()) break;
                bool raises = false;
                for (int e = 0; e < ends; ++e) {
                  const uint32_t len = ghost_len(sl.off[e][k + 1] - sl.off[e][k]);
                  if (len > reach[e]) { raises = true; reach[e] = len; }
                }
                if (raises) for (int e = 0; e < ends; ++e) next[e].emplace_back(sl.blob[e].data() + sl.off[e][k], sl.off[e][k + 1] - sl.off[e][k]);
              }
            }
            for (size_t k = R.carry[0].size(); k-- > 0 && !closed();) {
              bool raises = false;
              for (int e = 0; e < ends; ++e) {
                const uint32_t len = ghost_len(R.carry[e][k].size());
                if (len > reach[e]) { raises = true; reach[e] = len; }
              }
              if (raises) for (int e = 0; e < ends; ++e) next[e].push_back(R.carry[e][k]);
            }
            for (int e = 0; e < ends; ++e) { std::reverse(next[e].begin(), next[e].end()); R.carry[e].swap(next[e]); }
          }
          live_batches.push_back(std::move(owned));
        }
        const size_t lead = b->carry[0].size();
        const size_t n = b->n + lead;
        { std::lock_guard<std::mutex> lk(mu); max_lead = std::max(max_lead, lead); }
        const uint64_t seq_no = b->seq;  // (a batch whose slices were handed over during the call may be recycled before it returns)
        bool queued = false;
        const auto t0 = now();
        trace("batch formed", b->seq, n);
        // the slices' reads, concatenated as the C ABI takes them (a single slice with nothing to lead it is used in place)
        const bool one = b->slices.size() == 1 && lead == 0;
        for (int e = 0; e < ends; ++e) {
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
          const size_t n_copy = std::min<size_t>(std::max<size_t>(1, std::min<size_t>(n_host / 4, 8)), std::max<size_t>(1, b->slices.size() / 4));
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
        auto queue_slice = [&](Slice &sl) {  // a slice whose results it holds itself goes to the formatters of its node
          std::lock_guard<std::mutex> lk(mu);
          nq[sl.node].format.push_back(&sl);
          wake_worker(sl.node);
        };
        if (n) {
          // a few CIGAR ops per read are typical; the worst case (read length + 2 each) is only
          // allocated if the first size turns out too small
          const uint64_t worst = std::max<uint64_t>(1, std::max(blob_n[0], blob_n[1]) + 2 * n);
          uint64_t cap = std::min<uint64_t>(worst, 4 * n + 1024);
          queued = false;
          for (;;) {
            int rc;
            if (virtual_gpus) {
              // diagnostic: what the pipeline around the GPUs can carry.  Every read "maps" somewhere inside the first
              // chromosome with one mismatch and a single-op CIGAR; nothing is sent to a GPU.  The hits are handed over
              // the way the library hands real ones over: slice by slice into the slice's own arrays (or, with
              // ABM_CLI_NO_STREAM, as one batch).
              const uint32_t c0 = ch.starts.size() > 2 ? ch.starts[1] : 0, c1 = ch.starts.size() > 2 ? ch.starts[2] : 0;
              const uint32_t span = c1 > c0 + 70000 ? c1 - c0 - 66000 : 1;
              auto made_up = [&](size_t i) {
                const uint32_t len = static_cast<uint32_t>(off_p[0][i + 1] - off_p[0][i]);
                abm_hit h;
                uint64_t key = 0;  // (a function of the read alone: the output does not depend on how batches were dealt)
                if (len >= 8) std::memcpy(&key, blob_p[0] + off_p[0][i] + len / 2 - 4, 8);
                key = (key ^ (key >> 29)) * 0x9E3779B97F4A7C15ull;
                h.diffs = 1; h.flags = (key >> 40 & 1) ? 0x10 : 0; h.pos = len ? c0 + static_cast<uint32_t>((key >> 8) % span) : 0;
                return h;
              };
              if (stream_slices) {
                // (an infinitely fast GPU: the slice's made-up hits are written by whichever worker formats it, from
                // its own reads -- filled in here, 8 M reads of a batch took this one thread 0.25 s, which a run of
                // 0.6 s then waited for at its end)
                for (auto &slp : b->slices) {
                  Slice &sl = *slp;
                  const size_t m = sl.n();
                  sl.own_se.resize(std::max<size_t>(m, 1)); sl.own_cig.resize(m + 1); sl.own_cig_off.resize(m + 1);
                  sl.own = true;
                  sl.virt = true;
                  queue_slice(sl);
                }
                queued = true;
              }
              else {
                cap = std::max<uint64_t>(cap, n);
                b->se[0].resize(n); b->cig[0].resize(cap); b->cig_off[0].resize(n + 1);
                for (size_t i = 0; i < n; ++i) {
                  b->se[0][i] = made_up(i);
                  b->cig[0][i] = static_cast<uint32_t>(off_p[0][i + 1] - off_p[0][i]) << 4;
                  b->cig_off[0][i] = i;
                }
                b->cig_off[0][n] = n;
              }
              rc = 0;
            }
            else if (!paired && stream_slices) {
              // results arrive slice by slice while the kernel runs: each slice takes its own copy inside the callback
              // and goes straight to the formatters
              std::vector<uint64_t> first(b->slices.size() + 1);
              first[0] = lead;
              for (size_t k = 0; k < b->slices.size(); ++k) first[k + 1] = first[k] + b->slices[k]->n();
              struct Taker {
                abm_ctx *ctx; Batch *b; const std::vector<uint64_t> *first; decltype(queue_slice) *queue; int rc; std::string err; bool device_sam;
              } taker{ctx, b, &first, &queue_slice, 0, std::string(), device_sam};
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
                sl.has_tails = false;
                if (t.rc == 0 && t.device_sam) {
                  const char *tails = nullptr; const uint32_t *lens = nullptr; uint32_t stride = 0;
                  if (abm_ctx_slice_sam_tails(t.ctx, lo, hi, &tails, &stride, &lens) != 0) { t.rc = -1; t.err = abm_last_error(); }
                  else if (tails) {
                    sl.tail_len.resize(std::max<uint64_t>(m, 1));
                    sl.tail_text.clear();
                    sl.tail_text.reserve(m * stride);
                    char *w = sl.tail_text.p;
                    for (uint64_t k = 0; k < m; ++k) {
                      const uint32_t len = lens[k];
                      sl.tail_len[k] = len;
                      if (len != 0 && len != 0xFFFFFFFFu) { std::memcpy(w, tails + k * stride, len); w += len; }
                    }
                    sl.tail_text.n = static_cast<size_t>(w - sl.tail_text.p);
                    sl.has_tails = true;
                  }
                }
                (*t.queue)(sl);
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
        const double dt = since(t0);
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_map += dt;
          if (!queued)
            for (auto &sl : b->slices) { nq[sl->node].format.push_back(sl.get()); wake_worker(sl->node); }
        }
      }
    }
    catch (...) { fail(); }
    std::lock_guard<std::mutex> lk(mu);
    --mappers_live;
    --R.mappers_live;
    for (NodeQueues &q : nq) q.cv.notify_all();
    R.cv_map.notify_all();  // (the region's other mappers share out what is left by their number)
    R.cv_write.notify_all();
  };

  auto fill_virtual = [&](Slice &sl) {  // a virtual GPU's results for one slice: a function of each read alone
    const uint32_t c0 = ch.starts.size() > 2 ? ch.starts[1] : 0, c1 = ch.starts.size() > 2 ? ch.starts[2] : 0;
    const uint32_t span = c1 > c0 + 70000 ? c1 - c0 - 66000 : 1;
    const size_t m = sl.n();
    for (size_t i = 0; i < m; ++i) {
      const uint32_t len = static_cast<uint32_t>(sl.off[0][i + 1] - sl.off[0][i]);
      abm_hit h;
      uint64_t key = 0;
      if (len >= 8) std::memcpy(&key, sl.blob[0].data() + sl.off[0][i] + len / 2 - 4, 8);
      key = (key ^ (key >> 29)) * 0x9E3779B97F4A7C15ull;
      h.diffs = 1; h.flags = (key >> 40 & 1) ? 0x10 : 0; h.pos = len ? c0 + static_cast<uint32_t>((key >> 8) % span) : 0;
      sl.own_se[i] = h;
      sl.own_cig[i] = len << 4;
      sl.own_cig_off[i] = i;
    }
    sl.own_cig_off[m] = m;
  };
  auto format_slice = [&](Slice &sl) {
    const Batch *b = sl.batch;
    t_bam = opt.bam;
    if (sl.virt) fill_virtual(sl);
    RawBuf &sam = sl.text;
    Stats3 &st = sl.stats;
    const size_t m = sl.n(), base = sl.base;
    sam.reserve(m * (paired ? 2 : 1) * 320);
    if (!paired) {
      // (a slice that took its own copy of the results indexes it by its own read numbers)
      const abm_hit *hits = sl.own ? sl.own_se.data() : b->se[0].data() + base;
      const uint32_t *cig_blob = sl.own ? sl.own_cig.data() : b->cig[0].data();
      const uint64_t *cig_off = sl.own ? sl.own_cig_off.data() : b->cig_off[0].data() + base;
      const char *tail_at = sl.has_tails ? sl.tail_text.p : nullptr;
      for (size_t k = 0; k < m; ++k) {
        abm_hit h = hits[k];
        const size_t len = sl.off[0][k + 1] - sl.off[0][k];
        const uint32_t *cg = cig_blob + cig_off[k];
        const size_t ncg = cig_off[k + 1] - cig_off[k];
        const uint32_t tl = sl.has_tails ? sl.tail_len[k] : 0xFFFFFFFFu;
        if (tl == 0xFFFFFFFFu) {  // (no text from the device for this read: format_se here)
          if (len && emit_se(sam, opt.ambig, h, ch, sl.names[0][k], sl.blob[0].data() + sl.off[0][k], len, cg, ncg) == UNMAPPED) h.pos = 0;
        }
        else if (len && (opt.ambig || !(h.flags & 0x100))) {
          // the kernel wrote the line after QNAME (or found no record: no hit, or one that runs across its chromosome's end)
          if (tl == 0) h.pos = 0;
          else {
            const NameRef &nm = sl.names[0][k];
            const size_t at0 = sam.size();
            sam.resize(at0 + nm.n + tl);
            std::memcpy(&sam[at0], nm.p, nm.n);
            std::memcpy(&sam[at0 + nm.n], tail_at, tl);
          }
        }
        if (tl != 0xFFFFFFFFu) tail_at += tl;
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
  // formats a slice; its place in its region's file is fixed once every earlier slice's size is known
  auto format_task = [&](Slice *sl) {
    const auto t0 = now();
    format_slice(*sl);
    if (opt.bam) {  // (the block stream is built in a buffer the thread keeps, then trades places with the slice's text)
      thread_local RawBuf z;
      z.clear();
      z.reserve(sl->text.size() / 2 + (1u << 16));
      bgzf_compress(sl->text, z);
      sl->text.swap(z);
    }
    const double dt = since(t0);
    Region &R = regions[sl->region];
    const auto tl = now();
    std::lock_guard<std::mutex> lk(mu);
    lock_wait += since(tl);
    busy_format += dt;
    trace("formatted", sl->g, static_cast<uint64_t>(dt * 1e6));
    R.formatted[sl->g] = sl;
    bool placed = false;
    for (auto it = R.formatted.find(R.next_to_place); it != R.formatted.end(); it = R.formatted.find(R.next_to_place)) {
      it->second->place = R.file_offset;
      R.file_offset += it->second->text.size();
      R.q_write.push_back(it->second);
      R.formatted.erase(it);
      ++R.next_to_place;
      placed = true;
    }
    if (placed) R.cv_write.notify_one();
  };

  // host workers: -t of them, pinned node by node; each takes the most urgent task its node has, else another node's
  auto worker = [&](int node) {
    topo.pin(node, workers_on[node]);
    std::vector<char> count_buf;
    const bool count_first = n_regions > 1;  // (the later regions cannot start before the whole input has been counted)
    try {
      for (;;) {
        std::unique_ptr<Slice> to_parse;
        Slice *to_format = nullptr;
        size_t ce = 0; uint64_t ck = 0; bool to_count = false;
        {
          std::unique_lock<std::mutex> lk(mu);
          for (;;) {
            if (failure) return;
            if (count_first && plain_input && take_chunk(ce, ck)) { to_count = true; break; }
            auto take_from = [&](int n) {
              if (!nq[n].format.empty()) { to_format = nq[n].format.front(); nq[n].format.pop_front(); return true; }
              if (!nq[n].parse.empty()) { to_parse = std::move(nq[n].parse.front()); nq[n].parse.pop_front(); return true; }
              return false;
            };
            if (take_from(node)) break;
            if (!count_first && plain_input && take_chunk(ce, ck)) { to_count = true; break; }
            bool got = false;
            for (int k = 1; k < n_nodes && !got; ++k) got = take_from((node + k) % n_nodes);
            if (got) break;
            if (mappers_live == 0) return;  // (nothing queued and nobody left to queue anything)
            ++nq[node].idle;
            nq[node].cv.wait(lk);
            --nq[node].idle;
            if (nq[node].waking > 0) --nq[node].waking;
          }
        }
        if (to_count) count_chunk(ce, ck, count_buf);
        else if (to_format) format_task(to_format);
        else parse_slice(std::move(to_parse));
      }
    }
    catch (...) { fail(); }
  };
  // ONE writer per output file: slices leave in order, each with a single pwrite at its place (or write, into a pipe).
  // One file takes the same rate from one thread as from several (they queue for its lock: 6.5 GB/s on the GPU box's
  // tmpfs with the source in DRAM, profiles/r04_sink_probe.log; round 2 had four writers among the formatting threads).
  auto writer = [&](int r) {
    Region &R = regions[r];
    topo.pin(R.writer_node, 1);
    try {
      for (;;) {
        Slice *to_write = nullptr;
        {
          std::unique_lock<std::mutex> lk(mu);
          R.cv_write.wait(lk, [&] { return failure || !R.q_write.empty() || (R.mappers_live == 0 && R.cut_done && R.slices_written == R.n_slices); });
          if (failure || R.q_write.empty()) break;
          to_write = R.q_write.front(); R.q_write.pop_front();
        }
        const auto t0 = now();
        write_all(R, to_write->text.data(), to_write->text.size(), to_write->place);
        const double dt = since(t0);
        std::unique_ptr<Batch> done_batch;
        std::vector<std::unique_ptr<Slice>> done_slices;
        {
          std::lock_guard<std::mutex> lk(mu);
          busy_write += dt;
          trace("written", to_write->g, static_cast<uint64_t>(dt * 1e6));
          Batch *b = to_write->batch;
          R.records += to_write->n();
          for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 6; ++j) gpu_stats[b->gpu].s[k].v[j] += to_write->stats.s[k].v[j];
          R.reads_in_flight -= std::min<size_t>(R.reads_in_flight, slice_reads);
          ++R.slices_written;
          for (size_t e = 0; e < lf.size(); ++e)
            if (lf[e].cmap) {  // BGZF: the inflated text behind this slice goes back to the system (whole pages of it)
              LineFile &F = lf[e];
              const uint64_t from = (F.released_upto + 4095) & ~4095ull, to = to_write->byte_hi[e] & ~4095ull;
              if (to > from) ::madvise(const_cast<char *>(F.map) + from, to - from, MADV_DONTNEED);
              F.released_upto = std::max(F.released_upto, to_write->byte_hi[e]);
              for (NodeQueues &q : nq) if (q.idle > 0) q.cv.notify_one();  // (chunks further on may be inflated now)
            }
          // the batch goes when its last slice is written; its slices and its own buffers are recycled
          if (--b->slices_left == 0) {
            for (auto it = live_batches.begin(); it != live_batches.end(); ++it)
              if (it->get() == b) {
                done_slices.swap((*it)->slices);
                done_batch = std::move(*it);
                live_batches.erase(it);
                break;
              }
          }
          R.cv_flow.notify_one();
          if (R.slices_written == R.n_slices) R.cv_write.notify_all();
        }
        for (auto &sl : done_slices) { const int n = sl->node; slice_pool[n].put(std::move(sl)); }
        if (done_batch) { const int n = done_batch->node; batch_pool[n].put(std::move(done_batch)); }
      }
    }
    catch (...) { fail(); }
  };

  std::vector<std::thread> threads;
  if (plain_input) for (int r = 0; r < n_regions; ++r) threads.emplace_back(cutter_plain, r);
  else threads.emplace_back(cutter_stream);
  for (unsigned t = 0; t < n_host; ++t) threads.emplace_back(worker, static_cast<int>(t % n_nodes));
  for (int slot = 0; slot < n_gpus * per_gpu; ++slot) threads.emplace_back(mapper, slot);
  for (int r = 0; r < n_regions; ++r) threads.emplace_back(writer, r);
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
  if (opt.bam) {  // BGZF end-of-file marker (closes the last part)
    static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    Region &R = regions.back();
    write_all(R, reinterpret_cast<const char *>(eof_block), 28, R.file_offset);
    R.file_offset += 28;
  }
  uint64_t out_bytes = 0;
  for (Region &R : regions) {
    const int fd = R.fd;
    R.fd = -1;
    out_bytes += R.file_offset;
    if (::close(fd) != 0) throw std::runtime_error("failed writing output file: " + R.path);
  }
  trace("output closed", out_bytes, 0);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  struct rusage ru1;
  ::getrusage(RUSAGE_SELF, &ru1);
  auto tv = [](const timeval &a, const timeval &b) { return static_cast<double>(a.tv_sec - b.tv_sec) + 1e-6 * static_cast<double>(a.tv_usec - b.tv_usec); };
  const double cpu_user = tv(ru1.ru_utime, ru0.ru_utime), cpu_sys = tv(ru1.ru_stime, ru0.ru_stime);
  uint64_t throttled1 = 0; double throttled_s1 = 0;
  quota.throttled(throttled1, throttled_s1);

  // statistics (6 counters x 3 structs, src/abismal.cpp:865-895, :1034-1037).  Every GPU's counters
  // already sit in this process, so the total is a host sum; with more than one GPU the same sum is
  // also taken with the path's one collective (RCCL all-reduce over xGMI, abm_stats_allreduce) and
  // the two must agree.  A collective that cannot run (no RCCL transport on this box) costs a
  // warning, never the statistics file of a finished run.
  static_assert(sizeof(Stats3) == 18 * sizeof(uint64_t), "18 counters");
  Stats3 tot;
  uint64_t total_records = 0;
  for (const Region &R : regions) total_records += R.records;
  for (const Stats3 &g : gpu_stats)
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 6; ++j) tot.s[k].v[j] += g.s[k].v[j];
  if (n_gpus > 1 && !virtual_gpus && !std::getenv("ABM_CLI_NO_RCCL")) {
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
      if (opt.verbose)
        std::cerr << "[abismal-amd] statistics summed over " << n_gpus << " GPUs with "
                  << (shared_device ? "a host sum inside abm_stats_allreduce (replicas share a device)\n" : "one RCCL all-reduce\n");
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
       << ", \"sam_text_by\": \"" << (device_sam ? "device" : "host") << "\", \"window_records_serve_reads_up_to\": " << (ctxs.empty() ? 0u : abm_ctx_window_records(ctxs[0])) << ", \"host_threads\": " << n_host << ", \"numa_nodes\": " << n_nodes << ", \"pinned\": " << (topo.pinning ? "true" : "false")
       << ", \"out_parts\": " << n_regions << ", \"out_bytes\": " << out_bytes << ", \"batches\": " << n_batches << ", \"max_lead_in_records\": " << max_lead << ", \"region_lead_in_scanned_records\": " << region_lead_scanned << ", \"region_lead_in_records\": " << region_lead_records
       << ", \"batch_reads\": " << batch_reads << ", \"host_ceiling\": " << (virtual_gpus ? "true" : "false")
       << ", \"batches_per_gpu\": [";
    for (int g = 0; g < n_gpus; ++g) tj << (g ? ", " : "") << gpu_batches[g];
    tj << "], \"reads_per_gpu\": [";
    for (int g = 0; g < n_gpus; ++g) tj << (g ? ", " : "") << gpu_reads[g];
    tj << "], \"region_first_batch_s\": [";
    for (int r = 0; r < n_regions; ++r) tj << (r ? ", " : "") << region_first_batch_s[r];
    struct rusage ru_end;
    getrusage(RUSAGE_SELF, &ru_end);
    uint64_t lib_pinned = 0;
    for (abm_ctx *c : ctxs) lib_pinned += abm_ctx_pinned_bytes(c);
    tj << "], \"count_done_s\": " << count_done_s << ", \"peak_rss_mb\": " << (ru_end.ru_maxrss >> 10)
       << ", \"pinned_mb\": {\"batches\": " << (g_pinned_bytes.load() >> 20) << ", \"library_contexts\": " << (lib_pinned >> 20) << "}";
    tj << ", \"cpu_quota_cpus\": " << quota.cpus << ", \"throttled_periods\": " << (throttled1 - throttled0) << ", \"throttled_s\": " << (throttled_s1 - throttled_s0)
       << ", \"cpu_s\": {\"user\": " << cpu_user << ", \"sys\": " << cpu_sys << "}, \"busy_s\": {\"split\": " << busy_split
       << ", \"parse\": " << busy_parse << ", \"map\": " << busy_map << ", \"format\": " << busy_format << ", \"write\": "
       << busy_write << ", \"lock_wait\": " << lock_wait << "}}\n";
  }
  if (opt.verbose)
    std::cerr << "[abismal-amd] " << total_records << (paired ? " pairs" : " reads") << " on " << n_gpus << " GPU(s) in "
              << secs << " s (" << (paired ? 2 : 1) * total_records / secs << " reads/s incl. host I/O)\n"
              << "[abismal-amd] busy seconds: count " << busy_split << ", parse " << busy_parse << ", format " << busy_format << " ("
              << n_host << " host workers on " << n_nodes << " NUMA node(s)), map " << busy_map << " (" << n_gpus * per_gpu
              << " threads), write " << busy_write << " (" << n_regions << " file(s)); process CPU " << cpu_user << " s user + " << cpu_sys << " s system"
              << "; CPU quota " << quota.cpus << " CPUs (0 = none), throttled in " << (throttled1 - throttled0) << " periods for " << (throttled_s1 - throttled_s0) << " s\n";
  if (opt.verbose)
    for (int g = 0; g < n_gpus; ++g)
      std::cerr << "[abismal-amd] GPU " << g << (virtual_gpus ? " (virtual)" : "") << ": " << gpu_batches[g] << " batches, " << gpu_reads[g] << (paired ? " pairs\n" : " reads\n");
  // Reads (and ends of pairs) of any length the reference takes are mapped; longer ones stop the run while the input is
  // parsed, with the reference's message -- so the library's count of reads beyond its supported length stays zero here
  // (kept as a safeguard: a non-zero count fails the run unless -skip-long accepts it).
  uint64_t too_long = 0;
  for (abm_ctx *c : ctxs) too_long += abm_ctx_reads_too_long(c);
  if (too_long)
    std::cerr << "[abismal-amd] " << (opt.skip_long ? "warning: " : "error: ") << too_long << (paired ? " pairs" : " reads")
              << " beyond the supported read length were not mapped (written as unmapped" << (opt.skip_long ? ")\n" : "); -skip-long accepts this\n");
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
    if (cmd == "host") {  // what the host pipeline sees of the machine: NUMA nodes, cores, the container's CPU quota, its default workers
      const Topology topo;
      const CpuQuota quota;
      std::cout << "numa_nodes: " << topo.n_nodes() << "\n";
      for (int n = 0; n < topo.n_nodes(); ++n)
        std::cout << "  node" << n << ": " << topo.primary[n].size() << " cores, " << topo.all[n].size() << " hardware threads usable\n";
      std::cout << "pinning: " << (topo.pinning ? "on" : "off") << "\ncpu_quota_cpus: " << quota.cpus << " (0 = none)\n";
      for (int g : {1, 2, 4, 8}) {
        unsigned n_host = static_cast<unsigned>(std::min<size_t>(std::max<size_t>(topo.n_cores(), 1), 8u + 8u * static_cast<unsigned>(g)));
        if (quota.cpus > 0) n_host = std::min(n_host, std::max(1u, static_cast<unsigned>(quota.cpus + 0.5)));
        std::cout << "default host workers with " << g << " GPU(s): " << n_host << "\n";
      }
      std::cout << "hip_devices: " << abm_device_count() << "\n";
      return EXIT_SUCCESS;
    }
    if (cmd == "bgzf") {  // abismal-amd bgzf [-z n] <in> <out>: any file as BGZF blocks, the way -B output is compressed (a test hook)
      std::vector<std::string> pos;
      for (int i = 2; i < argc; ++i) {
        if (std::string(argv[i]) == "-z" && i + 1 < argc) g_bgzf_level = std::max(0, std::min(9, std::atoi(argv[++i])));
        else pos.push_back(argv[i]);
      }
      if (pos.size() != 2) { std::cerr << "usage: abismal-amd bgzf [-z n] <in> <out>\n"; return EXIT_FAILURE; }
      std::ifstream in(pos[0], std::ios::binary);
      if (!in) throw std::runtime_error("cannot open " + pos[0]);
      std::string raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>()), z;
      bgzf_compress(raw, z);
      static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      z.append(reinterpret_cast<const char *>(eof_block), 28);
      std::ofstream out(pos[1], std::ios::binary);
      out.write(z.data(), static_cast<std::streamsize>(z.size()));
      return out ? EXIT_SUCCESS : EXIT_FAILURE;
    }
    std::cerr << "ERROR: invalid command " << cmd << '\n';
    return EXIT_SUCCESS;
  }
  catch (const std::exception &e) {
    std::cerr << e.what() << '\n';
    return EXIT_FAILURE;
  }
}
