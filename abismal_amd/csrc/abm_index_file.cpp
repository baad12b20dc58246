#include "abm_index_file.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>

namespace abm {

namespace {
struct Closer { void operator()(FILE *f) const { if (f) std::fclose(f); } };
template <class T> void need(FILE *f, T *dst, size_t n, const char *what) {
  if (n && std::fread(dst, sizeof(T), n, f) != n)
    throw std::runtime_error(what);
}
}  // namespace

void HostIndex::load(const std::string &path) {
  std::unique_ptr<FILE, Closer> fp(std::fopen(path.c_str(), "rb"));
  if (!fp)
    throw std::runtime_error("cannot open input file " + path);
  FILE *f = fp.get();
  char magic[12];
  if (std::fread(magic, 1, 12, f) != 12 || std::memcmp(magic, "AbismalIndex", 12) != 0)
    throw std::runtime_error("index file format problem: " + path);
  // seed::read, src/AbismalIndex.cpp:988-1024
  uint32_t seed[3];
  need(f, seed, 3, "failed to read seed data");
  if (seed[0] != 25u)
    throw std::runtime_error("inconsistent k-mer size. Expected: 25, got: " + std::to_string(seed[0]));
  if (seed[1] != 20u && seed[1] != 12u)  // (a reference binary takes only the one it was compiled for; this one follows the file)
    throw std::runtime_error("inconsistent window size size. Expected: 20, got: " + std::to_string(seed[1]));
  window = seed[1];
  if (seed[2] != 256u)
    throw std::runtime_error("inconsistent sorting size size. Expected: 256, got: " + std::to_string(seed[2]));
  // ChromLookup::read, :1225-1258
  const char *cerr = "failed loading chrom info from index";
  uint32_t n_chroms = 0;
  need(f, &n_chroms, 1, cerr);
  chrom_names.assign(n_chroms, std::string());
  for (auto &nm : chrom_names) {
    uint32_t len = 0;
    need(f, &len, 1, cerr);
    nm.resize(len);
    need(f, nm.data(), len, cerr);
  }
  chrom_starts.assign(static_cast<size_t>(n_chroms) + 1, 0);
  need(f, chrom_starts.data(), chrom_starts.size(), cerr);

  const char *ierr = "failed loading index file";
  const uint64_t gwords = (static_cast<uint64_t>(chrom_starts.back()) + 15) / 16;
  genome.assign(gwords + 2, 0);
  need(f, genome.data(), gwords, ierr);
  multibit_genome = false;
  for (uint64_t w = 0; w < gwords && !multibit_genome; ++w) {
    // a nibble is one-hot or zero iff clearing its lowest set bit leaves nothing
    const uint64_t x = genome[w];
    const uint64_t low = x & 0x1111111111111111ull, b1 = (x >> 1) & 0x1111111111111111ull,
                   b2 = (x >> 2) & 0x1111111111111111ull, b3 = (x >> 3) & 0x1111111111111111ull;
    if (((low + b1 + b2 + b3) & 0xEEEEEEEEEEEEEEEEull) != 0) multibit_genome = true;
  }
  need(f, &max_candidates, 1, ierr);
  need(f, &counter_size, 1, ierr);
  need(f, &counter_size3, 1, ierr);
  need(f, &index_size, 1, ierr);
  need(f, &index_size3, 1, ierr);
  counter.resize(counter_size + 1);
  counter_t.resize(counter_size3 + 1);
  counter_a.resize(counter_size3 + 1);
  need(f, counter.data(), counter.size(), ierr);
  need(f, counter_t.data(), counter_t.size(), ierr);
  need(f, counter_a.data(), counter_a.size(), ierr);
  index.resize(index_size);
  index_t.resize(index_size3);
  index_a.resize(index_size3);
  need(f, index.data(), index.size(), ierr);
  need(f, index_t.data(), index_t.size(), ierr);
  need(f, index_a.data(), index_a.size(), ierr);
}

uint64_t HostIndex::device_bytes() const {
  return genome.size() * 8 + (counter.size() + counter_t.size() + counter_a.size() + index.size() +
                              index_t.size() + index_a.size()) * 4ull;
}

bool HostIndex::locate(uint32_t pos, uint32_t reflen, int32_t &chrom, uint32_t &off) const {
  auto it = std::upper_bound(chrom_starts.begin(), chrom_starts.end(), pos);
  if (it == chrom_starts.begin())
    return false;
  --it;
  chrom = static_cast<int32_t>(it - chrom_starts.begin());
  off = pos - chrom_starts[chrom];
  return pos + reflen <= chrom_starts[chrom + 1];
}

}  // namespace abm
