// abismal_amd host side: AbismalIndex file as the product consumes it
// (layout of src/AbismalIndex.cpp:1037-1072, reader semantics of :1082-1146).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace abm {

struct HostIndex {
  std::vector<std::string> chrom_names;  // incl. pad_start / pad_end
  std::vector<uint32_t> chrom_starts;    // n_chroms + 1
  uint32_t max_candidates = 100;
  // seed window: 20, or 12 for an index made for short reads (the reference's --enable-short build:
  // configure.ac:70-73, src/AbismalIndex.hpp:73-77).  Stored in the file; the mapper follows it.
  uint32_t window = 20;
  bool multibit_genome = false;  // some genome nibble has 2+ bits (IUPAC code): Hamming sums can go negative
  uint64_t counter_size = 0, counter_size3 = 0, index_size = 0, index_size3 = 0;
  // one contiguous arena so the upload is a handful of large copies
  std::vector<uint64_t> genome;  // + 2 guard words (filter reads one word past)
  std::vector<uint32_t> counter, counter_t, counter_a, index, index_t, index_a;

  void load(const std::string &path);  // throws std::runtime_error
  uint64_t device_bytes() const;
  // ChromLookup::get_chrom_idx_and_offset, src/AbismalIndex.cpp:1305-1320
  bool locate(uint32_t pos, uint32_t reflen, int32_t &chrom, uint32_t &off) const;
};

}  // namespace abm
