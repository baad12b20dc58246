// ORACLE — test infrastructure only (see abo_common.hpp).
// AbismalIndex in-memory layout, file reader/writer and builder restatement.
#ifndef ABO_INDEX_HPP
#define ABO_INDEX_HPP

#include "abo_common.hpp"

namespace abo {

// Field-for-field the on-disk layout of src/AbismalIndex.cpp:1037-1072.
struct Index {
  ChromTable chroms;
  std::vector<u64> genome;  // 16 nibbles per word, base k at bits 4(k%16)
  u32 max_candidates = 100;
  // seed window: 20, or 12 in an --enable-short build of the reference (configure.ac:70-73,
  // src/AbismalIndex.hpp:73-77); stored in the file, and the mapper's read-length rules follow it
  u32 window = kWindow;
  u32 min_read_len() const { return kKeyWeight + window - 1; }  // src/abismal.cpp:212-213
  u64 counter_size = 0, counter_size3 = 0, index_size = 0, index_size3 = 0;
  std::vector<u32> counter, counter_t, counter_a;  // bucket start offsets (+1 end)
  std::vector<u32> index, index_t, index_a;        // genome positions

  void read(const std::string &path);         // src/AbismalIndex.cpp:1082-1146
  void write(const std::string &path) const;  // src/AbismalIndex.cpp:1037-1072
  // src/AbismalIndex.cpp:281-331 (+ everything it calls)
  // targets: `abismal idx -A` file (src/AbismalIndex.cpp:206-279); empty = whole genome
  void build_from_fasta(const std::string &fasta, unsigned n_threads = 1, const std::string &targets = "",
                        u32 window_size = kWindow);
};

}  // namespace abo
#endif
