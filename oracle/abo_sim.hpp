// ORACLE — test infrastructure only (see abo_common.hpp).
#ifndef ABO_SIM_HPP
#define ABO_SIM_HPP
#include "abo_common.hpp"
#include <limits>

namespace abo {
// option defaults follow src/simreads.cpp:444-473
struct SimParams {
  std::string fasta, out_prefix;
  bool single_end = false, pbat = false, random_pbat = false;
  std::size_t read_len = 100, min_frag = 100, max_frag = 250, n_reads = 100;
  std::size_t seed = 1;
  double mut_rate = 0.0, sub_rate = 1.0, ins_rate = 1.0, del_rate = 1.0, bs_conv = 1.0;
  char strand = 'b';
};
void simulate_reads(const SimParams &p);
}  // namespace abo
#endif
