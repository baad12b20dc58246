// abismal_amd host side: AbismalIndex builder + writer.
#pragma once
#include "abm_index_file.hpp"

namespace abm {
// FASTA -> padded text + chromosome table (load_genome, src/AbismalIndex.cpp:1322-1360)
void load_fasta(const std::string &path, std::string &text, std::vector<std::string> &names,
                std::vector<uint32_t> &starts);
// `abismal idx -A targets`: blank everything outside the listed regions (src/AbismalIndex.cpp:83-123, :206-241)
void mask_outside_targets(const std::string &targets_path, std::string &text, const std::vector<std::string> &names,
                          const std::vector<uint32_t> &starts);
// consumes `text` (freed early to bound peak memory)
void build_index(std::string &text, const std::vector<std::string> &names,
                 const std::vector<uint32_t> &starts, unsigned n_threads, HostIndex &out, uint32_t window = 20);
// AbismalIndex::write, src/AbismalIndex.cpp:1037-1072
void write_index(const HostIndex &h, const std::string &path);
}  // namespace abm
