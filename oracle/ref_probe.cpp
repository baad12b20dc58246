// ORACLE/_ref — test infrastructure only.
// A thin C driver around the parts of the REFERENCE that are header-only and
// therefore compile from the sources where they lie under /root/reference/src
// (no stand-ins for the un-vendored smithlab_cpp / bamxx / htslib are written:
// abismal.cpp, AbismalIndex.cpp and simreads.cpp are NOT compiled).
// Built by oracle/Makefile into oracle/_ref/libref_probe.so; used by
// tests/test_oracle_vs_reference_headers.py to check the restatement against the
// real AbismalAlign, the real encoding tables and the real hash functions.
#include "AbismalAlign.hpp"   // reference header (pulls abismal_cigar_utils.hpp, dna_four_bit_bisulfite.hpp)
#include "AbismalIndex.hpp"   // reference header: get_bit, get_three_letter_num, hashes

#include <cstdint>
#include <vector>

using RefAligner = AbismalAlign<simple_aln::mismatch_score, simple_aln::indel>;

extern "C" {

int ref_read_code(int c, int a_rich) { return a_rich ? encode_base_a_rich[c & 127] : encode_base_t_rich[c & 127]; }
int ref_genome_code(int c) { return dna_four_bit_encoding[c & 127]; }
int ref_get_bit(int nt) { return get_bit(static_cast<std::uint8_t>(nt)); }
int ref_trit(int nt, int g_to_a_conv) {
  return g_to_a_conv ? get_three_letter_num<g_to_a>(static_cast<std::uint8_t>(nt))
                     : get_three_letter_num<c_to_t>(static_cast<std::uint8_t>(nt));
}
uint32_t ref_hash2(const uint8_t *nib) { std::uint32_t k = 0; get_1bit_hash(nib, k); return k; }
uint32_t ref_hash3(const uint8_t *nib, int g_to_a_conv) {
  std::uint32_t k = 0;
  if (g_to_a_conv) get_base_3_hash<g_to_a>(nib, k); else get_base_3_hash<c_to_t>(nib, k);
  return k;
}
uint32_t ref_roll2(uint32_t k, int nt) { shift_hash_key(static_cast<std::uint8_t>(nt), k); return k; }
uint32_t ref_roll3(uint32_t k, int nt, int g_to_a_conv) {
  if (g_to_a_conv) shift_three_key<g_to_a>(static_cast<std::uint8_t>(nt), k);
  else shift_three_key<c_to_t>(static_cast<std::uint8_t>(nt), k);
  return k;
}

// AbismalAlign::align<tb> (+ build_cigar_len_and_pos and simple_aln::edit_distance when tb)
int ref_align(const uint64_t *genome, uint64_t n_words, const uint8_t *q, uint32_t qlen, int diffs,
              int max_diffs, uint32_t t_pos, int do_tb, uint32_t *cig_out, uint32_t cig_cap,
              uint32_t *n_cig, uint32_t *aln_len, uint32_t *new_pos, int *nm) {
  const std::vector<std::size_t> g(genome, genome + n_words);
  const genome_four_bit_itr gi(std::cbegin(g));
  RefAligner aln(gi);
  aln.reset(qlen);
  const std::vector<std::uint8_t> query(q, q + qlen);
  score_t scr;
  if (!do_tb)
    return aln.align<false>(static_cast<score_t>(diffs), static_cast<score_t>(max_diffs), query, t_pos);
  scr = aln.align<true>(static_cast<score_t>(diffs), static_cast<score_t>(max_diffs), query, t_pos);
  bam_cigar_t cigar;
  std::uint32_t len = 0, pos = t_pos;
  aln.build_cigar_len_and_pos(static_cast<score_t>(diffs), static_cast<score_t>(max_diffs), cigar, len, pos);
  *n_cig = static_cast<uint32_t>(cigar.size());
  for (std::size_t i = 0; i < cigar.size() && i < cig_cap; ++i) cig_out[i] = cigar[i];
  *aln_len = len;
  *new_pos = pos;
  *nm = simple_aln::edit_distance(scr, len, cigar);
  return scr;
}

}  // extern "C"
