// ORACLE — test infrastructure only.  Not part of the shipped product.
//
// CPU restatement of the abismal (v3.3.0) read-mapping algorithm, written
// from the reference's behaviour, used solely as the parity checker for the
// HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
// Pinned against the 16 md5 goldens of /root/reference/data/md5sum.txt
// (see oracle/README.md and tests/test_oracle_goldens.py).
//
// This header: scalar primitives shared by the index builder, simulator and
// mapper restatements.  Each function cites the reference lines it follows.
#ifndef ABO_COMMON_HPP
#define ABO_COMMON_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace abo {

using u8 = std::uint8_t;
using u16 = std::uint16_t;
using u32 = std::uint32_t;
using u64 = std::uint64_t;
using i16 = std::int16_t;
using i32 = std::int32_t;

// seed constants: src/AbismalIndex.hpp:66-93
constexpr u32 kKeyWeight = 25;       // letters in the 2-letter hash
constexpr u32 kKeyWeight3 = 16;      // letters in the 3-letter hash
constexpr u32 kWindow = 20;          // minimiser-like selection window
constexpr u32 kSortDepth = 256;      // letters used to sort a bucket
constexpr u32 kHashMask2 = (1u << kKeyWeight) - 1;
constexpr u32 kHashMod3 = 43046721u; // 3^16
constexpr u32 kPadding = 32767;      // Ns added at both genome ends
constexpr u32 kMinReadLen = kKeyWeight + kWindow - 1;  // 44, src/abismal.cpp:212
constexpr u32 kMaxNRun = 256;        // src/AbismalIndex.hpp:249

// flag bits carried in a hit: src/common.hpp:112-126, src/abismal.cpp:81-84
constexpr u16 kFlagRC = 0x10;
constexpr u16 kFlagAmbig = 0x100;
constexpr u16 kFlagARich = 0x1000;

// three-letter alphabets: src/AbismalIndex.hpp:156-159
enum Conv : int { C_TO_T = 0, G_TO_A = 1 };

// read nibble: src/dna_four_bit_bisulfite.hpp:26-57 (only A,C,G,T map to
// non-zero; T-rich T=1010 also matches C, A-rich A=0101 also matches G)
inline u8 read_nibble(char c, bool a_rich) {
  switch (c) {
  case 'A': case 'a': return a_rich ? 5 : 1;
  case 'C': case 'c': return 2;
  case 'G': case 'g': return 4;
  case 'T': case 't': return a_rich ? 8 : 10;
  default: return 0;
  }
}

// genome nibble: src/dna_four_bit_bisulfite.hpp:156-165 (IUPAC sets).  NB the
// table as compiled maps 'N' (and every non-IUPAC byte) to 0 -- "matches
// nothing" -- not to 15 as that header's comment block suggests.
inline u8 genome_nibble(unsigned char c) {
  switch (c & 0xDF) {  // fold case for letters
  case 'A': return 1;  case 'B': return 14; case 'C': return 2;
  case 'D': return 13; case 'G': return 4;  case 'H': return 11;
  case 'K': return 12; case 'M': return 3;
  case 'R': return 5;  case 'S': return 6;  case 'T': return 8;
  case 'V': return 7;  case 'W': return 9;  case 'Y': return 10;
  default: return 0;
  }
}

// nibble k of the packed genome: src/dna_four_bit_bisulfite.hpp:191-262
inline u8 gnib(const u64 *g, u64 k) {
  return static_cast<u8>((g[k >> 4] >> ((k & 15u) << 2)) & 15u);
}

// 2-letter symbol: src/AbismalIndex.hpp:255-258
inline u32 bit2(u8 nt) { return (nt & 5) == 0; }

// 3-letter digit for hashing: src/AbismalIndex.hpp:260-269
inline u32 trit(u8 nt, Conv c) {
  return c == C_TO_T ? ((((nt & 4) != 0) << 1) | ((nt & 1) != 0))
                     : ((((nt & 8) != 0) << 1) | ((nt & 2) != 0));
}

// 3-letter symbol as used for bucket ordering / narrowing (NOT 0,1,2):
// src/abismal.cpp:1196-1203, src/AbismalIndex.cpp:877-883
inline u32 sortsym3(u8 nt, Conv c) { return c == C_TO_T ? (nt & 5) : (nt & 10); }

// rolling updates: src/AbismalIndex.hpp:271-281
inline void roll2(u8 nt, u32 &k) { k = ((k << 1) | bit2(nt)) & kHashMask2; }
inline void roll3(u8 nt, Conv c, u32 &k) { k = (k * 3 + trit(nt, c)) % kHashMod3; }

// the reference's deterministic base generator (process-global LCG):
// src/AbismalIndex.hpp:39-64
struct BaseLCG {
  u32 x = 1;
  char next() {
    x = (1103515245u * x + 12345u) & 0x7fffffffu;
    return "ACGT"[x & 3];
  }
};

// reverse complement of a read as the mapper does it
// (src/common.hpp:28-44: table indexed by c-'A', everything but ACGT -> N)
inline std::string revcomp_read(const std::string &s) {
  std::string t(s.size(), 'N');
  for (std::size_t i = 0; i < s.size(); ++i) {
    char c = s[s.size() - 1 - i], o = 'N';
    if (c == 'A') o = 'T'; else if (c == 'C') o = 'G';
    else if (c == 'G') o = 'C'; else if (c == 'T') o = 'A';
    t[i] = o;
  }
  return t;
}

struct ChromTable {
  std::vector<std::string> names;  // includes pad_start / pad_end
  std::vector<u32> starts;         // names.size()+1 entries
  u32 genome_size() const { return starts.back(); }
  // src/AbismalIndex.cpp:1283-1294
  void locate(u32 pos, i32 &chrom, u32 &off) const;
  // src/AbismalIndex.cpp:1305-1320
  bool locate(u32 pos, u32 reflen, i32 &chrom, u32 &off) const;
};

// FASTA -> padded text genome: src/AbismalIndex.cpp:1322-1360
void load_fasta_padded(const std::string &path, std::string &genome, ChromTable &ct);


}  // namespace abo
#endif
