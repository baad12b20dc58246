// ORACLE — test infrastructure only (see abo_common.hpp).
// Restatement of abismal's per-read mapping path: seed lookup -> candidate
// filter -> candidate sets -> banded alignment -> SE selection / PE mating.
#ifndef ABO_MAP_HPP
#define ABO_MAP_HPP

#include "abo_index.hpp"

#include <array>

namespace abo {

// se_element: src/abismal.cpp:224-297
struct Hit {
  i16 diffs = 32767;
  u16 flags = 0;
  u32 pos = 0;
  bool rc() const { return flags & kFlagRC; }
  bool a_rich() const { return flags & kFlagARich; }
  bool ambig() const { return flags & kFlagAmbig; }
  bool empty() const { return pos == 0; }
  void clear() { pos = 0; diffs = 32767; }                 // reset()
  void clear(u32 readlen) { pos = 0; diffs = static_cast<i16>(0.4 * readlen); }
  bool same_site(const Hit &o) const { return pos == o.pos && flags == o.flags; }
};

// pe_element: src/abismal.cpp:547-619
struct PairHit {
  i16 aln_score = 0, max_aln_score = 0;
  Hit r1, r2;
  i16 diffs() const { return static_cast<i16>(r1.diffs + r2.diffs); }
  bool empty() const { return r1.empty(); }
  bool ambig() const { return r1.ambig(); }
  bool sure_ambig() const { return ambig() && aln_score == max_aln_score; }
  bool should_report(bool allow_ambig) const { return !empty() && (allow_ambig || !ambig()); }
  void clear() { aln_score = 0; r1.clear(); r2.clear(); }
  void clear(u32 l1, u32 l2) {
    aln_score = 0; r1.clear(l1); r2.clear(l2);
    max_aln_score = static_cast<i16>(static_cast<i16>(2 * l1) + static_cast<i16>(2 * l2));
  }
  bool offer(i16 scr, const Hit &s1, const Hit &s2);
};

enum SeMode : int { SE_T_RICH = 0, SE_A_RICH = 1, SE_RANDOM = 2 };
enum PeMode : int { PE_NORMAL = 0, PE_PBAT = 1, PE_RANDOM = 2 };

struct MapParams {
  u32 max_candidates = 100;
  double valid_frac = 0.1;
  u32 min_frag = 32, max_frag = 3000;
  bool allow_ambig = false;  // only steers the PE single-end fallback (src/abismal.cpp:1991)
};

// per-read work tallies feeding the algorithmic-bytes model
struct Work {
  u64 seed_iters = 0, search_probes = 0, candidates = 0, words = 0,
      set_updates = 0, aligns = 0, aligns_tb = 0, dp_cells = 0, reads = 0;
};

using Cigar = std::vector<u32>;
u32 cigar_ref_len(const Cigar &c);  // src/abismal.cpp:451-462

// the restatement's aligner, exposed for the cross-check against the reference's
// header-only AbismalAlign (oracle/_ref): same arguments as ref_align in ref_probe.cpp
int probe_align(const u64 *genome, const u8 *q, u32 qlen, i16 diffs, i16 max_diffs, u32 t_pos, bool tb,
                Cigar *cig, u32 *aln_len, u32 *new_pos, int *nm);

class Mapper {
public:
  Mapper(const Index &ix, const MapParams &p);
  ~Mapper();
  // body of the loops at src/abismal.cpp:1552-1581 / :1645-1685, up to (not
  // including) format_se.  `read` is already trimmed; empty = skipped.
  void map_se(const std::string &read, SeMode mode, Hit &best, Cigar &cig);
  // body of src/abismal.cpp:1950-1999 / :2094-2155, up to select_output.
  void map_pe(const std::string &r1, const std::string &r2, PeMode mode, PairHit &best,
              Hit &se1, Hit &se2, Cigar &cig1, Cigar &cig2);
  // Only the side effect mapping a read has on LATER reads: its encodings overwrite the reused
  // buffers (src/abismal.cpp:1377-1386), which reads of 44-46 bases hash past their end
  // (SURVEY A.11).  A worker that starts in the middle of the input replays this for the reads
  // before its share, so that sharded runs equal the reference at -t 1.
  void touch_se(const std::string &read, SeMode mode);
  void touch_pe(const std::string &r1, const std::string &r2, PeMode mode);
  Work work;

private:
  struct Impl;
  Impl *m;
};

}  // namespace abo
#endif
