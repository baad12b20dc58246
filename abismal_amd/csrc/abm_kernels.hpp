// abismal_amd: kernel argument blocks and launchers (host-visible side of
// abm_kernels.hip).
#pragma once
#include "abm_device.hpp"

namespace abm {

struct SeArgs {
  DevIndex ix;
  const u64 *packed;  // [n][4][W]
  const u32 *lens;    // [n]
  const u32 *order;   // [n] processing order (heaviest first), or null
  u64 n_reads;
  u32 W, WB, GW;      // words per packed encoding / 2-letter bit string / genome window
  u32 max_len;        // longest read of the batch
  u32 tb_extra;       // tb_extra_bytes()
  u32 G;              // lanes sharing one candidate window (WaveLds::G)
  int mode;           // ABM_SE_*
  double valid_frac;
  double size_frac;   // valid_frac used to size LDS bands (1.0 when the genome has IUPAC codes)
  Hit *res;           // [n]
  u32 *cig;           // [n][cig_stride]
  u32 cig_stride;
  u32 ctmp_cap;       // LDS scratch for a CIGAR's ops: longest read + 2
  u8 *long_tb;        // long-read launch only: per-wave traceback tables (long_tb_bytes each) ...
  u32 *long_ctmp;     // ... and CIGAR scratch (ctmp_cap rounded up to even, per wave), in global memory
  u64 long_tb_bytes;
  u32 *cig_arena;     // CIGARs longer than a slot (see CigarSink); null = none
  u32 *cig_arena_count;
  u32 cig_arena_cap;
  u32 *cig_n;         // [n]
  u32 *status;        // ABM_STATUS_* bits
  unsigned long long *next_read;  // work counter, zero at launch
  u32 *drained;       // optional host-visible flag, set once every read has been handed to a wave
  u32 *finished;      // optional: waves that have run to their end (zero at launch) ...
  u32 *host_tail;     // ... the last of which writes {arena count, status} here (pinned host memory): a host that takes
                      // its results from pinned buffers then needs no device-to-host copy at all after the kernel
  // optional, for hosts that take results slice by slice while the kernel runs (abm_map_se_batch_sliced): the slice a
  // read belongs to (0xFFFF = none), the reads each slice still waits for (device memory), and one word per slice in
  // pinned host memory that the wave finishing a slice's last read sets -- after a system-scope fence, so the slice's
  // hits, counts, slots and arena entries are in host memory by then
  const u16 *slice_id;
  u32 *slice_left;
  u32 *slice_done;
  // optional: the read's SAM line after QNAME written by the kernel itself -- "\tFLAG\tRNAME\tPOS\t255\tCIGAR\t*\t0\t0\tSEQ\t*\t
  // NM:i:n\tCV:A:c\n" as format_se writes it (src/abismal.cpp:481-545): the wave has the hit, the CIGAR and the conversion,
  // the reads' text is on the device anyway (blob / off: what the batch was packed from), and the chromosome table rides
  // with the index.  sam_len[r]: bytes written into sam_tail[r * sam_stride ..), 0 = the read has no record (unmapped,
  // ambiguous and not allowed, or across a chromosome's end), 0xFFFFFFFF = not written here (a CIGAR beyond its slot or
  // the text beyond the stride: the host formats that one)
  const char *blob;
  const u64 *off;
  char *sam_tail;
  u32 *sam_len;
  u32 sam_stride;
  int sam_allow_ambig;
  u32 *read_cycles;   // optional [n], diagnostic kernel only: per-read shader cycles / 1024
  unsigned long long *work;  // optional [16]: seed_iters, search probes, candidates,
                             // read words compared, set updates, alignments
};

// paired-end launch arguments (abm_kernels_pe.hip)
struct PeArgs {
  DevIndex ix;
  const u64 *packed1, *packed2;  // [n][4][W]
  const u32 *lens1, *lens2;
  const u32 *order;              // tier 1: heaviest-first order or null
  const u32 *subset;             // tier 2: ids of the pairs to redo
  const u32 *subset_count;       // tier 2: how many
  u64 n_pairs;
  u32 W, WB, GW, tb_extra, G, max_len;
  int mode;
  double valid_frac;
  u32 min_frag, max_frag;
  int allow_ambig;
  Hit *pairs;                    // [n] as abm_pair: {i16 score, i16 pad, Hit r1, Hit r2} = 5 x u32
  Hit *se1, *se2;
  u32 *cig1, *cig2;
  u32 cig_stride;
  u32 ctmp_cap;                  // LDS scratch for a CIGAR's ops: longest read + 2
  u32 *cig_arena;                // CIGARs longer than a slot (see CigarSink; both ends share it); null = none
  u32 *cig_arena_count;
  u32 cig_arena_cap;
  u32 *cig_n1, *cig_n2;
  u32 *status;
  unsigned long long *next_read;
  unsigned long long *work;
  u8 *need_big;                  // [n] tier 1 -> tier 2 hand-off
  u32 *payload_ws;               // [grid][cap]
  u32 *list_ws;                  // tier 2: [grid][2][cap] positions, then diffs and scores (i16)
  u32 *heap_ws;                  // tier 2: [grid][cap] candidate heap / sort buffer
  u32 *log_ws;                   // tier 2: [grid][32 + 12 cap] lists kept for a deferred best_single
  u32 cap;
  u32 *pair_diag;                // optional [n], diagnostic kernel only: largest set << 16 | shader cycles >> 20
  u32 *pair_phases;              // optional [n][8], diagnostic kernel only: the pair's shader cycles >> 10 by phase -- probe + narrow,
                                 // window gather + Hamming, replay, sort + unique, pairable-entry scoring, mating + tracebacks,
                                 // best_single, single-end fallback (added to: the seed and mate kernels each fill in theirs)
  // the long-end launch only (pairs with an end of kLdsReadLen + 1 .. kMaxReadLen bases, listed in `subset`, packed by
  // list position): per wave, both ends' encodings and bit strings (pe_long_q_words u64), a traceback table
  // (long_tb_bytes) and CIGAR scratch (ctmp_cap rounded up to even), all in global memory
  u64 *long_q;
  u8 *long_tb;
  u32 *long_ctmp;
  u64 long_tb_bytes;
  // optional (host entry point): waves that have run to their end (zero at launch), the last of which writes
  // {arena count, status} to host_tail (pinned host memory) -- as in SeArgs: results that lie in pinned memory need
  // no device-to-host copy after the kernels
  u32 *finished;
  u32 *host_tail;
  // The phase-split launches (round 5): a SEED kernel runs both seed passes of every orientation call's two ends and
  // hands the finished candidate lists over in global memory; MATE kernels take them from there (sort, scoring,
  // mating, tracebacks, best_single, the single-end fallback).  Per pair and list (orientation call o, end which:
  // slot 2 o + which of 2 n_or) two words {where the list starts in hand_pos / hand_d, entries | worth << 16 |
  // heap_order << 17}; lists are bump-allocated from hand_count (a list that finds no room sends its pair through
  // the whole-pair kernel).  need_big[] is the pair's route: kRouteSmall (every list fits the mate kernel's LDS),
  // kRouteWhole (a set outgrew the seed kernel: the whole-pair kernel with its 32768-entry sets), kRouteBig (lists
  // in global memory: the mate kernel's tier-2 form).
  u32 *hand_hdr;                 // [n][2 n_or][2]
  u32 *hand_pos;                 // [hand_cap]
  i16 *hand_d;                   // [hand_cap]
  unsigned long long *hand_count;
  u32 hand_cap;
  // seed kernel: a list that outgrows its LDS slot (cap entries) moves to this wave's staging area and keeps growing
  // there, up to scap entries (scap <= cap: no staging)
  u32 *stage_pos;                // [grid][scap]
  i16 *stage_d;                  // [grid][scap]
  u32 scap;
  unsigned long long *split_stats;  // optional [4]: pairs by route (small, whole, big), [3] unused
};
constexpr u8 kRouteSmall = 0, kRouteWhole = 1, kRouteBig = 2;

// bytes the traceback table needs beyond the LDS it overlays (genome-window slots 1.. and the
// window cache, both idle while a traceback runs); the kernels carve exactly this much extra
u32 tb_extra_bytes(u32 GW, u32 max_len, double valid_frac);
size_t pe_lds_bytes(u32 W, u32 WB, u32 GW, u32 cig_stride, u32 max_len, double valid_frac, u32 cap, bool big);
int pe_waves_per_simd(size_t lds, bool timed, bool coop);  // which build of the pair kernels a launch with this much LDS per wave takes
int pe_resident_waves(size_t lds, bool big, int wps);
hipError_t launch_map_pe(const PeArgs &a, size_t lds, u32 grid, bool big, bool timed, int wps, hipStream_t st);
hipError_t launch_collect_long_pairs(const u32 *d_lens1, const u32 *d_lens2, u64 n, u32 *d_list, u32 *d_count, hipStream_t st);
size_t pe_long_lds_bytes(u32 GW);
size_t pe_long_q_words(u32 W, u32 WB);
int pe_long_resident_waves(u32 GW);
hipError_t launch_map_pe_long(const PeArgs &a, u32 grid, hipStream_t st);
// the pairs whose route (PeArgs::need_big) is `want`, heaviest weight class first
hipError_t launch_collect_big(const u8 *need_big, const u8 *cls, u64 n, u8 want, u32 *class33, u32 *subset, u32 *count,
                              hipStream_t st);
// the phase-split launches (seed kernel -> hand-over area -> mate kernels; abm_kernels_pe.hip)
size_t pe_seed_lds_bytes(u32 W, u32 WB, u32 max_len, u32 cap);
size_t pe_mate_lds_bytes(u32 W, u32 GW, u32 cig_stride, u32 max_len, double valid_frac, u32 cap, bool big);
int pe_seed_resident_waves(size_t lds, bool coop);
int pe_mate_resident_waves(size_t lds, bool big);
hipError_t launch_pe_seed(const PeArgs &a, size_t lds, u32 grid, bool timed, hipStream_t st);
hipError_t launch_pe_mate(const PeArgs &a, size_t lds, u32 grid, bool big, bool timed, hipStream_t st);
#ifndef ABM_PE_TIER1_CAP
#define ABM_PE_TIER1_CAP 128
#endif
constexpr u32 kPeTier1Cap = ABM_PE_TIER1_CAP;
// (a power of two: sort_unique pads a list to the next power of two inside a buffer of this many entries -- builds with
// 48 and 96 returned wrong pairs, profiles/r03_exp_pe_tier1_cap_small.log; 32 ... 256 measured: 64 and 128 level, 256 slower)
static_assert(kPeTier1Cap >= 32 && (kPeTier1Cap & (kPeTier1Cap - 1)) == 0, "tier-1 list capacity: a power of two, at least the sets' initial 32");

u32 se_window_words(u32 max_len, double valid_frac);
size_t se_lds_bytes(u32 W, u32 WB, u32 cig_stride, u32 max_len, double valid_frac);
// seed-extension tables (abm_ext.hip): keys of table `mode` (0: 2-letter, 1 / 2: 3-letter C->T / G->A) with `extra`
// letters beyond the hashed ones; bytes of scratch a build needs; the build itself (out[ext_keys] entries, n_idx =
// entries of the table's index array)
constexpr u32 kExtGapCap = 1u << 20;
u64 ext_keys(int mode, u32 extra);
size_t ext_scratch_bytes(int mode, u32 extra);
hipError_t build_ext_table(const DevIndex &ix, int mode, u32 extra, u32 maxc, u64 n_idx, uint2 *out, void *scratch, u32 *d_fail,
                           hipStream_t st);
// window records (DevIndex::wrec): the longest read records of `blocks` blocks serve (2 L - key weight <= 64 blocks: the
// first seed offset's window must end, and the last one's begin, inside the record), the blocks a read length needs, the
// bytes of the table, and the build (records of index, index_t, index_a one after the other; n_idx = their entries)
u32 window_record_max_len(u32 blocks);
u32 window_record_blocks_for(u32 max_len);
size_t window_record_bytes(u64 n_entries, u32 blocks);
hipError_t build_window_records(const DevIndex &ix, u64 n_plane_blocks, u64 n_bases, const u64 n_idx[3], u32 blocks, u64 *out, hipStream_t st);
// bit-plane copies of the genome for the Hamming filter (DevIndex::planes): n_blocks blocks of 64 bases each,
// from the first n_words words of nibbles; blank nibbles (N) mark their surroundings in nmap (DevIndex::nmap,
// zeroed by the caller); *bad is set if a nibble below n_bases has two or more bits
hipError_t launch_make_planes(const u64 *d_genome, u64 n_words, u64 n_bases, u64 n_blocks, u64 *d_planes0,
                              u64 *d_planes1, u32 *d_nmap, u32 *d_bad, hipStream_t st);
hipError_t launch_pack_reads(const char *d_blob, const u64 *d_off, u64 n, u32 W, u64 *d_packed,
                             u32 *d_lens, hipStream_t st);
hipError_t launch_order_reads(const DevIndex &ix, const u64 *d_packed, const u32 *d_lens, u64 n, u32 W,
                              int mode, u8 *d_cls, u32 *d_class33, u32 *d_order, hipStream_t st);
// the same for a batch whose results leave slice by slice: the few heaviest reads first (as many classes from the top
// as hold at most 1/32 of the batch), then the reads before the first slice, then slice after slice.  slice_first:
// [n_slices + 1] read indices in device memory; slice_id: [n] out; hist: [(n_slices + 1) * 33 + 33 + n_slices + 2] scratch
hipError_t launch_order_reads_sliced(const DevIndex &ix, const u64 *d_packed, const u32 *d_lens, u64 n, u32 W, int mode,
                                     u8 *d_cls, const u32 *d_slice_first, u32 n_slices, u16 *d_slice_id, u32 *d_hist,
                                     u32 *d_slice_left, u32 *d_order, hipStream_t st);
inline size_t order_sliced_hist_words(u32 n_slices) { return static_cast<size_t>(n_slices + 1) * 33 + 33 + n_slices + 2; }
hipError_t launch_compact_cigars(const Hit *d_res, const u32 *d_cig, const u32 *d_cig_n, u64 n, u32 stride,
                                 unsigned long long *d_off, u32 *d_blob, void *tmp, size_t *tmp_bytes, hipStream_t st);
hipError_t launch_gather_cigars(const u32 *d_cig, u32 stride, const unsigned long long *d_off, u64 n, u32 *d_blob,
                                hipStream_t st);
// n_waves = one-wave workgroups of the (persistent) grid
hipError_t launch_map_se(SeArgs a, u32 max_len, u32 n_waves, bool timed, hipStream_t st);
// the long-read launch (reads of kLdsReadLen + 1 .. kMaxReadLen bases, listed in a.order, packed by list position)
size_t se_long_lds_bytes(u32 W, u32 WB, u32 GW);
size_t se_long_tb_bytes(u32 max_len);
int se_long_resident_waves(u32 W, u32 WB, u32 GW);
hipError_t launch_collect_long(const u32 *d_lens, u64 n, u32 *d_list, u32 *d_count, hipStream_t st);
hipError_t launch_pack_listed(const char *d_blob, const u64 *d_off, const u32 *d_list, u64 m, u32 W, u64 *d_packed, hipStream_t st);
hipError_t launch_map_se_long(SeArgs a, u32 n_waves, hipStream_t st);
int se_resident_waves(u32 W, u32 WB, u32 cig_stride, u32 max_len, double valid_frac);

}  // namespace abm
