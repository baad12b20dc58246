/* abismal_amd — C ABI of the MI355X-native mapping path.
 *
 * The reference (smithlabcode/abismal v3.3.0) has no FFI; its seam for this
 * path is the body of the per-read loops in src/abismal.cpp — everything
 * between ReadLoader::load_reads (:1545) and format_se / select_output
 * (:1577, :2000).  Each entry point below names the reference code it
 * replaces.  All functions return 0 on success, <0 on error (text via
 * abm_last_error()); no ownership is transferred; plain pointers and sizes
 * only.
 */
#ifndef ABISMAL_AMD_H
#define ABISMAL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct abm_index abm_index; /* host copy of an AbismalIndex file */
typedef struct abm_ctx abm_ctx;     /* one GPU: index in HBM + workspaces */

/* se_element, src/abismal.cpp:224-297.  pos is the absolute position in the
 * padded concatenated genome; pos == 0 means "no hit" (flags/diffs are then
 * unspecified).  flags: 0x10 reverse strand, 0x100 ambiguous, 0x1000 A-rich. */
typedef struct { int16_t diffs; uint16_t flags; uint32_t pos; } abm_hit;

/* pe_element, src/abismal.cpp:547-619 (aln_score + both ends) */
typedef struct { int16_t aln_score; int16_t reserved; abm_hit r1, r2; } abm_pair;

/* the tunables the path reads: src/abismal.cpp:2329-2337, :2448-2452 */
typedef struct {
  uint32_t max_candidates; /* 0 = value stored in the index (-c) */
  double valid_frac;       /* -m, default 0.1 */
  uint32_t min_frag;       /* -l, default 32 (PE) */
  uint32_t max_frag;       /* -L, default 3000 (PE) */
  int32_t allow_ambig;     /* -a: only steers the PE single-end fallback */
} abm_params;

/* conversion modes = which per-read body runs */
enum { ABM_SE_T_RICH = 0, ABM_SE_A_RICH = 1, ABM_SE_RANDOM = 2 }; /* :1552-1581, -A/-P, :1645-1685 */
enum { ABM_PE_NORMAL = 0, ABM_PE_PBAT = 1, ABM_PE_RANDOM = 2 };   /* :1950-2002, -P, :2094-2158 */

const char *abm_last_error(void);
int abm_device_count(void); /* HIP devices visible to the process (0 = none: every abm_ctx_create will fail) */
/* NUMA node of the host memory nearest to `device` (from its PCI address; -1 = unknown): where a host driver places
 * the threads and buffers that feed that GPU (no reference counterpart: its worker threads are not placed). */
int abm_device_numa_node(int device);
/* Page-locked host memory, usable from every device (hipHostMalloc, portable): batches whose seq_blob / seq_off lie
 * in it are uploaded by the DMA engines at the link's rate instead of through the runtime's staging buffers (the
 * reference has no counterpart: its batches never leave the host). */
int abm_host_alloc(size_t bytes, void **out);
void abm_host_free(void *p);
void abm_default_params(abm_params *p);

/* AbismalIndex::read, src/AbismalIndex.cpp:1082-1146 (+ seed::read :988-1024,
 * ChromLookup::read :1225-1258).  Rejects files whose identifier or seed
 * constants (25 / 20 / 256) differ, with the reference's messages. */
int abm_index_open(const char *path, abm_index **out);
void abm_index_close(abm_index *ix);
uint32_t abm_index_max_candidates(const abm_index *ix);
uint32_t abm_index_n_chroms(const abm_index *ix);              /* incl. pad_start/pad_end */
const char *abm_index_chrom_name(const abm_index *ix, uint32_t i);
const uint32_t *abm_index_chrom_starts(const abm_index *ix);    /* n_chroms+1 entries */
uint64_t abm_index_bytes(const abm_index *ix);                  /* bytes resident in HBM after upload */

/* `abismal idx <genome.fa> <out.idx>` (src/abismalidx.cpp:35-115 ->
 * AbismalIndex::create_index + write, src/AbismalIndex.cpp:281-331, :1037-1072).
 * Host-side, multi-threaded; the file is byte-identical to the reference's. */
int abm_index_build(const char *fasta_path, const char *out_path, uint32_t n_threads);
/* `abismal idx -A <targets> ...` (AbismalIndex::create_index(targets, genome), src/AbismalIndex.cpp:206-279):
 * only the regions listed in targets_path ("chrom start end" per line) are indexed; NULL or "" = whole genome. */
int abm_index_build_targets(const char *fasta_path, const char *targets_path, const char *out_path, uint32_t n_threads);
/* The same with the seed window chosen: 20 (default) or 12, the reference's --enable-short build for short reads
 * (configure.ac:70-73, src/AbismalIndex.hpp:73-77: shortest mappable read 36 instead of 44 bases).  The window is
 * stored in the index file; abm_index_open takes either and the mapping entry points follow it. */
int abm_index_build_opts(const char *fasta_path, const char *targets_path, uint32_t window, const char *out_path,
                         uint32_t n_threads);
uint32_t abm_index_window(const abm_index *ix);

/* Seed-extension tables (no reference counterpart; results are unaffected).  The first context on a device also
 * derives, in HBM, three tables that answer the first steps of the bucket-narrowing loops of find_candidates /
 * find_candidates_three (src/abismal.cpp:1163-1259) with one load per seed offset: one entry per key of the hashed
 * letters plus letters2 more (2-letter table, at most 7) / letters3 more (3-letter tables, at most 4).  By default
 * the letters are chosen from the genome's length (none below 33 Mbp; 7 and 4 = 90 GB at hg38 scale, fewer if that
 * exceeds half of the free device memory); this call fixes them (0, 0 = no tables) and must precede the first
 * abm_ctx_create on the index.  The tables are built for the index's max_candidates (or the value given to
 * abm_index_set_max_candidates before the context was created); a call with another value runs without them --
 * nothing is ever rebuilt inside a mapping call.  abm_ctx_rebuild_seed_extension rebuilds them explicitly. */
int abm_index_set_seed_extension(abm_index *ix, int letters2, int letters3);
/* The automatic choice (letters from the index's size) capped at these (0, 0 = no tables).  At hg38 scale 7 + 4 letters are
 * 90 GB, 6 + 3 are 36 GB, 4 + 2 are 10 GB; since the kernels narrow every range beyond max_candidates directly the tables
 * are worth 3.7 / 3.1 / 2 % of the single-end kernel's time and nothing to the pair kernels
 * (profiles/r05_exp_e2e_and_tables.log) -- so a host that maps pairs caps at 0 and 0.  Contexts created afterwards. */
int abm_index_set_seed_extension_cap(abm_index *ix, int letters2, int letters3);
/* Window records (no reference counterpart; results are unaffected).  check_hits (src/abismal.cpp:1124-1150) compares
 * the read with the genome at every entry of a checked bucket: one random gather per candidate, and at hg38 scale 2,800
 * of them per 100-base read -- the line requests that bound the mapping kernels.  With 288 GB of HBM the index can carry
 * its candidates' windows itself: for every index entry the stretch of the genome (as bit planes, 2 bits per base) that
 * a window of a read of up to max_read_len bases can lie in, entries in index order, so that a bucket's windows are
 * consecutive 48-byte records (reads up to 108 bases; 64 bytes up to 140, 80 up to 172) instead of one 128-byte line
 * each.  8.4 GB at hg38 scale for 100-base reads.  Reads longer than the records serve are filtered from the bit planes
 * as before.  0 = none (the default); must precede the first abm_ctx_create on the index.  abm_ctx_window_records: the
 * longest read the context's records serve (0: none were built -- no planes, or not enough free device memory). */
int abm_index_set_window_records(abm_index *ix, int max_read_len);
uint32_t abm_ctx_window_records(const abm_ctx *ctx);
/* the max_candidates (-c, src/abismal.cpp:2329) the calls on this index will pass, when it is not the value stored
 * in the index file: the tables of contexts created afterwards are built for it (0 = the file's value) */
int abm_index_set_max_candidates(abm_index *ix, uint32_t max_candidates);
/* The mapping kernels narrow a seed's big ranges -- seeds inside high-copy repeats -- directly: find_candidates /
 * find_candidates_three (src/abismal.cpp:1163-1259) extend such a seed letter by letter, a bisection of the range per
 * letter; the same final range comes out of one bisection with whole-string comparisons plus ~30 probes (buckets are
 * sorted by those letters, src/AbismalIndex.cpp:857-978).  Results are unaffected.  min_entries: the smallest range
 * taken that way (default 64: every range beyond the default max_candidates; 0 = never); takes effect with the next call on any context of the index.  (Paired-end
 * calls since round 3, single-end calls since round 5.) */
int abm_index_set_direct_narrowing(abm_index *ix, uint32_t min_entries);
/* Rebuilds the tables of the context's device for another max_candidates (0 = the index file's).  Waits for the device,
 * frees and allocates the tables' memory and runs for seconds at hg38 scale: a set-up call, allowed only while the
 * context is the only one on its device. */
int abm_ctx_rebuild_seed_extension(abm_ctx *ctx, uint32_t max_candidates);
/* what the context's device holds: letters per table (0 = no tables) and their bytes */
int abm_ctx_seed_extension(const abm_ctx *ctx, uint32_t *letters2, uint32_t *letters3, uint64_t *bytes);

/* A context = one host thread's workspaces and stream on `device` (hipSetDevice
 * ordinal).  The first context on a device replicates the index into its HBM;
 * later ones share that replica (freed with the last of them).  Calls on one
 * context are serialised; use one context per mapper thread so that a batch's
 * transfers overlap another batch's kernels.  Destroy every context before
 * abm_index_close. */
int abm_ctx_create(const abm_index *ix, int device, abm_ctx **out);
/* Optional set-up: sizes the context's workspaces for host-buffer batches of up to n reads (pairs, if paired != 0)
 * of up to max_len bases and runs a few dummy reads through the kernels, so that the first real batch pays neither
 * allocations nor code loading (the reference's counterpart: the per-thread scratch built before the batch loop,
 * src/abismal.cpp:1535-1539, :1919-1929). */
int abm_ctx_reserve(abm_ctx *ctx, uint64_t n, uint32_t max_len, int paired);
void abm_ctx_destroy(abm_ctx *ctx);
/* Device memory: free and total bytes of `device` right now (hipMemGetInfo), and what one more context's workspaces for
 * paired-end batches of n pairs of up to max_len bases will take once reserved -- for a host that has to decide how many
 * mapper contexts fit beside the index (no reference counterpart: the reference's per-thread scratch is host memory). */
int abm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);
int abm_ctx_pe_footprint(abm_ctx *ctx, uint64_t n, uint32_t max_len, uint64_t *bytes);

/* Single-end batch, host buffers.  Replaces the loop body of
 * map_single_ended / map_single_ended_rand (src/abismal.cpp:1552-1576,
 * :1645-1680): reads are the strings ReadLoader would hand over (already
 * N-trimmed; empty = skipped), concatenated in seq_blob with n+1 offsets.
 * out_res[i] equals bests[i] and the CIGAR equals r[i].cig just before
 * format_se; CIGAR ops are BAM-encoded (len<<4|op) in out_cig_blob with n+1
 * offsets in out_cig_off (reads without a hit get an empty CIGAR).
 * cig_capacity is out_cig_blob's size in ops: a few ops per read are typical,
 * at most read length + 2; ABM_ERR_CAPACITY (-2) is returned if it does not
 * suffice, and the call can be repeated with a larger buffer.
 * The caller's buffers may be ordinary (pageable) memory: the kernel writes its
 * results into pinned staging buffers of the context (grow-only, see
 * abm_ctx_reserve) and the call copies them out on a few host threads, so
 * nothing has to cross the bus after the kernel while another context's batch
 * occupies the device. */
#define ABM_ERR_CAPACITY (-2)
int abm_map_se_batch(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                     const char *seq_blob, const uint64_t *seq_off, abm_hit *out_res,
                     uint32_t *out_cig_blob, uint64_t cig_capacity, uint64_t *out_cig_off);

/* The same batch with its results handed over SLICE BY SLICE while the kernel is still running -- for a caller that
 * formats and writes output as it arrives (the reference's counterpart: its output loop runs per batch after the
 * mapping loop, src/abismal.cpp:1578-1598; a GPU batch is millions of reads, and waiting for all of it leaves the
 * host idle for the length of a kernel and then the device idle for the length of the formatting).
 * slice_first: n_slices + 1 read indices, non-decreasing, the last one n; reads before slice_first[0] belong to no
 * slice (a batch's lead-in, mapped for its effect on later reads only).  The kernel works through the batch's few
 * heaviest reads first and then slice after slice; when a slice's last read is done, `done(user, slice)` is called
 * on the calling thread, and INSIDE that call abm_ctx_slice_results copies the slice's reads [lo, hi) out of the
 * context's pinned buffers: hits, and CIGARs as a compact blob with hi - lo + 1 offsets relative to the range
 * (ABM_ERR_CAPACITY if cig_capacity ops do not suffice; out_cig_off[hi - lo] then holds the number needed).  After
 * `done` returns the slice's results may be overwritten.  Every slice is handed over exactly once before the call
 * returns; slices of a batch with reads of more than 1024 bases all arrive at the end. */
typedef void (*abm_slice_done_fn)(void *user, uint32_t slice);
int abm_map_se_batch_sliced(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                            const char *seq_blob, const uint64_t *seq_off, uint32_t n_slices,
                            const uint64_t *slice_first, abm_slice_done_fn done, void *user);
int abm_ctx_slice_results(abm_ctx *ctx, uint64_t lo, uint64_t hi, abm_hit *out_res,
                          uint32_t *out_cig_blob, uint64_t cig_capacity, uint64_t *out_cig_off);

/* SAM text from the device (round 5; the reference's counterpart: format_se, src/abismal.cpp:481-545, run per read on the
 * host after the mapping loop).  When enabled, the single-end kernel of abm_map_se_batch_sliced also writes, for every
 * read, its SAM line after QNAME -- "\tFLAG\tRNAME\tPOS\t255\tCIGAR\t*\t0\t0\tSEQ\t*\tNM:i:n\tCV:A:c\n", byte for byte what
 * format_se + htslib's SAM writer print: the wave that mapped the read holds hit, CIGAR and conversion type, the reads'
 * text is on the device, and the chromosome table rides with the index -- so that the host's formatting of a record is
 * two copies (name, tail).  allow_ambig: the run's -a (ambiguous hits are written, flag 0x100).
 * Inside the slice callback, abm_ctx_slice_sam_tails hands out the context's pinned buffers for reads [lo, hi): tails at
 * *tails + k * *stride, lens[k] bytes each -- 0: the read has no record (unmapped, ambiguous and not allowed, or a hit
 * that runs across its chromosome's end), 0xFFFFFFFF: not written by the device (a CIGAR beyond its 4-op slot, a read of
 * more than 1024 bases, a line beyond the slot): format that one from abm_ctx_slice_results as before.  *tails is NULL
 * when the launch wrote none.  Valid until the callback returns. */
int abm_ctx_set_sam_tails(abm_ctx *ctx, int enable, int allow_ambig);
int abm_ctx_slice_sam_tails(abm_ctx *ctx, uint64_t lo, uint64_t hi, const char **tails, uint32_t *stride, const uint32_t **lens);

/* Same computation with every buffer already resident in HBM (d_* are device
 * pointers), enqueued on `stream` (a hipStream_t; NULL = default stream) and
 * not synchronised.  CIGARs land in fixed slots of cig_stride ops per read,
 * their op counts in d_cig_n.  A CIGAR with more ops than cig_stride (its count
 * says so) lies whole in the context's arena instead, beginning at the index its
 * slot's first word holds; abm_ctx_long_cigars() returns the arena of the last
 * call.  ABM_STATUS_CIGAR_OVERFLOW is only set if the arena itself ran out (one op
 * per read by default), in which case such slots hold what fitted.  d_status (one
 * uint32) is OR-ed with ABM_STATUS_* bits. max_len = longest read in the batch.
 * A context owns ONE set of workspaces: a device call first makes `stream` wait
 * for the context's previous device call (on whatever stream that ran), so calls
 * on one context execute one after the other; use one context per stream to
 * overlap batches.  Growing a workspace frees the old one, which synchronises
 * the device once -- size-stable batches never pay that. */
int abm_map_se_device(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                      const char *d_seq_blob, const uint64_t *d_seq_off, uint32_t max_len,
                      abm_hit *d_res, uint32_t *d_cig, uint32_t cig_stride, uint32_t *d_cig_n,
                      uint32_t *d_status, void *stream);

/* Paired-end batch.  Replaces the loop body of map_paired_ended /
 * map_paired_ended_rand (src/abismal.cpp:1950-1999, :2094-2155): out_pair[i],
 * out_se1[i], out_se2[i] and the two CIGARs equal bests[i], bests_se1[i],
 * bests_se2[i], r1[i].cig, r2[i].cig just before select_output. */
int abm_map_pe_batch(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                     const char *seq_blob1, const uint64_t *seq_off1, const char *seq_blob2,
                     const uint64_t *seq_off2, abm_pair *out_pair, abm_hit *out_se1,
                     abm_hit *out_se2, uint32_t *out_cig_blob1, uint64_t *out_cig_off1,
                     uint32_t *out_cig_blob2, uint64_t *out_cig_off2, uint64_t cig_capacity);

int abm_map_pe_device(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                      const char *d_seq_blob1, const uint64_t *d_seq_off1, const char *d_seq_blob2,
                      const uint64_t *d_seq_off2, uint32_t max_len, abm_pair *d_pair,
                      abm_hit *d_se1, abm_hit *d_se2, uint32_t *d_cig1, uint32_t *d_cig2,
                      uint32_t cig_stride, uint32_t *d_cig_n1, uint32_t *d_cig_n2,
                      uint32_t *d_status, void *stream);

/* How a paired-end batch is launched (no reference counterpart: the reference's per-pair body is one function,
 * src/abismal.cpp:1950-1999).  By default the body is split where its register pressure is: a SEED kernel runs both
 * seed passes of every orientation call's two ends (process_seeds into pe_candidates, :1269-1375, :775-863) and hands the
 * finished candidate lists over in device memory; MATE kernels take them from there (prepare_for_mating, best_pair,
 * best_single, the single-end fallback: :844-852, :1715-1831).  Pairs whose sets outgrow the seed kernel are mapped whole
 * by one wave each with 32768-entry sets, as every heavy pair was before round 5.  Results do not depend on any of this.
 *   split      -1 = default (split), 0 = seeding and mating in ONE kernel per pair (rounds 1-4; same-box comparisons), 1 = split
 *   seed_cap   entries a candidate list may reach inside the seed kernel (0 = default; up to 128 a list stays in LDS,
 *              beyond that it moves to a per-wave staging area in device memory; at most 16384)
 *   hand_entries  list entries the hand-over area holds (0 = default: a multiple of the batch size; a list that finds
 *              no room sends its pair through the whole-pair kernel -- slower, never wrong) */
int abm_ctx_set_pe_split(abm_ctx *ctx, int split, uint32_t seed_cap, uint64_t hand_entries);
/* Measurement hook: pairs by route since the previous call, then reset -- out[0] mated from LDS lists, out[1] mapped
 * whole (sets that outgrew the seed kernel), out[2] mated from lists in device memory -- and out[3] the hand-over entries
 * the context's last batch asked for (waits for the device). */
int abm_ctx_pe_split_stats(abm_ctx *ctx, uint64_t out[4]);
/* HIP-event brackets (abm_ctx_set_timing) the context's last paired-end call recorded, in launch order: split -- seed,
 * mate (LDS lists), mate (lists in device memory), whole pairs = 4; unsplit -- tier 1, tier 2 = 2. */
uint32_t abm_ctx_pe_timed_launches(const abm_ctx *ctx);
/* page-locked host memory the context holds for results on their way out (its pinned staging buffers), bytes */
uint64_t abm_ctx_pinned_bytes(const abm_ctx *ctx);

/* the arena of CIGARs longer than their slot left by the context's last device call (waits for it) */
int abm_ctx_long_cigars(abm_ctx *ctx, uint32_t *out_ops, uint64_t capacity, uint64_t *n_ops);

enum {
  ABM_STATUS_CIGAR_OVERFLOW = 1u, /* the arena for CIGARs longer than a slot ran out */
  ABM_STATUS_READ_TOO_LONG = 2u,  /* a read (or an end of a pair) exceeded abm_max_read_length() */
  ABM_STATUS_SET_OVERFLOW = 4u    /* PE candidate set outgrew its workspace */
};
/* Longest read (single-end, or end of a pair) that is mapped: 32766 bases, the reference's own limit (ReadLoader refuses
 * reads of 32767 bases or more, src/abismal.cpp:179-185).  Reads of up to 1024 bases are mapped by a batch's main
 * launch; longer ones -- and pairs with such an end -- by a launch of their own afterwards (traceback tables, and for
 * pairs the read data too, in global memory; the device entry point then waits once for the device, to learn how many
 * there are). */
uint32_t abm_max_read_length(void);
/* Reads (pairs with an end) beyond abm_max_read_length() handed to this context's host entry points so far: they come
 * back without a hit, everything else in their batch is mapped as usual. */
uint64_t abm_ctx_reads_too_long(abm_ctx *ctx);

/* Which form of the genome the Hamming filter of this context's device reads for batches of reads up to 448 bases:
   1 = the bit planes derived at upload (cooperative window loads; every genome without IUPAC ambiguity letters),
   0 = the nibble array, one lane per window (genomes with IUPAC letters).  Results are the same; the rate is not. */
int abm_ctx_filter_on_planes(const abm_ctx *ctx);

/* Measurement hook (no reference counterpart): exact work tallies accumulated by the launches on this context since
 * the previous call, then reset -- of the paired-end kernels always, of the single-end kernel only in its diagnostic
 * build (abm_ctx_set_phase_stamps: the production kernel keeps none, they cost it registers):
 * [0] seed offsets probed, [1] bucket-narrowing search probes, [2] candidates
 * compared, [3] read words compared, [4] candidate-set updates, [5] alignments.
 * Feeds the algorithmic-bytes model of SURVEY.md section 8(d). */
int abm_ctx_take_work(abm_ctx *ctx, uint64_t out[16]);
/* Diagnostic build of the mapping kernel: adds s_memtime stamps per phase;
 * take_work then also returns summed shader cycles in [6] probe+narrow,
 * [7] candidate gather+Hamming, [8] ordered replay, [9] alignment, [10] total.
 * Never enable it for a run whose time is quoted. */
int abm_ctx_set_phase_stamps(abm_ctx *ctx, int enable);
/* Paired-end runs keep separate tallies per tier: out[0..15] tier 1 (sets of up
 * up to 256 entries), out[16..31] tier 2 (the pairs redone with 32768-entry
 * sets); take_work returns their sum.  In the paired-end tallies [3] is the number of 128-byte lines the checked
 * buckets' index-entry runs span (the fetched windows' words follow from [2] and [11]).  With phase stamps the paired-end kernel
 * fills [6] probe+narrow, [7] gather+Hamming, [8] replay, [9] single-end
 * fallback, [10] total, [12] sort+unique, [13] pairable-entry alignments,
 * [14] mating + tracebacks, [15] best_single replay. */
int abm_ctx_take_work_tiers(abm_ctx *ctx, uint64_t out[32]);
/* Diagnostic kernel only: device array [n] receiving per-read shader cycles / 1024 (NULL = off). */
int abm_ctx_set_read_cycles(abm_ctx *ctx, uint32_t *d_read_cycles);
/* Paired-end diagnostic kernels only: device array [n][8] (zeroed by the caller) to which the kernels that mate a pair add
 * its shader cycles / 1024 by phase -- probe + narrow, window gather + Hamming, replay, sort + unique, scoring of the
 * pairable entries, mating + tracebacks, best_single, single-end fallback (NULL = off). */
int abm_ctx_set_pair_phases(abm_ctx *ctx, uint32_t *d_pair_phases);

/* Measurement hook: when enabled, every mapping-kernel launch is bracketed by
 * HIP events recorded on the stream it is launched on; take_kernel_time waits
 * for them and returns the number of launches and their summed duration. */
int abm_ctx_set_timing(abm_ctx *ctx, int enable);
int abm_ctx_take_kernel_time(abm_ctx *ctx, uint64_t *launches, double *total_ms);
/* Same, one duration per launch in launch order (paired-end: abm_ctx_pe_timed_launches() per call). */
int abm_ctx_take_kernel_times(abm_ctx *ctx, double *ms_out, uint64_t capacity, uint64_t *launches);

/* Mapping statistics are six counters per struct (src/abismal.cpp:865-895) in
 * up to three structs (pairs, read1, read2: :1034-1037) = 18 x u64.  Sums them
 * over every context in ctxs[] with one RCCL all-reduce (ncclSum over xGMI);
 * counters[k] points at the 18 host values of context k and receives the sum.
 * Contexts that share a device (two replicas on one GPU) are summed on the host instead: a communicator takes a
 * device once. */
int abm_stats_allreduce(abm_ctx *const *ctxs, int n_ctx, uint64_t *const *counters);

#ifdef __cplusplus
}
#endif
#endif /* ABISMAL_AMD_H */
