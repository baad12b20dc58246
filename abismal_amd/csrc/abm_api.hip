// abismal_amd: C ABI (include/abismal_amd.h) over the HIP kernels.
#include "../../include/abismal_amd.h"
#include "abm_index_file.hpp"
#include "abm_index_build.hpp"
#include "abm_kernels.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;

struct HipFail : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void hip_check(hipError_t e, const char *what) {
  if (e != hipSuccess)
    throw HipFail(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(x) hip_check((x), #x)

template <class F> int guarded(F &&f) {
  try { f(); return 0; }
  catch (const std::length_error &e) { g_error = e.what(); return ABM_ERR_CAPACITY; }
  catch (const std::exception &e) { g_error = e.what(); return -1; }
  catch (...) { g_error = "unknown error"; return -1; }
}

// ABM_TRACE_HOST=1: wall-clock of the host entry points' sections on stderr (diagnostic)
struct HostTrace {
  bool on;
  std::chrono::steady_clock::time_point t;
  explicit HostTrace() : on(std::getenv("ABM_TRACE_HOST") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    static const auto t_proc = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[abm host] t=%9.1f ms thr %04x %-22s %8.2f ms\n", std::chrono::duration<double, std::milli>(now - t_proc).count(),
                 static_cast<unsigned>(std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xFFFF), what,
                 std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// Experiment switches (ABM_COOP_WINDOWS, ABM_GRID_WAVES, ABM_PLANES_COPIES, ABM_EXT_LETTERS, ABM_PE_WPS / ABM_PE_WPS2) are honoured only when
// ABM_EXPERIMENTS=1 is set as well: a stray variable in a production environment changes nothing.
const char *experiment_env(const char *name) {
  static const bool on = [] { const char *e = std::getenv("ABM_EXPERIMENTS"); return e && e[0] == '1'; }();
  return on ? std::getenv(name) : nullptr;
}

// A buffer that grows frees its old allocation, and hipFree / hipHostFree wait for the whole device (the other context's
// mapping kernel included) holding the runtime's lock: abm_ctx_reserve exists so that it never happens mid-run, and
// ABM_TRACE_HOST=1 reports it if it does.
void note_regrowth(const char *what, size_t from, size_t to) {
  if (std::getenv("ABM_TRACE_HOST") != nullptr)
    std::fprintf(stderr, "[abm host] %s buffer regrown %zu -> %zu bytes (frees wait for the device)\n", what, from, to);
}

template <class T> struct DevBuf {  // grow-only device allocation
  T *p = nullptr;
  size_t cap = 0;
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) { note_regrowth("device", cap * sizeof(T), n * sizeof(T)); HIPCHK(hipFree(p)); }
    p = nullptr; cap = 0;
    HIPCHK(hipMalloc(&p, n * sizeof(T)));
    cap = n;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// pinned host staging (grow-only): device -> host copies into pinned memory run at the link's rate
template <class T> struct HostBuf {
  T *p = nullptr;
  size_t cap = 0;
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) { note_regrowth("pinned host", cap * sizeof(T), n * sizeof(T)); HIPCHK(hipHostFree(p)); }
    p = nullptr; cap = 0;
    // (portable + mapped: kernels of whichever device the context lives on write results straight into these buffers)
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(T), hipHostMallocPortable | hipHostMallocMapped));
    cap = n;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// the index arrays resident on one device, shared by every context created on it
struct DeviceReplica {
  void *arena = nullptr;  // one allocation holding the seven index arrays
  abm::DevIndex dix{};
  int refs = 0;
  void *ext_mem[3] = {nullptr, nullptr, nullptr};  // seed-extension tables (abm_ext.hip), built for dix.ext_maxc candidates
  void *wrec_mem = nullptr;                         // window records (DevIndex::wrec)
  uint64_t plane_blocks = 0;                        // blocks of one bit-plane copy (guard blocks included)
  abm::u32 ext_tried = 0;                           // max_candidates of the last build attempt (it may have built nothing)
  double ext_build_s = 0;
  std::mutex mu;  // guards arena / refs / tables: the replicas of different devices are set up side by side
  // The single-end host-buffer entry points of the contexts on one device take turns for the mapping
  // kernel (their transfers overlap freely): the kernels are bound by random line fetches, and two of them
  // resident together only evict each other's lines (see kSeWavesPerCu).
  std::mutex kernel_turn;
};

struct abm_index {
  abm::HostIndex h;
  int want_e2 = -1, want_e3 = -1;  // letters of the seed-extension tables; -1 = chosen from the index's size
  int cap_e2 = 7, cap_e3 = 4;      // ... but no more than these (abm_index_set_seed_extension_cap)
  int wrec_len = 0;                // window records for reads of up to this many bases (abm_index_set_window_records); 0 = none
  mutable std::mutex mu;  // guards the map itself and the wishes below (a replica's contents: DeviceReplica::mu)
  mutable std::map<int, DeviceReplica> replicas;  // by device ordinal; nodes stay for the index's lifetime
  uint32_t want_maxc = 0;  // max_candidates the tables are built for; 0 = the index file's
  uint32_t direct_min = abm::kDirectMin;  // pair kernels: smallest range narrowed directly (abm_index_set_direct_narrowing)
  uint32_t direct_min_se = abm::kDirectMinSe;  // ... and the single-end kernel's
};

struct abm_ctx {
  int device = 0;
  const abm_index *ix = nullptr;
  abm::DevIndex dix{};            // as of creation (arrays, planes); the seed-extension tables are read from the replica per call
  DeviceReplica *rep = nullptr;
  bool holds_replica = false;
  hipStream_t stream = nullptr;  // the host-buffer entry points run on the context's own stream
  std::mutex *kernel_turn = nullptr;
  abm::u32 *drained = nullptr;   // pinned, device-mapped: the mapping kernel has handed out its last read
  bool signal_drained = false;  // set by the host-buffer entry point around its launch
  // per-batch workspaces (grow-only; sized by the largest batch seen)
  DevBuf<abm::u64> packed, packed2;
  DevBuf<abm::u32> lens2, subset, subset_count, payload1, payload2, list2, heap2, log2;
  DevBuf<abm::u8> need_big;
  // the paired-end phase split (seed kernel -> hand-over area -> mate kernels; PeArgs::hand_*): list headers per pair,
  // the lists' entries, the bump counter, the seed kernel's per-wave staging area, the second route's pair list, and
  // {pairs by route (3), unused}
  DevBuf<abm::u32> hand_hdr, hand_pos, stage_pos, subset_b, subset_count_b, class33_b;
  DevBuf<abm::i16> hand_d, stage_d;
  DevBuf<unsigned long long> hand_count, split_stats;
  size_t hand_want = 0;  // hand-over entries to reserve at least (abm_ctx_set_pe_split)
  int pe_split = -1;     // -1: default (split), 0: tier 1 unsplit, 1: split
  uint32_t pe_scap = 0;  // 0: default
  uint32_t pe_timed_launches = 0;  // HIP-event brackets the last paired-end call recorded (abm_ctx_set_timing)
  DevBuf<abm::Hit> pe_out;  // staging: pairs (20 B each) then se1, se2
  DevBuf<abm::u32> cig2h, cig_n2h;
  DevBuf<char> blob2;
  DevBuf<unsigned long long> coff;
  DevBuf<char> scan_tmp;
  DevBuf<abm::u32> cblob;
  DevBuf<abm::u64> off2;
  DevBuf<abm::u32> lens, order, class33;
  DevBuf<abm::u32> long_list, long_count, long_ctmp;  // the long-read launch (se_long_reads): listed reads, per-wave scratch
  DevBuf<abm::u64> packed_long, packed_long2, long_q;  // (long_q, packed_long2: the paired-end long-end launch, pe_long_pairs)
  DevBuf<abm::u8> long_tb;
  DevBuf<abm::u8> cls;
  DevBuf<unsigned long long> work;
  DevBuf<unsigned long long> next_read;
  DevBuf<abm::u32> cig_arena, cig_arena_count;  // CIGARs longer than a slot (CigarSink)
  size_t arena_want = 0;                        // arena size the host entry points ask for (0 = default)
  uint64_t too_long = 0;                        // reads (pairs) beyond kMaxReadLen seen by the host entry points
  HostBuf<abm::u32> h_cn, h_slots, h_arena, h_cn2, h_slots2;
  HostBuf<uint64_t> h_rel, h_rel2;  // offsets relative to the batch's first read (batches that do not start at 0)
  HostBuf<abm::u32> h_tail;         // {arena count, status} of the last launch of a host-buffer entry point
  DevBuf<abm::u32> finished;
  // results leaving slice by slice (abm_map_se_batch_sliced): slice boundaries and per-read slice numbers on the device,
  // the reads each slice still waits for, the ordering kernels' histogram; one completion word per slice in pinned memory
  DevBuf<abm::u32> slice_first_d, slice_left, slice_hist;
  DevBuf<abm::u16> slice_id;
  HostBuf<abm::u32> h_slice_first, h_slice_done;
  uint32_t sliced_n = 0;           // slices of the launch being set up (0 = an ordinary launch)
  uint32_t sliced_stride = 0;      // slot width of the results abm_ctx_slice_results reads
  uint64_t sliced_reads = 0;
  bool host_results = false;        // set by abm_map_se_batch around its launches: arena and summary words in pinned memory
  // SAM text written by the single-end kernel (abm_ctx_set_sam_tails): the line after QNAME per read, in pinned memory
  bool sam_on = false;
  int sam_allow_ambig = 0;
  uint32_t sam_stride = 0;          // of the launch whose results the buffers hold
  HostBuf<char> h_sam;
  HostBuf<abm::u32> h_sam_len;
  HostBuf<abm_hit> h_res;           // hits on their way out (a pinned target keeps the copy on the DMA engines)
  HostBuf<abm_hit> h_pe_out;        // paired-end results (pairs, then both fallback hits) written by the kernels, pinned
  unsigned launch_seq = 0;
  // every device entry point reuses this context's workspaces: a call first makes its stream wait for
  // the previous call's work (whatever stream that ran on), so consecutive calls never overlap
  hipEvent_t last_done = nullptr;
  std::map<uint64_t, int> se_waves;
  // staging for the host-buffer entry points
  DevBuf<char> blob;
  DevBuf<abm::u64> off;
  DevBuf<abm::Hit> res;
  DevBuf<abm::u32> cig, cig_n, status;
  // optional HIP-event timing of the mapping kernel (abm_ctx_set_timing)
  bool timing = false;
  abm::u32 *read_cycles = nullptr;  // caller-owned device array for the diagnostic kernel
  abm::u32 *pair_phases = nullptr;  // caller-owned device array [n][8] for the paired-end diagnostic kernels
  bool phase_stamps = false;  // launch the diagnostic kernel variant with in-kernel phase stamps
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  std::mutex mu;
};

namespace {

// Seed-extension tables of one device replica for `maxc` candidates (see abm_ext.hip).  Letters per table: asked
// for through abm_index_set_seed_extension, else the fewest that give the tables as many keys as the genome has
// bases -- none below 33 Mbp, 7 and 4 at hg38 scale (34 + 2 x 28 GB) -- and never more than fit half of the
// device's free memory.  Genomes with IUPAC letters get none (their base-3 digits are not the sort's symbols).
void free_ext(DeviceReplica &rep) {
  for (auto &m : rep.ext_mem) { if (m) (void)hipFree(m); m = nullptr; }
  rep.dix.ext2 = rep.dix.ext3t = rep.dix.ext3a = nullptr;
  rep.dix.e2 = rep.dix.e3 = rep.dix.ext_maxc = 0;
}
void build_ext(DeviceReplica &rep, const abm_index &ix, abm::u32 maxc) {
  free_ext(rep);
  rep.ext_tried = maxc;
  const abm::HostIndex &h = ix.h;
  if (h.multibit_genome || maxc == 0) return;
  // (by the genome's length: the buckets that need narrowing belong to repeats, whose copy numbers grow with it --
  // measured at 3.1 Gbp, kernel time per 10 M reads: no tables 757-763 ms, 4+2 letters 739-747, 6+3 728-739, 7+4 719-728)
  const uint64_t n_bases = h.chrom_starts.empty() ? 0 : h.chrom_starts.back();
  int e2 = ix.want_e2, e3 = ix.want_e3;
  if (const char *e = experiment_env("ABM_EXT_LETTERS")) { int a = 0, b = 0; if (std::sscanf(e, "%d,%d", &a, &b) == 2) { e2 = a; e3 = b; } }
  if (e2 < 0) { e2 = 0; while (e2 < ix.cap_e2 && abm::ext_keys(0, e2) < n_bases) ++e2; }
  if (e3 < 0) { e3 = 0; while (e3 < ix.cap_e3 && abm::ext_keys(1, e3) < n_bases) ++e3; }
  e2 = std::min(e2, 7); e3 = std::min(e3, 4);
  if (e2 <= 0 || e3 <= 0) return;
  auto need = [&](int a, int b) {
    return 8 * (abm::ext_keys(0, a) + 2 * abm::ext_keys(1, b)) + std::max(abm::ext_scratch_bytes(0, a), abm::ext_scratch_bytes(1, b));
  };
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  while ((e2 > 1 || e3 > 1) && need(e2, e3) > free_b / 2) { if (e2 > 1) --e2; if (e3 > 1) --e3; }
  if (need(e2, e3) > free_b / 2) return;
  const auto t0 = std::chrono::steady_clock::now();
  void *scratch = nullptr;
  abm::u32 *d_fail = nullptr;
  try {
    HIPCHK(hipMalloc(&scratch, std::max(abm::ext_scratch_bytes(0, e2), abm::ext_scratch_bytes(1, e3))));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_fail), 64));
    HIPCHK(hipMemset(d_fail, 0, 64));
    const uint64_t n_idx[3] = {h.index.size(), h.index_t.size(), h.index_a.size()};
    for (int mode = 0; mode < 3; ++mode) {
      const abm::u32 extra = mode == 0 ? e2 : e3;
      HIPCHK(hipMalloc(&rep.ext_mem[mode], abm::ext_keys(mode, extra) * 8));
      HIPCHK(abm::build_ext_table(rep.dix, mode, extra, maxc, n_idx[mode], static_cast<uint2 *>(rep.ext_mem[mode]), scratch, d_fail, nullptr));
      HIPCHK(hipDeviceSynchronize());
    }
    abm::u32 fail = 0;
    HIPCHK(hipMemcpy(&fail, d_fail, 4, hipMemcpyDeviceToHost));
    (void)hipFree(scratch); scratch = nullptr;
    (void)hipFree(d_fail); d_fail = nullptr;
    if (fail) { free_ext(rep); return; }  // (an index whose hashed letters do not name its buckets: bisection only)
  }
  catch (...) { if (scratch) (void)hipFree(scratch); if (d_fail) (void)hipFree(d_fail); free_ext(rep); throw; }
  rep.dix.ext2 = static_cast<const uint2 *>(rep.ext_mem[0]);
  rep.dix.ext3t = static_cast<const uint2 *>(rep.ext_mem[1]);
  rep.dix.ext3a = static_cast<const uint2 *>(rep.ext_mem[2]);
  rep.dix.e2 = static_cast<abm::u32>(e2);
  rep.dix.e3 = static_cast<abm::u32>(e3);
  rep.dix.ext_maxc = maxc;
  rep.ext_build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Window records of one device replica (abm_ext.hip): for reads of up to ix.wrec_len bases, if the genome has bit planes,
// the record numbers fit 32 bits and the table fits a quarter of the free device memory.
void free_wrec(DeviceReplica &rep) {
  if (rep.wrec_mem) (void)hipFree(rep.wrec_mem);
  rep.wrec_mem = nullptr;
  rep.dix.wrec = nullptr;
  rep.dix.wrec_t0 = rep.dix.wrec_a0 = rep.dix.wrec_blocks = rep.dix.wrec_back = rep.dix.wrec_max_len = 0;
}
void build_wrec(DeviceReplica &rep, const abm_index &ix) {
  free_wrec(rep);
  int want = ix.wrec_len;
  if (const char *e = experiment_env("ABM_WINDOW_RECORDS")) want = std::atoi(e);
  if (want <= 0 || rep.dix.planes[0] == nullptr) return;
  const abm::HostIndex &h = ix.h;
  const abm::u32 blocks = abm::window_record_blocks_for(static_cast<abm::u32>(std::min(want, 172)));  // (groups of four lanes: reads up to 192 bases)
  const uint64_t n_idx[3] = {h.index.size(), h.index_t.size(), h.index_a.size()};
  const uint64_t n_entries = n_idx[0] + n_idx[1] + n_idx[2];
  if (n_entries == 0 || (n_entries + 2) * blocks >= 0xFFFFFF00ull) return;
  const size_t bytes = abm::window_record_bytes(n_entries, blocks);
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  if (bytes > free_b / 4) return;
  HIPCHK(hipMalloc(&rep.wrec_mem, bytes));
  try {
    HIPCHK(hipMemset(rep.wrec_mem, 0, bytes));
    HIPCHK(abm::build_window_records(rep.dix, rep.plane_blocks, h.chrom_starts.empty() ? 0 : h.chrom_starts.back(), n_idx, blocks, static_cast<abm::u64 *>(rep.wrec_mem), nullptr));
    HIPCHK(hipDeviceSynchronize());
  }
  catch (...) { free_wrec(rep); throw; }
  rep.dix.wrec = static_cast<const abm::u64 *>(rep.wrec_mem);
  rep.dix.wrec_t0 = static_cast<abm::u32>(n_idx[0]);
  rep.dix.wrec_a0 = static_cast<abm::u32>(n_idx[0] + n_idx[1]);
  rep.dix.wrec_blocks = blocks;
  rep.dix.wrec_max_len = abm::window_record_max_len(blocks);
  rep.dix.wrec_back = rep.dix.wrec_max_len - abm::kKeyWeight;
}

// the index as a launch sees it: the context's arrays plus the replica's seed-extension tables -- if they were built
// for the call's max_candidates; otherwise the call goes without tables (the kernels then bisect from the counters).
abm::DevIndex current_index(abm_ctx *ctx, abm::u32 maxc, bool single_end);

abm::u32 words_for(abm::u32 max_len) { return std::max(1u, (max_len + 15) / 16); }
// bytes of a read's SAM-text slot (SeArgs::sam_stride): longest read + longest chromosome name + the fixed fields and a
// slot's CIGAR as text, in whole 16 bytes
abm::u32 sam_stride_for(const abm_ctx *ctx, abm::u32 eff_len, abm::u32 cig_stride) {
  size_t longest_name = 0;
  for (const std::string &nm : ctx->ix->h.chrom_names) longest_name = std::max(longest_name, nm.size());
  return static_cast<abm::u32>((eff_len + longest_name + 64 + 12 * static_cast<size_t>(std::min<abm::u32>(cig_stride, 8)) + 15) & ~static_cast<size_t>(15));
}
abm::u32 bitwords_for(abm::u32 max_len) { return (max_len + 63) / 64 + 1; }

void check_params(const abm_params *p) {
  if (!p) throw std::invalid_argument("params is null");
  if (!(p->valid_frac >= 0.0 && p->valid_frac <= 1.0)) throw std::invalid_argument("valid_frac out of range");
}

// HIP-event bracket around one mapping-kernel launch (abm_ctx_set_timing); returns the closing event
hipEvent_t begin_timed(abm_ctx *ctx, hipStream_t st) {
  if (!ctx->timing) return nullptr;
  if (ctx->events_used == ctx->events.size()) {
    hipEvent_t x, y;
    HIPCHK(hipEventCreate(&x));
    HIPCHK(hipEventCreate(&y));
    ctx->events.emplace_back(x, y);
  }
  const hipEvent_t e0 = ctx->events[ctx->events_used].first, e1 = ctx->events[ctx->events_used].second;
  ++ctx->events_used;
  HIPCHK(hipEventRecord(e0, st));
  return e1;
}

// The launch for a batch's reads of kLdsReadLen + 1 .. kMaxReadLen bases (map_se_long_kernel): the reads are listed
// on the device, the list's length is fetched (the one place a device entry point waits for the device -- only when
// the caller announced such reads through max_len), and the list is mapped in rounds of at most 1024 reads, each
// packed into encodings of its own and mapped by one wave per CU with per-wave traceback tables in global memory.
// Results land where the main launch left these reads unmapped.  `a` = the main launch's arguments.
void se_long_reads(abm_ctx *ctx, const abm::SeArgs &main, uint64_t n, const char *d_blob, const uint64_t *d_off,
                   abm::u32 max_len, double valid_frac, hipStream_t st) {
  if (n >= (1ull << 32)) throw std::invalid_argument("batch too large for the long-read launch (>= 2^32 reads)");
  ctx->long_list.reserve(n);
  ctx->long_count.reserve(1);
  HIPCHK(hipMemsetAsync(ctx->long_count.p, 0, 4, st));
  HIPCHK(abm::launch_collect_long(ctx->lens.p, n, ctx->long_list.p, ctx->long_count.p, st));
  abm::u32 count = 0;
  HIPCHK(hipMemcpyAsync(&count, ctx->long_count.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const abm::u32 W = words_for(max_len), WB = bitwords_for(max_len);
  const double size_frac = ctx->ix->h.multibit_genome ? 1.0 : valid_frac;
  const abm::u32 GW = abm::se_window_words(max_len, size_frac);
  const abm::u32 cap2 = (max_len + 2 + 1) & ~1u;
  int waves = 0;
  if (count) {
    waves = abm::se_long_resident_waves(W, WB, GW);
    if (waves <= 0) throw HipFail("map_se_long_kernel does not fit on this device (LDS)");
  }
  const abm::u32 round = 1024;
  for (abm::u32 at = 0; at < count; at += round) {
    const abm::u32 m = std::min(round, count - at);
    const abm::u32 grid = std::min<abm::u32>(m, static_cast<abm::u32>(waves));
    ctx->packed_long.reserve(static_cast<size_t>(m) * 4 * W);
    ctx->long_tb.reserve(static_cast<size_t>(grid) * abm::se_long_tb_bytes(max_len));
    ctx->long_ctmp.reserve(static_cast<size_t>(grid) * cap2);
    HIPCHK(abm::launch_pack_listed(d_blob, reinterpret_cast<const abm::u64 *>(d_off), ctx->long_list.p + at, m, W, ctx->packed_long.p, st));
    abm::SeArgs a = main;
    a.packed = ctx->packed_long.p;
    a.order = ctx->long_list.p + at;
    a.n_reads = m;
    a.W = W; a.WB = WB; a.GW = GW;
    a.max_len = max_len;
    a.tb_extra = 0;
    a.G = 0;
    a.size_frac = size_frac;
    a.ctmp_cap = max_len + 2;
    a.long_tb = ctx->long_tb.p;
    a.long_ctmp = ctx->long_ctmp.p;
    a.long_tb_bytes = abm::se_long_tb_bytes(max_len);
    a.drained = nullptr;
    a.finished = nullptr;
    a.host_tail = nullptr;
    a.read_cycles = nullptr;
    unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    a.next_read = counter;
    HIPCHK(abm::launch_map_se_long(a, grid, st));
  }
  if (ctx->host_results) {  // what the main launch's last wave would have published (abm_map_se_batch waits for the stream)
    HIPCHK(hipMemcpyAsync(&ctx->h_tail.p[0], main.cig_arena_count, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&ctx->h_tail.p[1], main.status, 4, hipMemcpyDeviceToHost, st));
  }
}

void se_device(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n, const char *d_blob,
               const uint64_t *d_off, uint32_t max_len, abm_hit *d_res, uint32_t *d_cig,
               uint32_t cig_stride, uint32_t *d_cig_n, uint32_t *d_status, hipStream_t st) {
  check_params(params);
  if (mode < 0 || mode > 2) throw std::invalid_argument("bad single-end mode");
  if (cig_stride == 0) throw std::invalid_argument("cig_stride must be > 0");
  if (n == 0) return;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamWaitEvent(st, ctx->last_done, 0));
  // reads beyond kLdsReadLen bases (rare: long-read libraries) are left out of this launch -- its workspaces, LDS and
  // filter stay what the batch's ordinary reads need -- and mapped by a launch of their own afterwards (se_long_reads)
  const bool has_long = max_len > abm::kLdsReadLen;
  const abm::u32 eff_len = std::min<abm::u32>(std::max<abm::u32>(max_len, 1), abm::kLdsReadLen);
  const abm::u32 W = words_for(eff_len), WB = bitwords_for(eff_len);
  ctx->packed.reserve(n * 4 * W);
  ctx->lens.reserve(n);
  ctx->work.reserve(32);
  HIPCHK(abm::launch_pack_reads(d_blob, reinterpret_cast<const abm::u64 *>(d_off), n, W, ctx->packed.p,
                                ctx->lens.p, st));
  abm::SeArgs a{};
  a.ix = current_index(ctx, params->max_candidates ? params->max_candidates : ctx->dix.max_candidates, true);
  const bool sliced = ctx->sliced_n != 0 && !has_long && n < (1ull << 32);
  if (sliced) {  // (slice boundaries are uploaded by the entry point)
    ctx->order.reserve(n);
    ctx->cls.reserve(n);
    ctx->slice_id.reserve(n);
    ctx->slice_left.reserve(ctx->sliced_n);
    ctx->slice_hist.reserve(abm::order_sliced_hist_words(ctx->sliced_n));
    HIPCHK(abm::launch_order_reads_sliced(a.ix, ctx->packed.p, ctx->lens.p, n, W, mode, ctx->cls.p, ctx->slice_first_d.p,
                                          ctx->sliced_n, ctx->slice_id.p, ctx->slice_hist.p, ctx->slice_left.p, ctx->order.p, st));
    a.order = ctx->order.p;
    a.slice_id = ctx->slice_id.p;
    a.slice_left = ctx->slice_left.p;
    a.slice_done = ctx->h_slice_done.p;
  }
  else if (n < (1ull << 32)) {
    ctx->order.reserve(n);
    ctx->cls.reserve(n);
    ctx->class33.reserve(33);
    HIPCHK(abm::launch_order_reads(a.ix, ctx->packed.p, ctx->lens.p, n, W, mode, ctx->cls.p, ctx->class33.p,
                                   ctx->order.p, st));
    a.order = ctx->order.p;
  }
  a.packed = ctx->packed.p;
  a.lens = ctx->lens.p;
  a.n_reads = n;
  a.W = W;
  a.WB = WB;
  a.max_len = eff_len;
  // IUPAC genome letters can drive a Hamming sum below zero, which the reference turns into the
  // widest band (61); size the LDS for it only when such letters exist
  const double size_frac = ctx->ix->h.multibit_genome ? 1.0 : params->valid_frac;
  a.size_frac = size_frac;
  a.GW = abm::se_window_words(eff_len, size_frac);
  a.tb_extra = abm::tb_extra_bytes(a.GW, eff_len, size_frac);
  // cooperative window loads from the genome's bit planes (hamming_planes): not for genomes with IUPAC letters (no
  // planes; their admission rule needs full_compare's word-by-word running sums) nor for reads beyond 448 bases; ABM_COOP_WINDOWS=0 switches them off (experiments)
  // (G lanes x 64 bases cover a window of eff_len + 63 bases)
  a.G = a.ix.planes[0] == nullptr ? 0u : (eff_len <= 2 * abm::kPlaneBlock ? 2u : (eff_len <= 4 * abm::kPlaneBlock - 64 ? 4u : (eff_len <= 8 * abm::kPlaneBlock - 64 ? 8u : 0u)));
  if (const char *e = experiment_env("ABM_COOP_WINDOWS")) { if (e[0] == '0') a.G = 0; else if (e[0] == '4' && a.G == 2) a.G = 4; }
  a.mode = mode;
  a.valid_frac = params->valid_frac;
  a.res = reinterpret_cast<abm::Hit *>(d_res);
  a.cig = d_cig;
  a.cig_stride = cig_stride;
  a.ctmp_cap = eff_len + 2;
  {  // arena for the CIGARs that outgrow their slot: few reads do, one op per read is ample
    // (150-base reads with sim-like indel rates need 1.5 ops of arena per read beside their 4-op slots; an arena that
    // overflows costs a second mapping of the batch)
    const size_t want = std::max<size_t>(ctx->arena_want, std::max<size_t>(1u << 16, 2 * n));
    ctx->cig_arena_count.reserve(1);
    HIPCHK(hipMemsetAsync(ctx->cig_arena_count.p, 0, 4, st));
    a.cig_arena_count = ctx->cig_arena_count.p;
    if (ctx->host_results) {  // (abm_map_se_batch: the arena lies in pinned host memory, like the rest of its results)
      ctx->h_arena.reserve(std::min<size_t>(want, 0xFFFFFF00u));
      ctx->h_tail.reserve(2);
      ctx->finished.reserve(1);
      HIPCHK(hipMemsetAsync(ctx->finished.p, 0, 4, st));
      ctx->h_tail.p[0] = ctx->h_tail.p[1] = 0;
      a.cig_arena = ctx->h_arena.p;
      a.cig_arena_cap = static_cast<abm::u32>(std::min<size_t>(ctx->h_arena.cap, 0xFFFFFF00u));
      a.finished = ctx->finished.p;
      a.host_tail = has_long ? nullptr : ctx->h_tail.p;  // (with a long-read launch to follow, the summary words are copied out after it)
    }
    else {
      ctx->cig_arena.reserve(std::min<size_t>(want, 0xFFFFFF00u));
      a.cig_arena = ctx->cig_arena.p;
      a.cig_arena_cap = static_cast<abm::u32>(std::min<size_t>(ctx->cig_arena.cap, 0xFFFFFF00u));
    }
  }
  a.cig_n = d_cig_n;
  a.status = d_status;
  a.work = ctx->work.p;
  a.blob = d_blob;
  a.off = reinterpret_cast<const abm::u64 *>(d_off);
  ctx->sam_stride = 0;
  if (ctx->sam_on && ctx->host_results && sliced) {
    // the kernel writes every read's SAM text (after QNAME) next to its hit; a line that does not fit its slot is formatted
    // by the host as before
    const abm::u32 stride = sam_stride_for(ctx, eff_len, cig_stride);
    ctx->h_sam.reserve(static_cast<size_t>(n) * stride);
    ctx->h_sam_len.reserve(n);
    a.sam_tail = ctx->h_sam.p;
    a.sam_len = ctx->h_sam_len.p;
    a.sam_stride = stride;
    a.sam_allow_ambig = ctx->sam_allow_ambig;
    ctx->sam_stride = stride;
  }
  // a small ring of work counters so launches queued on different streams never share one
  ctx->next_read.reserve(64);
  auto fresh_counter = [&](hipStream_t on) {
    unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), on));
    return counter;
  };
  a.read_cycles = ctx->phase_stamps ? ctx->read_cycles : nullptr;
  // the occupancy query costs milliseconds: remember it per launch shape
  const uint64_t shape = (static_cast<uint64_t>(W) << 48) ^ (static_cast<uint64_t>(eff_len) << 8) ^ static_cast<uint64_t>(size_frac * 255.0);
  int waves;
  auto it = ctx->se_waves.find(shape);
  if (it != ctx->se_waves.end()) waves = it->second;
  else { waves = abm::se_resident_waves(W, WB, a.ctmp_cap, eff_len, size_frac); ctx->se_waves[shape] = waves; }
  if (waves <= 0) throw HipFail("map_se_kernel does not fit on this device (LDS/occupancy)");
  abm::u32 grid = static_cast<abm::u32>(waves);  // persistent: one wave per resident slot
  if (const char *e = experiment_env("ABM_GRID_WAVES")) grid = std::max(64, std::atoi(e));

  const hipEvent_t e1 = begin_timed(ctx, st);
  a.next_read = fresh_counter(st);
  a.drained = ctx->signal_drained ? ctx->drained : nullptr;
  HIPCHK(abm::launch_map_se(a, eff_len, grid, ctx->phase_stamps, st));
  if (e1) HIPCHK(hipEventRecord(e1, st));
  if (has_long) se_long_reads(ctx, a, n, d_blob, d_off, std::min<abm::u32>(max_len, abm::kMaxReadLen), params->valid_frac, st);
  HIPCHK(hipEventRecord(ctx->last_done, st));
}

// CIGARs as the kernels leave them -- `stride` ops per read in fixed slots, longer ones whole in the launch's
// arena with slot[0] = where -- into the caller's compact blob + n + 1 offsets, in read order.  Host-side on
// purpose: no kernel has to run after the mapping kernel, so a batch's results leave the GPU while the next
// batch's (persistent, device-filling) mapping kernel is already running.
// fn(lo, hi) over [0, n) on a few host threads (the batch entry points' passes over per-read arrays: with 8 M reads
// per batch a single-threaded pass costs tens of ms -- hundreds while the CLI's parser threads own the memory bus)
template <class F> void parallel_ranges(uint64_t n, F &&fn) {
  const unsigned nt = n > (1u << 18) ? 8u : 1u;
  if (nt == 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  std::exception_ptr err;
  std::mutex emu;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      try { fn(n * t / nt, n * (t + 1) / nt); }
      catch (...) { std::lock_guard<std::mutex> lk(emu); err = std::current_exception(); }
    });
  for (auto &x : th) x.join();
  if (err) std::rethrow_exception(err);
}

// offsets of a host batch: monotone?  longest read, reads beyond the kernels' cap; offsets relative to the first
// (into `rel`, pinned, when the batch does not start at 0 -- else the caller's array is used as it is)
struct OffsetScan { uint32_t max_len = 0; uint64_t too_long = 0; const uint64_t *use = nullptr; };
OffsetScan scan_offsets(const uint64_t *seq_off, uint64_t n, HostBuf<uint64_t> &rel) {
  OffsetScan out;
  const uint64_t base = seq_off[0];
  if (base != 0) rel.reserve(n + 1);
  uint64_t *r = base != 0 ? rel.p : nullptr;
  std::mutex mu;
  parallel_ranges(n, [&](uint64_t lo, uint64_t hi) {
    uint32_t ml = 0;
    uint64_t tl = 0;
    for (uint64_t i = lo; i < hi; ++i) {
      if (seq_off[i + 1] < seq_off[i]) throw std::invalid_argument("seq_off not monotone");
      const uint64_t len = seq_off[i + 1] - seq_off[i];
      ml = std::max<uint32_t>(ml, static_cast<uint32_t>(std::min<uint64_t>(len, 0xFFFFFFFFu)));
      tl += len > abm::kMaxReadLen;
      if (r) r[i] = seq_off[i] - base;
    }
    std::lock_guard<std::mutex> lk(mu);
    out.max_len = std::max(out.max_len, ml);
    out.too_long += tl;
  });
  if (r) r[n] = seq_off[n] - base;
  out.use = r ? r : seq_off;
  return out;
}

// ops per CIGAR slot of the paired-end host entry point (longer CIGARs go to the arena): 150-base ends with sim-like
// indel rates have more than 4 ops a quarter of the time, more than 8 one time in fifty
constexpr uint32_t kPeHostSlotOps = 8;
void assemble_cigars(uint64_t n, uint32_t stride, const uint32_t *cn, const uint32_t *slots, const uint32_t *arena,
                     uint64_t arena_n, uint32_t *out_blob, uint64_t cap, uint64_t *out_off) {
  out_off[0] = 0;
  for (uint64_t i = 0; i < n; ++i) out_off[i + 1] = out_off[i] + cn[i];
  if (out_off[n] > cap) throw std::length_error("cig_capacity too small");
  auto fill = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; ++i) {
      const uint32_t k = cn[i];
      if (k == 0) continue;
      const uint32_t *src = slots + i * stride;
      if (k > stride) {
        if (static_cast<uint64_t>(src[0]) + k > arena_n) throw std::runtime_error("CIGAR arena reference out of range");
        src = arena + src[0];
      }
      std::memcpy(out_blob + out_off[i], src, k * 4ull);
    }
  };
  parallel_ranges(n, fill);
}

// tier 2's per-wave workspaces in global memory: scratch table, the two lists (positions, diffs, scores), heap / sort
// buffer, best_single log
void pe_tier2_reserve(abm_ctx *ctx, size_t waves) {
  const size_t cap = abm::kPeCapLarge;
  ctx->payload2.reserve(waves * cap);
  ctx->list2.reserve(waves * 4 * cap);
  ctx->heap2.reserve(waves * cap);
  ctx->log2.reserve(waves * (32 + 12 * cap));
}
// waves a tier-2 launch for reads of up to max_len bases can keep resident (what pe_device computes per call)
int pe_tier2_waves(abm_ctx *ctx, abm::u32 max_len, double valid_frac) {
  const abm::u32 eff_len = std::min<abm::u32>(std::max<abm::u32>(max_len, 1), abm::kLdsReadLen);
  const abm::u32 W = (eff_len + 15) / 16, WB = (eff_len + 63) / 64 + 1;
  const double size_frac = ctx->ix->h.multibit_genome ? 1.0 : valid_frac;
  const abm::u32 GW = abm::se_window_words(eff_len, size_frac);
  const bool coop = ctx->dix.planes[0] != nullptr && eff_len <= 8 * abm::kPlaneBlock - 64;
  const size_t lds = abm::pe_lds_bytes(W, WB, GW, eff_len + 2, eff_len, size_frac, abm::kPeCapLarge, true);
  return abm::pe_resident_waves(lds, true, abm::pe_waves_per_simd(lds, false, coop));
}

// The launch for a batch's pairs with an end of kLdsReadLen + 1 .. kMaxReadLen bases (map_pe_kernel<.., LONG>), after
// tiers 1 and 2 (which treat such a pair as empty): the pairs are listed on the device, the list's length is fetched
// (the one place the paired-end device entry point waits for the device -- only when the caller announced such ends
// through max_len), and the list is mapped in rounds, both ends packed into encodings of their own, by one wave per CU
// with its read data, traceback table, CIGAR scratch and tier 2's lists in global memory.  `main` = tier 2's arguments.
void pe_long_pairs(abm_ctx *ctx, const abm::PeArgs &main, uint64_t n, const char *d_blob1, const uint64_t *d_off1,
                   const char *d_blob2, const uint64_t *d_off2, abm::u32 max_len, double valid_frac, hipStream_t st) {
  ctx->long_list.reserve(n);
  ctx->long_count.reserve(2);
  HIPCHK(hipMemsetAsync(ctx->long_count.p, 0, 8, st));
  HIPCHK(abm::launch_collect_long_pairs(ctx->lens.p, ctx->lens2.p, n, ctx->long_list.p, ctx->long_count.p, st));
  abm::u32 count = 0;
  HIPCHK(hipMemcpyAsync(&count, ctx->long_count.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (count == 0) return;
  const abm::u32 W = words_for(max_len), WB = bitwords_for(max_len);
  const double size_frac = ctx->ix->h.multibit_genome ? 1.0 : valid_frac;
  const abm::u32 GW = abm::se_window_words(max_len, size_frac);
  const abm::u32 cap2 = (max_len + 2 + 1) & ~1u;
  const int waves = abm::pe_long_resident_waves(GW);
  if (waves <= 0) throw HipFail("the paired-end long-end launch does not fit on this device (LDS)");
  const abm::u32 round = 256;  // (a round's packed encodings: 2 x 256 x 4 W words = 32 MB at the longest reads)
  for (abm::u32 at = 0; at < count; at += round) {
    const abm::u32 m = std::min(round, count - at);
    const abm::u32 grid = std::min<abm::u32>(m, static_cast<abm::u32>(waves));
    const size_t cap = abm::kPeCapLarge;
    ctx->packed_long.reserve(static_cast<size_t>(m) * 4 * W);
    ctx->packed_long2.reserve(static_cast<size_t>(m) * 4 * W);
    ctx->long_q.reserve(static_cast<size_t>(grid) * abm::pe_long_q_words(W, WB));
    ctx->long_tb.reserve(static_cast<size_t>(grid) * abm::se_long_tb_bytes(max_len));
    ctx->long_ctmp.reserve(static_cast<size_t>(grid) * cap2);
    ctx->payload2.reserve(static_cast<size_t>(grid) * cap);
    ctx->list2.reserve(static_cast<size_t>(grid) * 4 * cap);
    ctx->heap2.reserve(static_cast<size_t>(grid) * cap);
    ctx->log2.reserve(static_cast<size_t>(grid) * (32 + 12 * cap));
    HIPCHK(abm::launch_pack_listed(d_blob1, reinterpret_cast<const abm::u64 *>(d_off1), ctx->long_list.p + at, m, W, ctx->packed_long.p, st));
    HIPCHK(abm::launch_pack_listed(d_blob2, reinterpret_cast<const abm::u64 *>(d_off2), ctx->long_list.p + at, m, W, ctx->packed_long2.p, st));
    abm::PeArgs a = main;
    a.packed1 = ctx->packed_long.p; a.packed2 = ctx->packed_long2.p;
    a.order = nullptr;
    a.subset = ctx->long_list.p + at;
    HIPCHK(hipMemcpyAsync(ctx->long_count.p + 1, &m, 4, hipMemcpyHostToDevice, st));  // (pageable source: copied before the call returns)
    a.subset_count = ctx->long_count.p + 1;
    a.W = W; a.WB = WB; a.GW = GW;
    a.max_len = max_len;
    a.tb_extra = 0;
    a.G = 0;
    a.ctmp_cap = max_len + 2;
    a.cap = static_cast<abm::u32>(cap);
    a.log_ws = ctx->log2.p; a.heap_ws = ctx->heap2.p; a.payload_ws = ctx->payload2.p; a.list_ws = ctx->list2.p;
    a.long_q = ctx->long_q.p;
    a.long_tb = ctx->long_tb.p;
    a.long_ctmp = ctx->long_ctmp.p;
    a.long_tb_bytes = abm::se_long_tb_bytes(max_len);
    a.pair_diag = nullptr;
    a.pair_phases = nullptr;
    unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    a.next_read = counter;
    HIPCHK(abm::launch_map_pe_long(a, grid, st));
    HIPCHK(hipStreamSynchronize(st));  // (the next round reuses the packed encodings and the count word)
  }
}

void pe_device(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n, const char *d_blob1,
               const uint64_t *d_off1, const char *d_blob2, const uint64_t *d_off2, uint32_t max_len,
               abm_pair *d_pair, abm_hit *d_se1, abm_hit *d_se2, uint32_t *d_cig1, uint32_t *d_cig2,
               uint32_t cig_stride, uint32_t *d_cig_n1, uint32_t *d_cig_n2, uint32_t *d_status, hipStream_t st) {
  check_params(params);
  if (mode < 0 || mode > 2) throw std::invalid_argument("bad paired-end mode");
  if (cig_stride == 0) throw std::invalid_argument("cig_stride must be > 0");
  if (n == 0) return;
  if (n >= (1ull << 32)) throw std::invalid_argument("batch too large (>= 2^32 pairs)");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamWaitEvent(st, ctx->last_done, 0));
  const abm::u32 eff_len = std::min<abm::u32>(std::max<abm::u32>(max_len, 1), abm::kLdsReadLen);  // (pairs: no long-read launch)
  const abm::u32 W = words_for(eff_len), WB = bitwords_for(eff_len);
  ctx->packed.reserve(n * 4 * W);
  ctx->packed2.reserve(n * 4 * W);
  ctx->lens.reserve(n);
  ctx->lens2.reserve(n);
  ctx->work.reserve(32);
  ctx->order.reserve(n);
  ctx->cls.reserve(n);
  ctx->class33.reserve(33);
  ctx->need_big.reserve(n);
  ctx->subset.reserve(n);
  ctx->subset_count.reserve(1);
  ctx->next_read.reserve(64);
  HIPCHK(abm::launch_pack_reads(d_blob1, reinterpret_cast<const abm::u64 *>(d_off1), n, W, ctx->packed.p, ctx->lens.p, st));
  HIPCHK(abm::launch_pack_reads(d_blob2, reinterpret_cast<const abm::u64 *>(d_off2), n, W, ctx->packed2.p, ctx->lens2.p, st));
  abm::PeArgs a{};
  a.ix = current_index(ctx, params->max_candidates ? params->max_candidates : ctx->dix.max_candidates, false);
  HIPCHK(abm::launch_order_reads(a.ix, ctx->packed.p, ctx->lens.p, n, W, mode == 1 ? 1 : 0, ctx->cls.p, ctx->class33.p,
                                 ctx->order.p, st));
  a.packed1 = ctx->packed.p; a.packed2 = ctx->packed2.p;
  a.lens1 = ctx->lens.p; a.lens2 = ctx->lens2.p;
  a.order = ctx->order.p;
  a.subset = ctx->subset.p; a.subset_count = ctx->subset_count.p;
  a.n_pairs = n;
  const double size_frac = ctx->ix->h.multibit_genome ? 1.0 : std::max(params->valid_frac, params->valid_frac);
  a.W = W; a.WB = WB; a.max_len = eff_len; a.GW = abm::se_window_words(eff_len, size_frac);
  a.tb_extra = abm::tb_extra_bytes(a.GW, eff_len, size_frac);
  // cooperative window loads from the bit planes, as in the single-end path (see there)
  a.G = a.ix.planes[0] == nullptr ? 0u : (eff_len <= 4 * abm::kPlaneBlock - 64 ? 4u : (eff_len <= 8 * abm::kPlaneBlock - 64 ? 8u : 0u));
  if (const char *e = experiment_env("ABM_COOP_WINDOWS")) if (e[0] == '0') a.G = 0;
  a.mode = mode;
  a.valid_frac = params->valid_frac;
  a.min_frag = params->min_frag; a.max_frag = params->max_frag;
  a.allow_ambig = params->allow_ambig;
  a.pairs = reinterpret_cast<abm::Hit *>(d_pair);
  a.se1 = reinterpret_cast<abm::Hit *>(d_se1); a.se2 = reinterpret_cast<abm::Hit *>(d_se2);
  a.cig1 = d_cig1; a.cig2 = d_cig2; a.cig_stride = cig_stride; a.cig_n1 = d_cig_n1; a.cig_n2 = d_cig_n2;
  a.ctmp_cap = eff_len + 2;
  const bool has_long = max_len > abm::kLdsReadLen;
  {
    // (two ends, and sim-like 150-base reads carry two or more indels a quarter of the time: 2 n ops overflowed, and an
    // overflowing arena means the whole batch is mapped AGAIN with a larger one -- with 8 contexts per GPU half of an
    // end-to-end run's batches were, profiles/r04_pe_e2e_variants_arena.log)
    const size_t want = std::max<size_t>(ctx->arena_want, std::max<size_t>(1u << 16, 4 * n));
    ctx->cig_arena_count.reserve(1);
    HIPCHK(hipMemsetAsync(ctx->cig_arena_count.p, 0, 4, st));
    a.cig_arena_count = ctx->cig_arena_count.p;
    if (ctx->host_results) {  // (abm_map_pe_batch: the arena lies in pinned host memory, like the rest of its results)
      ctx->h_arena.reserve(std::min<size_t>(want, 0xFFFFFF00u));
      ctx->h_tail.reserve(2);
      ctx->finished.reserve(1);
      ctx->h_tail.p[0] = ctx->h_tail.p[1] = 0;
      a.cig_arena = ctx->h_arena.p;
      a.cig_arena_cap = static_cast<abm::u32>(std::min<size_t>(ctx->h_arena.cap, 0xFFFFFF00u));
    }
    else {
      ctx->cig_arena.reserve(std::min<size_t>(want, 0xFFFFFF00u));
      a.cig_arena = ctx->cig_arena.p;
      a.cig_arena_cap = static_cast<abm::u32>(std::min<size_t>(ctx->cig_arena.cap, 0xFFFFFF00u));
    }
  }
  a.status = d_status;
  a.work = ctx->work.p;
  a.need_big = ctx->need_big.p;
  a.pair_diag = ctx->phase_stamps ? ctx->read_cycles : nullptr;
  a.pair_phases = ctx->phase_stamps ? ctx->pair_phases : nullptr;
  bool split = ctx->pe_split != 0;  // (0: tier 1 as ONE kernel per pair, as in rounds 1-4 -- same-box comparisons)
  if (const char *e = experiment_env("ABM_PE_SPLIT")) split = e[0] != '0';
  const size_t events_before = ctx->events_used;
  const size_t lds1 = abm::pe_lds_bytes(W, WB, a.GW, a.ctmp_cap, eff_len, size_frac, abm::kPeTier1Cap, false);
  if (!split) {
    // tier 1 unsplit: every pair, seeding and mating in one kernel, small sets in LDS
    a.cap = abm::kPeTier1Cap;
    int wps = abm::pe_waves_per_simd(lds1, ctx->phase_stamps, a.G != 0);
    if (const char *e = experiment_env("ABM_PE_WPS")) { if (!ctx->phase_stamps && a.G != 0 && (e[0] == '3' || e[0] == '4')) wps = e[0] - '0'; }
    const int waves = abm::pe_resident_waves(lds1, false, wps);
    if (waves <= 0) throw HipFail("map_pe_kernel (tier 1) does not fit on this device");
    ctx->payload1.reserve(static_cast<size_t>(waves) * a.cap);
    a.payload_ws = ctx->payload1.p;
    a.list_ws = nullptr;
    unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    a.next_read = counter;
    const hipEvent_t e1 = begin_timed(ctx, st);
    HIPCHK(abm::launch_map_pe(a, lds1, static_cast<abm::u32>(std::min<uint64_t>(n, waves)), false, ctx->phase_stamps, wps, st));
    if (e1) HIPCHK(hipEventRecord(e1, st));
  }
  else {
    // tier 1 split by phase.  SEED: both seed passes of every orientation call's two ends, shaped like the single-end
    // kernel (no alignment state, 6 KB of LDS per wave at 2x150); the finished lists go to the hand-over area.
    a.cap = abm::kPeTier1Cap;
    const abm::u32 n_slots = mode == 2 ? 8u : 4u;
    ctx->hand_hdr.reserve(n * n_slots * 2);
    ctx->hand_count.reserve(1);
    if (!ctx->split_stats.p) { ctx->split_stats.reserve(4); HIPCHK(hipMemsetAsync(ctx->split_stats.p, 0, 4 * sizeof(unsigned long long), st)); }
    abm::u32 scap = ctx->pe_scap ? ctx->pe_scap : abm::kPeTier1Cap;  // entries a list may grow to in the seed kernel (beyond kPeTier1Cap: in its staging area)
    if (const char *e = experiment_env("ABM_PE_SCAP")) scap = static_cast<abm::u32>(std::atoi(e));
    scap = std::min<abm::u32>(16384, std::max<abm::u32>(abm::kPeTier1Cap, scap));
    size_t per_pair = scap > abm::kPeTier1Cap ? 256 : 64;  // hand-over entries per pair (a list that finds no room sends its pair to the whole-pair kernel)
    if (const char *e = experiment_env("ABM_PE_HAND_PER_PAIR")) per_pair = static_cast<size_t>(std::max(4, std::atoi(e)));
    // (abm_ctx_set_pe_split's hand_entries, when given, is taken as it is: tests run the area out of room with it)
    const size_t hand_cap = ctx->hand_want ? std::max<size_t>(ctx->hand_want, 64) : std::min<size_t>(std::max<size_t>(n * per_pair, size_t(1) << 16), 0xFFFFFF00u);
    ctx->hand_pos.reserve(hand_cap);
    ctx->hand_d.reserve(hand_cap);
    a.hand_hdr = ctx->hand_hdr.p; a.hand_pos = ctx->hand_pos.p; a.hand_d = ctx->hand_d.p;
    a.hand_count = ctx->hand_count.p;
    a.hand_cap = static_cast<abm::u32>(std::min<size_t>(std::min(ctx->hand_pos.cap, ctx->hand_d.cap), 0xFFFFFF00u));
    a.split_stats = ctx->split_stats.p;
    HIPCHK(hipMemsetAsync(ctx->hand_count.p, 0, sizeof(unsigned long long), st));
    const size_t lds_s = abm::pe_seed_lds_bytes(W, WB, eff_len, a.cap);
    const int waves_s = abm::pe_seed_resident_waves(lds_s, a.G != 0);
    if (waves_s <= 0) throw HipFail("map_pe_kernel (seed) does not fit on this device");
    const abm::u32 grid_s = static_cast<abm::u32>(std::min<uint64_t>(n, waves_s));
    a.scap = scap;
    if (scap > a.cap) {
      ctx->stage_pos.reserve(static_cast<size_t>(grid_s) * scap);
      ctx->stage_d.reserve(static_cast<size_t>(grid_s) * scap);
      a.stage_pos = ctx->stage_pos.p; a.stage_d = ctx->stage_d.p;
    }
    a.payload_ws = nullptr;
    a.list_ws = nullptr;
    {
      unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
      HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
      a.next_read = counter;
      const hipEvent_t e1 = begin_timed(ctx, st);
      HIPCHK(abm::launch_pe_seed(a, lds_s, grid_s, ctx->phase_stamps, st));
      if (e1) HIPCHK(hipEventRecord(e1, st));
    }
    // MATE, small lists (every list of the pair within kPeTier1Cap entries: LDS): sort, scoring, mating, tracebacks,
    // best_single, fallback -- instruction-bound, it overlaps with the other contexts' seed kernels
    {
      const size_t lds_m = abm::pe_mate_lds_bytes(W, a.GW, a.ctmp_cap, eff_len, size_frac, a.cap, false);
      const int waves_m = abm::pe_mate_resident_waves(lds_m, false);
      if (waves_m <= 0) throw HipFail("map_pe_kernel (mate) does not fit on this device");
      a.order = nullptr;  // (in input order: the lists were handed over in whatever order the seed kernel finished them)
      unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
      HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
      a.next_read = counter;
      const hipEvent_t e1 = begin_timed(ctx, st);
      HIPCHK(abm::launch_pe_mate(a, lds_m, static_cast<abm::u32>(std::min<uint64_t>(n, waves_m)), false, ctx->phase_stamps, st));
      if (e1) HIPCHK(hipEventRecord(e1, st));
    }
  }
  // tier 2: lists and heaps of up to 32768 entries per wave in global memory
  {
    a.cap = abm::kPeCapLarge;
    a.order = nullptr;
    const size_t lds = abm::pe_lds_bytes(W, WB, a.GW, a.ctmp_cap, eff_len, size_frac, a.cap, true);
    int wps = abm::pe_waves_per_simd(lds, ctx->phase_stamps, a.G != 0);
    if (const char *e = experiment_env("ABM_PE_WPS2")) { if (!ctx->phase_stamps && a.G != 0 && (e[0] == '3' || e[0] == '4')) wps = e[0] - '0'; }
    int waves = abm::pe_resident_waves(lds, true, wps);
    if (waves <= 0) throw HipFail("map_pe_kernel (tier 2) does not fit on this device");
    const size_t lds_mb = abm::pe_mate_lds_bytes(W, a.GW, a.ctmp_cap, eff_len, size_frac, a.cap, true);
    int waves_mb = split ? abm::pe_mate_resident_waves(lds_mb, true) : 0;
    if (split && waves_mb <= 0) throw HipFail("map_pe_kernel (mate, tier 2) does not fit on this device");
    // (no more waves than the batch has pairs: every wave owns 2.3 MB of lists, heap and log in global memory -- 7.6 GB for
    // a full grid -- which a batch of a few thousand pairs, or the 32 contexts of two replicas on one device, must not ask for;
    // abm_ctx_reserve reserves for the batch size it is told)
    waves = static_cast<int>(std::min<uint64_t>(static_cast<uint64_t>(waves), std::max<uint64_t>(n, 64)));
    waves_mb = std::min(waves_mb, waves);  // (the two launches share the workspaces)
    pe_tier2_reserve(ctx, static_cast<size_t>(waves));
    a.log_ws = ctx->log2.p;
    a.heap_ws = ctx->heap2.p;
    a.payload_ws = ctx->payload2.p;
    a.list_ws = ctx->list2.p;
    a.work = ctx->work.p + 16;  // tier 2 tallies separately (abm_ctx_take_work_tiers)
    if (split) {
      // the pairs whose lists outgrew LDS inside the seed kernel's staging area: mated from global memory (nothing is seeded twice)
      ctx->subset_b.reserve(n); ctx->subset_count_b.reserve(1); ctx->class33_b.reserve(33);
      HIPCHK(abm::launch_collect_big(ctx->need_big.p, ctx->cls.p, n, abm::kRouteBig, ctx->class33_b.p, ctx->subset_b.p, ctx->subset_count_b.p, st));
      a.subset = ctx->subset_b.p; a.subset_count = ctx->subset_count_b.p;
      unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
      HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
      a.next_read = counter;
      const hipEvent_t e1 = begin_timed(ctx, st);
      HIPCHK(abm::launch_pe_mate(a, lds_mb, static_cast<abm::u32>(waves_mb), true, ctx->phase_stamps, st));
      if (e1) HIPCHK(hipEventRecord(e1, st));
      a.subset = ctx->subset.p; a.subset_count = ctx->subset_count.p;
    }
    // the pairs whose candidate sets outgrew tier 1 (the seed kernel): the whole pair again, one wave each, 32768-entry sets
    HIPCHK(abm::launch_collect_big(ctx->need_big.p, ctx->cls.p, n, abm::kRouteWhole, ctx->class33.p, ctx->subset.p, ctx->subset_count.p, st));
    unsigned long long *counter = ctx->next_read.p + (ctx->launch_seq++ & 63u);
    HIPCHK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    a.next_read = counter;
    if (ctx->host_results && !has_long) {  // the batch's last launch: its last wave publishes arena count and status to the host
      HIPCHK(hipMemsetAsync(ctx->finished.p, 0, 4, st));
      a.finished = ctx->finished.p;
      a.host_tail = ctx->h_tail.p;
    }
    const hipEvent_t e1 = begin_timed(ctx, st);
    HIPCHK(abm::launch_map_pe(a, lds, static_cast<abm::u32>(waves), true, ctx->phase_stamps, wps, st));
    if (e1) HIPCHK(hipEventRecord(e1, st));
    a.finished = nullptr;
    a.host_tail = nullptr;
  }
  ctx->pe_timed_launches = static_cast<uint32_t>(ctx->events_used - events_before);
  if (has_long) {
    pe_long_pairs(ctx, a, n, d_blob1, d_off1, d_blob2, d_off2, std::min<abm::u32>(max_len, abm::kMaxReadLen), params->valid_frac, st);
    if (ctx->host_results) {  // (rare: with a long-end launch the two summary words are copied out after it)
      HIPCHK(hipMemcpyAsync(&ctx->h_tail.p[0], a.cig_arena_count, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(hipMemcpyAsync(&ctx->h_tail.p[1], a.status, 4, hipMemcpyDeviceToHost, st));
    }
  }
  HIPCHK(hipEventRecord(ctx->last_done, st));
}

}  // namespace

namespace {
abm::DevIndex current_index(abm_ctx *ctx, abm::u32 maxc, bool single_end) {
  abm::DevIndex d = ctx->dix;
  d.max_candidates = maxc;
  { std::lock_guard<std::mutex> lk(ctx->ix->mu); d.direct_min = d.planes[0] != nullptr ? (single_end ? ctx->ix->direct_min_se : ctx->ix->direct_min) : 0u; }
  DeviceReplica &rep = *ctx->rep;
  std::lock_guard<std::mutex> lk(rep.mu);
  // (never rebuilt here: a rebuild waits for the whole device, frees and allocates tens of gigabytes and runs for
  // seconds -- inside an asynchronous entry point, and once per call for a caller that alternates two values.  Tables
  // follow abm_index_set_max_candidates at context creation, or abm_ctx_rebuild_seed_extension.)
  if (rep.dix.ext2 != nullptr && rep.dix.ext_maxc == maxc) {
    d.ext2 = rep.dix.ext2; d.ext3t = rep.dix.ext3t; d.ext3a = rep.dix.ext3a;
    d.e2 = rep.dix.e2; d.e3 = rep.dix.e3; d.ext_maxc = maxc;
  }
  else { d.ext2 = d.ext3t = d.ext3a = nullptr; d.e2 = d.e3 = d.ext_maxc = 0; }
  return d;
}
}  // namespace

extern "C" {

const char *abm_last_error(void) { return g_error.c_str(); }

void abm_default_params(abm_params *p) {
  if (!p) return;
  p->max_candidates = 0;
  p->valid_frac = 0.1;
  p->min_frag = 32;
  p->max_frag = 3000;
  p->allow_ambig = 0;
}

int abm_host_alloc(size_t bytes, void **out) {
  return guarded([&] {
    if (!out) throw std::invalid_argument("null argument");
    *out = nullptr;
    HIPCHK(hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocPortable));
  });
}
void abm_host_free(void *p) { if (p) (void)hipHostFree(p); }

int abm_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
// NUMA node of the host memory nearest to a device (sysfs, from the device's PCI address); -1 = unknown
int abm_device_numa_node(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, static_cast<int>(sizeof(bdf)), device) != hipSuccess) return -1;
  for (char *c = bdf; *c; ++c) *c = static_cast<char>(std::tolower(static_cast<unsigned char>(*c)));
  const std::string path = std::string("/sys/bus/pci/devices/") + bdf + "/numa_node";
  std::FILE *f = std::fopen(path.c_str(), "r");
  if (!f) return -1;
  int node = -1;
  if (std::fscanf(f, "%d", &node) != 1) node = -1;
  std::fclose(f);
  return node;
}
int abm_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  return guarded([&] {
    if (!free_bytes || !total_bytes) throw std::invalid_argument("null argument");
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
  });
}

// (what pe_device and abm_ctx_reserve allocate on the device for such batches: tier 2's per-wave workspaces, the packed
// encodings, blobs and offsets, orders and routes, the hand-over area, result slots for the device entry point)
int abm_ctx_pe_footprint(abm_ctx *ctx, uint64_t n, uint32_t max_len, uint64_t *bytes) {
  return guarded([&] {
    if (!ctx || !bytes) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    const uint32_t L = std::min<uint32_t>(std::max<uint32_t>(max_len, 48), abm::kLdsReadLen);
    const uint64_t W = words_for(L);
    const int waves = pe_tier2_waves(ctx, L, 0.1);
    const uint64_t w2 = std::min<uint64_t>(static_cast<uint64_t>(std::max(waves, 0)), std::max<uint64_t>(n, 64));
    const uint64_t cap = abm::kPeCapLarge;
    uint64_t b = w2 * (cap * 4 + 4 * cap * 4 + cap * 4 + (32 + 12 * cap) * 4);  // payload, lists, heap, log
    b += 2 * n * 4 * W * 8;                   // packed encodings of both ends
    b += 2 * (n * L + (n + 1) * 8);           // blobs and offsets
    b += n * (4 + 4 + 4 + 1 + 1 + 4 + 4);     // lens x 2, order, class, route, two pair lists
    b += n * 8 * 2 * 4 + n * 64 * 6;          // hand-over headers and entries
    b += 4 * n;                               // CIGAR arena (device entry point)
    *bytes = b + (uint64_t(64) << 20);
  });
}

uint32_t abm_max_read_length(void) { return abm::kMaxReadLen; }
uint64_t abm_ctx_reads_too_long(abm_ctx *ctx) { return ctx ? ctx->too_long : 0; }
uint32_t abm_ctx_window_records(const abm_ctx *ctx) { return ctx && ctx->dix.wrec != nullptr ? ctx->dix.wrec_max_len : 0u; }
int abm_ctx_filter_on_planes(const abm_ctx *ctx) { return ctx && ctx->dix.planes[0] != nullptr ? 1 : 0; }

int abm_index_open(const char *path, abm_index **out) {
  return guarded([&] {
    if (!path || !out) throw std::invalid_argument("null argument");
    auto *ix = new abm_index;
    try { ix->h.load(path); }
    catch (...) { delete ix; throw; }
    *out = ix;
  });
}
void abm_index_close(abm_index *ix) { delete ix; }
uint32_t abm_index_max_candidates(const abm_index *ix) { return ix->h.max_candidates; }
uint32_t abm_index_n_chroms(const abm_index *ix) { return static_cast<uint32_t>(ix->h.chrom_names.size()); }
const char *abm_index_chrom_name(const abm_index *ix, uint32_t i) {
  return i < ix->h.chrom_names.size() ? ix->h.chrom_names[i].c_str() : nullptr;
}
const uint32_t *abm_index_chrom_starts(const abm_index *ix) { return ix->h.chrom_starts.data(); }
uint64_t abm_index_bytes(const abm_index *ix) { return ix->h.device_bytes(); }

int abm_index_build(const char *fasta_path, const char *out_path, uint32_t n_threads) {
  return abm_index_build_targets(fasta_path, nullptr, out_path, n_threads);
}

int abm_index_build_targets(const char *fasta_path, const char *targets_path, const char *out_path, uint32_t n_threads) {
  return abm_index_build_opts(fasta_path, targets_path, 20, out_path, n_threads);
}

uint32_t abm_index_window(const abm_index *ix) { return ix->h.window; }

int abm_index_set_seed_extension(abm_index *ix, int letters2, int letters3) {
  return guarded([&] {
    if (!ix) throw std::invalid_argument("index is null");
    if (letters2 > 7 || letters3 > 4) throw std::invalid_argument("at most 7 and 4 letters");
    std::lock_guard<std::mutex> lk(ix->mu);
    for (auto &r : ix->replicas) if (r.second.refs) throw std::invalid_argument("set the seed extension before the first context is created");
    ix->want_e2 = letters2;
    ix->want_e3 = letters3;
  });
}

int abm_index_set_window_records(abm_index *ix, int max_read_len) {
  return guarded([&] {
    if (!ix) throw std::invalid_argument("index is null");
    std::lock_guard<std::mutex> lk(ix->mu);
    for (auto &r : ix->replicas) if (r.second.refs) throw std::invalid_argument("set the window records before the first context is created");
    ix->wrec_len = std::max(0, max_read_len);
  });
}

int abm_index_set_seed_extension_cap(abm_index *ix, int letters2, int letters3) {
  return guarded([&] {
    if (!ix) throw std::invalid_argument("index is null");
    if (letters2 < 0 || letters2 > 7 || letters3 < 0 || letters3 > 4) throw std::invalid_argument("seed-extension caps: 0..7 and 0..4 letters");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->cap_e2 = letters2;
    ix->cap_e3 = letters3;
  });
}

int abm_index_set_direct_narrowing(abm_index *ix, uint32_t min_entries) {
  return guarded([&] {
    if (!ix) throw std::invalid_argument("index is null");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->direct_min = ix->direct_min_se = min_entries;
  });
}

int abm_index_set_max_candidates(abm_index *ix, uint32_t max_candidates) {
  return guarded([&] {
    if (!ix) throw std::invalid_argument("index is null");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->want_maxc = max_candidates;
  });
}

int abm_ctx_rebuild_seed_extension(abm_ctx *ctx, uint32_t max_candidates) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("context is null");
    const abm::u32 maxc = max_candidates ? max_candidates : ctx->ix->h.max_candidates;
    DeviceReplica &rep = *ctx->rep;
    std::lock_guard<std::mutex> lk(rep.mu);
    if (rep.dix.ext_maxc == maxc) return;
    if (rep.refs != 1) throw std::runtime_error("seed-extension tables are rebuilt only while the context is the only one on its device");
    if (ctx->ix->h.multibit_genome) return;  // (such genomes get no tables)
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    build_ext(rep, *ctx->ix, maxc);
  });
}

int abm_ctx_seed_extension(const abm_ctx *ctx, uint32_t *letters2, uint32_t *letters3, uint64_t *bytes) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->rep->mu);
    const abm::DevIndex &d = ctx->rep->dix;
    const bool on = d.ext2 != nullptr;
    if (letters2) *letters2 = on ? d.e2 : 0;
    if (letters3) *letters3 = on ? d.e3 : 0;
    if (bytes) *bytes = on ? 8 * (abm::ext_keys(0, d.e2) + 2 * abm::ext_keys(1, d.e3)) : 0;
  });
}

int abm_index_build_opts(const char *fasta_path, const char *targets_path, uint32_t window, const char *out_path,
                         uint32_t n_threads) {
  return guarded([&] {
    if (!fasta_path || !out_path) throw std::invalid_argument("null argument");
    std::string text;
    std::vector<std::string> names;
    std::vector<uint32_t> starts;
    abm::load_fasta(fasta_path, text, names, starts);
    if (targets_path && targets_path[0]) abm::mask_outside_targets(targets_path, text, names, starts);
    abm::HostIndex h;
    abm::build_index(text, names, starts, n_threads ? n_threads : 1u, h, window);
    abm::write_index(h, out_path);
  });
}

int abm_ctx_create(const abm_index *ix, int device, abm_ctx **out) {
  return guarded([&] {
    if (!ix || !out) throw std::invalid_argument("null argument");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (n_dev <= 0) throw HipFail("no HIP device present: the mapping path has no CPU fallback");
    if (device < 0 || device >= n_dev) throw std::invalid_argument("device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    auto *c = new abm_ctx;
    try {
      c->device = device;
      c->ix = ix;
      {
        DeviceReplica *rep_p;
        abm::u32 tables_maxc;
        { std::lock_guard<std::mutex> lk(ix->mu); rep_p = &ix->replicas[device]; tables_maxc = ix->want_maxc ? ix->want_maxc : ix->h.max_candidates; }
        DeviceReplica &rep = *rep_p;
        std::lock_guard<std::mutex> lk(rep.mu);
        if (rep.refs == 0) {
          const abm::HostIndex &h = ix->h;
          // one arena, 256-byte aligned sub-arrays: genome first (u64), then the u32 tables
          auto up = [](size_t b) { return (b + 255) & ~static_cast<size_t>(255); };
          const size_t sz[7] = {h.genome.size() * 8,  h.counter.size() * 4, h.counter_t.size() * 4,
                                h.counter_a.size() * 4, h.index.size() * 4, h.index_t.size() * 4,
                                h.index_a.size() * 4};
          const void *src[7] = {h.genome.data(),    h.counter.data(), h.counter_t.data(), h.counter_a.data(),
                                h.index.data(),     h.index_t.data(), h.index_a.data()};
          size_t offs[11], total = 0;
          for (int k = 0; k < 7; ++k) { offs[k] = total; total += up(sz[k] + 64); }
          // the filter's bit-plane copies of the genome (DevIndex::planes), derived on the device
          const uint64_t n_bases = h.chrom_starts.empty() ? 0 : h.chrom_starts.back();
          const uint64_t n_blocks = h.multibit_genome ? 0 : (n_bases + abm::kPlaneBlock - 1) / abm::kPlaneBlock + 16;
          for (int k = 7; k < 9; ++k) { offs[k] = total; total += up(n_blocks * 16 + 128); }
          const uint64_t nmap_words = n_blocks ? ((n_bases >> abm::kPlaneChunkBits) + 64) / 32 + 1 : 0;
          offs[9] = total; total += up(nmap_words * 4 + 64);
          // the chromosome table (starts, name offsets, names) for the kernels that write SAM text themselves
          const size_t n_chr = h.chrom_names.size();
          std::vector<abm::u32> name_off(n_chr + 1, 0);
          std::string names_cat;
          for (size_t k = 0; k < n_chr; ++k) { name_off[k] = static_cast<abm::u32>(names_cat.size()); names_cat += h.chrom_names[k]; }
          name_off[n_chr] = static_cast<abm::u32>(names_cat.size());
          offs[10] = total; total += up((n_chr + 1) * 8 + names_cat.size() + 64);
          void *arena = nullptr;
          HIPCHK(hipMalloc(&arena, total));
          try {
            HIPCHK(hipMemset(arena, 0, total));
            char *base = static_cast<char *>(arena);
            for (int k = 0; k < 7; ++k)
              if (sz[k]) HIPCHK(hipMemcpy(base + offs[k], src[k], sz[k], hipMemcpyHostToDevice));
            rep.dix.genome = reinterpret_cast<const abm::u64 *>(base + offs[0]);
            rep.dix.counter = reinterpret_cast<const abm::u32 *>(base + offs[1]);
            rep.dix.counter_t = reinterpret_cast<const abm::u32 *>(base + offs[2]);
            rep.dix.counter_a = reinterpret_cast<const abm::u32 *>(base + offs[3]);
            rep.dix.index = reinterpret_cast<const abm::u32 *>(base + offs[4]);
            rep.dix.index_t = reinterpret_cast<const abm::u32 *>(base + offs[5]);
            rep.dix.index_a = reinterpret_cast<const abm::u32 *>(base + offs[6]);
            rep.dix.max_candidates = h.max_candidates;
            rep.dix.window = h.window;
            rep.dix.min_len = abm::kKeyWeight + h.window - 1;
            rep.dix.planes[0] = rep.dix.planes[1] = nullptr;
            rep.dix.nmap = nullptr;
            {
              char *ct = base + offs[10];
              HIPCHK(hipMemcpy(ct, h.chrom_starts.data(), (n_chr + 1) * 4, hipMemcpyHostToDevice));
              HIPCHK(hipMemcpy(ct + (n_chr + 1) * 4, name_off.data(), (n_chr + 1) * 4, hipMemcpyHostToDevice));
              if (!names_cat.empty()) HIPCHK(hipMemcpy(ct + (n_chr + 1) * 8, names_cat.data(), names_cat.size(), hipMemcpyHostToDevice));
              rep.dix.chrom_starts = reinterpret_cast<const abm::u32 *>(ct);
              rep.dix.chrom_name_off = reinterpret_cast<const abm::u32 *>(ct + (n_chr + 1) * 4);
              rep.dix.chrom_names = ct + (n_chr + 1) * 8;
              rep.dix.n_chroms = static_cast<abm::u32>(n_chr);
            }
            if (n_blocks) {
              auto *p0 = reinterpret_cast<abm::u64 *>(base + offs[7]), *p1 = reinterpret_cast<abm::u64 *>(base + offs[8] + 64);
              abm::u32 *d_bad = reinterpret_cast<abm::u32 *>(base + offs[8]);  // (the first 64 bytes of copy 1's array are free)
              auto *nmap = reinterpret_cast<abm::u32 *>(base + offs[9]);
              HIPCHK(abm::launch_make_planes(rep.dix.genome, h.genome.size(), n_bases, n_blocks, p0, p1, nmap, d_bad, nullptr));
              abm::u32 bad = 0;
              HIPCHK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
              if (!bad) { rep.dix.planes[0] = p0; rep.dix.planes[1] = p1; rep.dix.nmap = nmap; rep.plane_blocks = n_blocks; }
              if (const char *e = experiment_env("ABM_PLANES_COPIES")) if (e[0] == '1') rep.dix.planes[1] = p0;
            }
          }
          catch (...) { (void)hipFree(arena); throw; }
          rep.arena = arena;
          rep.dix.ext2 = rep.dix.ext3t = rep.dix.ext3a = nullptr;
          rep.dix.e2 = rep.dix.e3 = rep.dix.ext_maxc = 0;
          rep.dix.wrec = nullptr;
          rep.dix.wrec_t0 = rep.dix.wrec_a0 = rep.dix.wrec_blocks = rep.dix.wrec_back = rep.dix.wrec_max_len = 0;
          try { build_wrec(rep, *ix); build_ext(rep, *ix, tables_maxc); }
          catch (...) { free_wrec(rep); (void)hipFree(arena); rep.arena = nullptr; throw; }
        }
        ++rep.refs;
        c->holds_replica = true;
        c->rep = &rep;
        c->dix = rep.dix;
        c->kernel_turn = &rep.kernel_turn;
      }
      HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming));
      HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&c->drained), sizeof(abm::u32), hipHostMallocMapped | hipHostMallocCoherent));
      *c->drained = 0;
      c->work.reserve(32);
      HIPCHK(hipMemset(c->work.p, 0, 32 * sizeof(unsigned long long)));
    }
    catch (...) { abm_ctx_destroy(c); throw; }
    *out = c;
  });
}

void abm_ctx_destroy(abm_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->last_done) (void)hipEventDestroy(c->last_done);
  if (c->drained) (void)hipHostFree(c->drained);
  if (c->holds_replica && c->rep) {
    std::lock_guard<std::mutex> lk(c->rep->mu);
    if (--c->rep->refs == 0) {
      free_ext(*c->rep);
      free_wrec(*c->rep);
      (void)hipFree(c->rep->arena);
      c->rep->arena = nullptr;
    }
  }
  c->packed.release(); c->packed2.release(); c->lens2.release(); c->subset.release(); c->subset_count.release(); c->payload1.release(); c->payload2.release(); c->list2.release(); c->heap2.release(); c->log2.release(); c->need_big.release(); c->hand_hdr.release(); c->hand_pos.release(); c->hand_d.release(); c->hand_count.release(); c->split_stats.release(); c->stage_pos.release(); c->stage_d.release(); c->subset_b.release(); c->subset_count_b.release(); c->class33_b.release(); c->pe_out.release(); c->cig2h.release(); c->cig_n2h.release(); c->blob2.release(); c->off2.release(); c->coff.release(); c->scan_tmp.release(); c->cblob.release(); c->lens.release(); c->long_list.release(); c->long_count.release(); c->long_ctmp.release(); c->packed_long.release(); c->packed_long2.release(); c->long_q.release(); c->long_tb.release(); c->order.release(); c->class33.release(); c->cls.release(); c->work.release(); c->next_read.release(); c->cig_arena.release(); c->cig_arena_count.release(); c->h_cn.release(); c->h_slots.release(); c->h_arena.release(); c->h_cn2.release(); c->h_slots2.release(); c->h_rel.release(); c->h_rel2.release(); c->h_res.release(); c->h_pe_out.release(); c->h_sam.release(); c->h_sam_len.release(); c->h_tail.release(); c->finished.release(); c->blob.release(); c->off.release();
  c->res.release(); c->cig.release(); c->cig_n.release(); c->status.release();
  for (auto &e : c->events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete c;
}

// work tallies accumulated by every launch since the last read (reset on read):
// seed_iters, search probes, candidates, read words compared, set updates, alignments
int abm_ctx_set_phase_stamps(abm_ctx *ctx, int enable) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->phase_stamps = enable != 0;
  });
}

int abm_ctx_set_read_cycles(abm_ctx *ctx, uint32_t *d_read_cycles) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->read_cycles = d_read_cycles;
  });
}

int abm_ctx_set_pair_phases(abm_ctx *ctx, uint32_t *d_pair_phases) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->pair_phases = d_pair_phases;
  });
}

int abm_ctx_take_work(abm_ctx *ctx, uint64_t out[16]) {
  return guarded([&] {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long tmp[32];
    HIPCHK(hipMemcpy(tmp, ctx->work.p, sizeof(tmp), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(ctx->work.p, 0, sizeof(tmp)));
    for (int k = 0; k < 16; ++k) out[k] = tmp[k] + tmp[16 + k];
  });
}

int abm_ctx_take_work_tiers(abm_ctx *ctx, uint64_t out[32]) {
  return guarded([&] {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long tmp[32];
    HIPCHK(hipMemcpy(tmp, ctx->work.p, sizeof(tmp), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(ctx->work.p, 0, sizeof(tmp)));
    for (int k = 0; k < 32; ++k) out[k] = tmp[k];
  });
}

int abm_ctx_set_timing(abm_ctx *ctx, int enable) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->timing = enable != 0;
  });
}

int abm_ctx_take_kernel_time(abm_ctx *ctx, uint64_t *launches, double *total_ms) {
  return guarded([&] {
    if (!ctx || !launches || !total_ms) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    double sum = 0;
    for (size_t k = 0; k < ctx->events_used; ++k) {
      HIPCHK(hipEventSynchronize(ctx->events[k].second));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ctx->events[k].first, ctx->events[k].second));
      sum += ms;
    }
    *launches = ctx->events_used;
    *total_ms = sum;
    ctx->events_used = 0;
  });
}

int abm_ctx_take_kernel_times(abm_ctx *ctx, double *ms_out, uint64_t capacity, uint64_t *launches) {
  return guarded([&] {
    if (!ctx || !launches || (capacity && !ms_out)) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    for (size_t k = 0; k < ctx->events_used; ++k) {
      HIPCHK(hipEventSynchronize(ctx->events[k].second));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, ctx->events[k].first, ctx->events[k].second));
      if (k < capacity) ms_out[k] = ms;
    }
    *launches = ctx->events_used;
    ctx->events_used = 0;
  });
}

int abm_map_se_device(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                      const char *d_seq_blob, const uint64_t *d_seq_off, uint32_t max_len,
                      abm_hit *d_res, uint32_t *d_cig, uint32_t cig_stride, uint32_t *d_cig_n,
                      uint32_t *d_status, void *stream) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    se_device(ctx, mode, params, n, d_seq_blob, d_seq_off, max_len, d_res, d_cig, cig_stride, d_cig_n,
              d_status, static_cast<hipStream_t>(stream));
  });
}

int abm_map_se_batch(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                     const char *seq_blob, const uint64_t *seq_off, abm_hit *out_res,
                     uint32_t *out_cig_blob, uint64_t cig_capacity, uint64_t *out_cig_off) {
  return guarded([&] {
    if (!ctx || !seq_off || !out_res || !out_cig_off) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    out_cig_off[0] = 0;
    if (n == 0) return;
    HIPCHK(hipSetDevice(ctx->device));
    const hipStream_t st = ctx->stream;
    const uint64_t base = seq_off[0], bytes = seq_off[n] - base;
    // reads beyond the kernels' length cap are not mapped (the reference takes reads below 32767 bases); they come
    // back without a hit and are counted (abm_ctx_reads_too_long), the rest of the batch is unaffected
    HostTrace t0;
    const OffsetScan scan = scan_offsets(seq_off, n, ctx->h_rel);
    const uint32_t max_len = scan.max_len;
    ctx->too_long += scan.too_long;
    t0.mark("  offsets scanned");
    // one pass: upload, map, hits + compact CIGARs back
    auto run = [&](uint64_t m, const char *blob, uint64_t nbytes, const uint64_t *offs, uint32_t stride, bool take_turn) {
      ctx->blob.reserve(std::max<uint64_t>(nbytes, 1));
      ctx->off.reserve(m + 1);
      ctx->status.reserve(1);
      // The kernel writes hits, op counts and CIGAR slots (28 bytes per read) straight into pinned host memory:
      // device-to-host copies queued behind it are carried out by copy kernels, which get no compute unit while
      // another context's device-filling mapping kernel runs -- a finished batch's results used to sit on the
      // device for as long as the next batch's kernel ran (0.4-0.7 s in the CLI's timeline).
      ctx->h_res.reserve(m);
      ctx->h_cn.reserve(m);
      ctx->h_slots.reserve(m * stride);
      HostTrace t2;
      if (nbytes) HIPCHK(hipMemcpyAsync(ctx->blob.p, blob, nbytes, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(ctx->off.p, offs, (m + 1) * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemsetAsync(ctx->status.p, 0, 4, st));
      HIPCHK(hipStreamSynchronize(st));
      t2.mark("  H2D");
      uint32_t status = 0;
      {
        // the turn lasts until the kernel has handed out its last read; the few heavy reads still
        // running by then (hundreds of ms each on one wave) overlap with the next batch's kernel
        std::unique_lock<std::mutex> turn(*ctx->kernel_turn, std::defer_lock);
        if (take_turn) turn.lock();  // (the handful of reads of a long-CIGAR rerun just go ahead)
        __atomic_store_n(ctx->drained, 0u, __ATOMIC_RELAXED);
        ctx->signal_drained = true;
        ctx->host_results = true;
        try {
          se_device(ctx, mode, params, m, ctx->blob.p, ctx->off.p, max_len, ctx->h_res.p, ctx->h_slots.p, stride, ctx->h_cn.p,
                    ctx->status.p, st);
        }
        catch (...) { ctx->host_results = false; ctx->signal_drained = false; throw; }
        ctx->host_results = false;
        ctx->signal_drained = false;
        while (__atomic_load_n(ctx->drained, __ATOMIC_RELAXED) == 0u && hipStreamQuery(st) == hipErrorNotReady)
          std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (take_turn) turn.unlock();
        t2.mark("  map: drained");
        HIPCHK(hipStreamSynchronize(st));
        status = ctx->h_tail.p[1];  // (written by the kernel's last wave, with the arena count)
        t2.mark("  map: tail");
      }
      if (status & ~static_cast<uint32_t>(ABM_STATUS_CIGAR_OVERFLOW | ABM_STATUS_READ_TOO_LONG))
        throw std::runtime_error("kernel reported status " + std::to_string(status));
      return status;
    };
    // four ops per slot cover nearly every read; longer CIGARs come back through the arena.  If the arena
    // itself ran out (its default size is one op per read), the batch is mapped again with a larger one.
    const uint32_t stride = 4;
    HostTrace tr;
    for (;;) {
      const uint32_t status = run(n, seq_blob + base, bytes, scan.use, stride, true);
      tr.mark("upload+map");
      if (!(status & ABM_STATUS_CIGAR_OVERFLOW)) break;
      if (ctx->h_arena.cap >= 0xFFFFFF00u) throw std::runtime_error("CIGAR arena exhausted");
      ctx->arena_want = ctx->h_arena.cap * 4;
    }
    const uint32_t arena_n = ctx->h_tail.p[0];
    tr.mark("hits+cigars back");
    parallel_ranges(n, [&](uint64_t lo, uint64_t hi) { std::memcpy(out_res + lo, ctx->h_res.p + lo, (hi - lo) * sizeof(abm_hit)); });
    assemble_cigars(n, stride, ctx->h_cn.p, ctx->h_slots.p, ctx->h_arena.p, arena_n, out_cig_blob, cig_capacity, out_cig_off);
    tr.mark("cigars assembled");
  });
}

int abm_map_se_batch_sliced(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n, const char *seq_blob,
                            const uint64_t *seq_off, uint32_t n_slices, const uint64_t *slice_first,
                            abm_slice_done_fn done, void *user) {
  return guarded([&] {
    if (!ctx || !seq_off || !slice_first || !done || n_slices == 0) throw std::invalid_argument("null argument");
    if (slice_first[n_slices] != n) throw std::invalid_argument("slice_first must end at n");
    for (uint32_t s = 0; s < n_slices; ++s)
      if (slice_first[s] > slice_first[s + 1]) throw std::invalid_argument("slice_first not monotone");
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::vector<char> delivered(n_slices, 0);
    uint32_t n_delivered = 0;
    const uint32_t stride = 4;
    auto deliver = [&](uint32_t s) {
      delivered[s] = 1;
      ++n_delivered;
      done(user, s);
    };
    ctx->sliced_stride = stride;
    ctx->sliced_reads = n;
    if (n == 0) {
      for (uint32_t s = 0; s < n_slices; ++s) deliver(s);
      return;
    }
    HIPCHK(hipSetDevice(ctx->device));
    const hipStream_t st = ctx->stream;
    const uint64_t base = seq_off[0], bytes = seq_off[n] - base;
    HostTrace t0;
    const OffsetScan scan = scan_offsets(seq_off, n, ctx->h_rel);
    const uint32_t max_len = scan.max_len;
    ctx->too_long += scan.too_long;
    t0.mark("  offsets scanned");
    // slices complete while the kernel runs unless the batch holds reads of the long-read launch (which follows the
    // ordinary one) or more slices than a 16-bit slice number holds: then they are all handed over at the end
    const bool stream = max_len <= abm::kLdsReadLen && n_slices < 65000u && n < (1ull << 32);
    ctx->blob.reserve(std::max<uint64_t>(bytes, 1));
    ctx->off.reserve(n + 1);
    ctx->status.reserve(1);
    ctx->h_res.reserve(n);
    ctx->h_cn.reserve(n);
    ctx->h_slots.reserve(n * stride);
    if (stream) {
      ctx->h_slice_first.reserve(n_slices + 1);
      ctx->slice_first_d.reserve(n_slices + 1);
      ctx->h_slice_done.reserve(n_slices);
      for (uint32_t s = 0; s <= n_slices; ++s) ctx->h_slice_first.p[s] = static_cast<abm::u32>(slice_first[s]);
    }
    HostTrace t2;
    if (bytes) HIPCHK(hipMemcpyAsync(ctx->blob.p, seq_blob + base, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->off.p, scan.use, (n + 1) * 8, hipMemcpyHostToDevice, st));
    if (stream) HIPCHK(hipMemcpyAsync(ctx->slice_first_d.p, ctx->h_slice_first.p, (n_slices + 1) * 4ull, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));
    t2.mark("  H2D");
    for (;;) {
      HIPCHK(hipMemsetAsync(ctx->status.p, 0, 4, st));
      if (stream)  // (an empty slice has no read to complete it: it counts as done from the start)
        for (uint32_t s = 0; s < n_slices; ++s)
          __atomic_store_n(&ctx->h_slice_done.p[s], slice_first[s] == slice_first[s + 1] ? 1u : 0u, __ATOMIC_RELAXED);
      uint32_t status = 0;
      {
        std::unique_lock<std::mutex> turn(*ctx->kernel_turn);
        __atomic_store_n(ctx->drained, 0u, __ATOMIC_RELAXED);
        ctx->signal_drained = true;
        ctx->host_results = true;
        ctx->sliced_n = stream ? n_slices : 0;
        try {
          se_device(ctx, mode, params, n, ctx->blob.p, ctx->off.p, max_len, ctx->h_res.p, ctx->h_slots.p, stride, ctx->h_cn.p,
                    ctx->status.p, st);
        }
        catch (...) { ctx->host_results = false; ctx->signal_drained = false; ctx->sliced_n = 0; throw; }
        ctx->host_results = false;
        ctx->signal_drained = false;
        ctx->sliced_n = 0;
        // until the kernel is through: pass the turn on once it has handed out its last read, and hand every slice
        // whose completion word has arrived to the caller (slices complete roughly in order: the scan starts at the
        // first one still open)
        uint32_t first_open = 0;
        for (;;) {
          const bool running = hipStreamQuery(st) == hipErrorNotReady;
          // (the turn is passed on before any callback runs and looked at again before each one: a slow callback --
          // result copies, the caller's queue locks -- must not hold up the other context's kernel launch)
          auto pass_turn = [&](bool now) {
            if (turn.owns_lock() && (now || __atomic_load_n(ctx->drained, __ATOMIC_RELAXED) != 0u)) {
              turn.unlock();
              t2.mark("  map: drained");
            }
          };
          pass_turn(!running);
          bool any = false;
          if (stream) {
            while (first_open < n_slices && delivered[first_open]) ++first_open;
            for (uint32_t s = first_open; s < n_slices; ++s)
              if (!delivered[s] && __atomic_load_n(&ctx->h_slice_done.p[s], __ATOMIC_ACQUIRE) != 0u) { pass_turn(false); deliver(s); any = true; }
          }
          if (!running) break;
          if (!any) std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        HIPCHK(hipStreamSynchronize(st));
        status = ctx->h_tail.p[1];
        t2.mark("  map: tail");
      }
      if (status & ~static_cast<uint32_t>(ABM_STATUS_CIGAR_OVERFLOW | ABM_STATUS_READ_TOO_LONG))
        throw std::runtime_error("kernel reported status " + std::to_string(status));
      if (!(status & ABM_STATUS_CIGAR_OVERFLOW)) break;
      // some CIGAR found no room in the arena: its slice stayed open; map again with a larger arena (slices handed
      // over already were complete -- their results have been taken -- and are not handed over twice)
      if (ctx->h_arena.cap >= 0xFFFFFF00u) throw std::runtime_error("CIGAR arena exhausted");
      ctx->arena_want = ctx->h_arena.cap * 4;
    }
    for (uint32_t s = 0; s < n_slices; ++s)
      if (!delivered[s]) deliver(s);
  });
}

int abm_ctx_slice_results(abm_ctx *ctx, uint64_t lo, uint64_t hi, abm_hit *out_res, uint32_t *out_cig_blob,
                          uint64_t cig_capacity, uint64_t *out_cig_off) {
  // (called from inside abm_map_se_batch_sliced's callback, on its thread: the context is already locked)
  return guarded([&] {
    if (!ctx || !out_res || !out_cig_off) throw std::invalid_argument("null argument");
    if (lo > hi || hi > ctx->sliced_reads) throw std::invalid_argument("bad read range");
    const uint64_t m = hi - lo;
    const uint32_t stride = ctx->sliced_stride;
    const uint32_t *cn = ctx->h_cn.p + lo, *slots = ctx->h_slots.p + lo * stride;
    out_cig_off[0] = 0;
    for (uint64_t i = 0; i < m; ++i) out_cig_off[i + 1] = out_cig_off[i] + cn[i];
    if (out_cig_off[m] > cig_capacity || (out_cig_off[m] && !out_cig_blob)) throw std::length_error("cig_capacity too small");
    std::memcpy(out_res, ctx->h_res.p + lo, m * sizeof(abm_hit));
    const uint64_t arena_n = ctx->h_arena.cap;
    for (uint64_t i = 0; i < m; ++i) {
      const uint32_t k = cn[i];
      if (k == 0) continue;
      const uint32_t *src = slots + i * stride;
      if (k > stride) {
        if (static_cast<uint64_t>(src[0]) + k > arena_n) throw std::runtime_error("CIGAR arena reference out of range");
        src = ctx->h_arena.p + src[0];
      }
      std::memcpy(out_cig_blob + out_cig_off[i], src, k * 4ull);
    }
  });
}

int abm_ctx_set_sam_tails(abm_ctx *ctx, int enable, int allow_ambig) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->sam_on = enable != 0;
    ctx->sam_allow_ambig = allow_ambig != 0;
  });
}

int abm_ctx_slice_sam_tails(abm_ctx *ctx, uint64_t lo, uint64_t hi, const char **tails, uint32_t *stride, const uint32_t **lens) {
  return guarded([&] {
    if (!ctx || !tails || !stride || !lens) throw std::invalid_argument("null argument");
    if (lo > hi || hi > ctx->sliced_reads) throw std::invalid_argument("slice range outside the batch");
    // (no lock: called from inside the sliced entry point's callback, which holds it)
    if (ctx->sam_stride == 0) { *tails = nullptr; *lens = nullptr; *stride = 0; return; }
    *tails = ctx->h_sam.p + lo * ctx->sam_stride;
    *lens = ctx->h_sam_len.p + lo;
    *stride = ctx->sam_stride;
  });
}

int abm_ctx_set_pe_split(abm_ctx *ctx, int split, uint32_t seed_cap, uint64_t hand_entries) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    if (split < -1 || split > 1) throw std::invalid_argument("split: -1, 0 or 1");
    if (seed_cap > 16384) throw std::invalid_argument("seed_cap: at most 16384");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->pe_split = split;
    ctx->pe_scap = seed_cap;
    ctx->hand_want = static_cast<size_t>(std::min<uint64_t>(hand_entries, 0xFFFFFF00u));
  });
}

int abm_ctx_pe_split_stats(abm_ctx *ctx, uint64_t out[4]) {
  return guarded([&] {
    if (!ctx || !out) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    out[0] = out[1] = out[2] = out[3] = 0;
    if (ctx->split_stats.p) {
      unsigned long long tmp[4];
      HIPCHK(hipMemcpy(tmp, ctx->split_stats.p, sizeof(tmp), hipMemcpyDeviceToHost));
      HIPCHK(hipMemset(ctx->split_stats.p, 0, sizeof(tmp)));
      for (int k = 0; k < 3; ++k) out[k] = tmp[k];
    }
    if (ctx->hand_count.p) {
      unsigned long long cnt = 0;
      HIPCHK(hipMemcpy(&cnt, ctx->hand_count.p, sizeof(cnt), hipMemcpyDeviceToHost));
      out[3] = cnt;
    }
  });
}

uint32_t abm_ctx_pe_timed_launches(const abm_ctx *ctx) { return ctx ? ctx->pe_timed_launches : 0; }

uint64_t abm_ctx_pinned_bytes(const abm_ctx *c) {
  if (!c) return 0;
  uint64_t b = 0;
  b += (c->h_cn.cap + c->h_slots.cap + c->h_arena.cap + c->h_cn2.cap + c->h_slots2.cap + c->h_tail.cap + c->h_slice_first.cap + c->h_slice_done.cap) * 4;
  b += (c->h_rel.cap + c->h_rel2.cap + c->h_res.cap + c->h_pe_out.cap) * 8;
  return b;
}

int abm_map_pe_device(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                      const char *d_seq_blob1, const uint64_t *d_seq_off1, const char *d_seq_blob2,
                      const uint64_t *d_seq_off2, uint32_t max_len, abm_pair *d_pair,
                      abm_hit *d_se1, abm_hit *d_se2, uint32_t *d_cig1, uint32_t *d_cig2,
                      uint32_t cig_stride, uint32_t *d_cig_n1, uint32_t *d_cig_n2,
                      uint32_t *d_status, void *stream) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    std::lock_guard<std::mutex> lk(ctx->mu);
    pe_device(ctx, mode, params, n, d_seq_blob1, d_seq_off1, d_seq_blob2, d_seq_off2, max_len, d_pair, d_se1,
              d_se2, d_cig1, d_cig2, cig_stride, d_cig_n1, d_cig_n2, d_status, static_cast<hipStream_t>(stream));
  });
}

int abm_map_pe_batch(abm_ctx *ctx, int mode, const abm_params *params, uint64_t n,
                     const char *seq_blob1, const uint64_t *seq_off1, const char *seq_blob2,
                     const uint64_t *seq_off2, abm_pair *out_pair, abm_hit *out_se1,
                     abm_hit *out_se2, uint32_t *out_cig_blob1, uint64_t *out_cig_off1,
                     uint32_t *out_cig_blob2, uint64_t *out_cig_off2, uint64_t cig_capacity) {
  return guarded([&] {
    if (!ctx || !seq_off1 || !seq_off2 || !out_pair || !out_se1 || !out_se2 || !out_cig_off1 || !out_cig_off2)
      throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    out_cig_off1[0] = out_cig_off2[0] = 0;
    if (n == 0) return;
    HIPCHK(hipSetDevice(ctx->device));
    const hipStream_t st = ctx->stream;
    const OffsetScan scan1 = scan_offsets(seq_off1, n, ctx->h_rel), scan2 = scan_offsets(seq_off2, n, ctx->h_rel2);
    const uint32_t max_len = std::max(scan1.max_len, scan2.max_len);
    if (max_len > abm::kMaxReadLen) {  // (pairs with an end beyond what is mapped at all: counted once; rare, so a plain pass)
      for (uint64_t i = 0; i < n; ++i)
        ctx->too_long += (seq_off1[i + 1] - seq_off1[i]) > abm::kMaxReadLen || (seq_off2[i + 1] - seq_off2[i]) > abm::kMaxReadLen;
    }
    // Nothing is copied back after the kernels: they write pairs, fallback hits, op counts and CIGAR slots (and the
    // arena of longer CIGARs) straight into pinned host memory, and the last wave of the batch's last launch
    // publishes the two summary words there as well.  Device-to-host copies queued after a kernel are carried out
    // by copy kernels, which get no compute unit while other contexts' device-filling persistent kernels run -- with
    // eight paired-end batches in flight every batch's results sat on the device until the device had a gap: 8 M
    // pairs took 9.5 s end to end for 3.7 s of kernel work (profiles/r04_pe_e2e_variants_before.log).
    const uint32_t stride = kPeHostSlotOps;
    abm_pair *h_pair = nullptr;
    abm_hit *h_se1 = nullptr, *h_se2 = nullptr;
    auto run = [&](uint64_t m, const char *b1, uint64_t nb1, const uint64_t *o1, const char *b2, uint64_t nb2,
                   const uint64_t *o2) {
      ctx->blob.reserve(std::max<uint64_t>(nb1, 1));
      ctx->blob2.reserve(std::max<uint64_t>(nb2, 1));
      ctx->off.reserve(m + 1);
      ctx->off2.reserve(m + 1);
      ctx->h_pe_out.reserve(m * 5);  // 20 B pairs + 8 B + 8 B, in units of 8 B
      ctx->h_cn.reserve(m); ctx->h_cn2.reserve(m);
      ctx->h_slots.reserve(m * stride); ctx->h_slots2.reserve(m * stride);
      ctx->status.reserve(1);
      if (nb1) HIPCHK(hipMemcpyAsync(ctx->blob.p, b1, nb1, hipMemcpyHostToDevice, st));
      if (nb2) HIPCHK(hipMemcpyAsync(ctx->blob2.p, b2, nb2, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(ctx->off.p, o1, (m + 1) * 8, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(ctx->off2.p, o2, (m + 1) * 8, hipMemcpyHostToDevice, st));
      char *outb = reinterpret_cast<char *>(ctx->h_pe_out.p);
      h_pair = reinterpret_cast<abm_pair *>(outb);
      h_se1 = reinterpret_cast<abm_hit *>(outb + m * 20 + (8 - (m * 20) % 8) % 8);
      h_se2 = h_se1 + m;
      HIPCHK(hipMemsetAsync(ctx->status.p, 0, 4, st));
      // (every pair's counts are written by exactly one launch; zeroed all the same, like the device arrays were)
      std::memset(ctx->h_cn.p, 0, m * 4);
      std::memset(ctx->h_cn2.p, 0, m * 4);
      // (no kernel turn here: a paired-end batch ends in a long tail of a few pairs with huge
      // candidate sets, which another context's batch fills)
      ctx->host_results = true;
      try {
        pe_device(ctx, mode, params, m, ctx->blob.p, ctx->off.p, ctx->blob2.p, ctx->off2.p, max_len, h_pair, h_se1,
                  h_se2, ctx->h_slots.p, ctx->h_slots2.p, stride, ctx->h_cn.p, ctx->h_cn2.p, ctx->status.p, st);
      }
      catch (...) { ctx->host_results = false; throw; }
      ctx->host_results = false;
      HIPCHK(hipStreamSynchronize(st));
      const uint32_t status = ctx->h_tail.p[1];
      if (status & ~static_cast<uint32_t>(ABM_STATUS_CIGAR_OVERFLOW | ABM_STATUS_READ_TOO_LONG))
        throw std::runtime_error("kernel reported status " + std::to_string(status));
      return status;
    };
    for (;;) {
      const uint32_t status = run(n, seq_blob1 + seq_off1[0], seq_off1[n] - seq_off1[0], scan1.use, seq_blob2 + seq_off2[0],
                                  seq_off2[n] - seq_off2[0], scan2.use);
      if (!(status & ABM_STATUS_CIGAR_OVERFLOW)) break;
      if (ctx->h_arena.cap >= 0xFFFFFF00u) throw std::runtime_error("CIGAR arena exhausted");
      ctx->arena_want = ctx->h_arena.cap * 4;
    }
    const uint32_t arena_n = static_cast<uint32_t>(std::min<uint64_t>(ctx->h_tail.p[0], ctx->h_arena.cap));
    parallel_ranges(n, [&](uint64_t lo, uint64_t hi) {
      std::memcpy(out_pair + lo, h_pair + lo, (hi - lo) * sizeof(abm_pair));
      std::memcpy(out_se1 + lo, h_se1 + lo, (hi - lo) * sizeof(abm_hit));
      std::memcpy(out_se2 + lo, h_se2 + lo, (hi - lo) * sizeof(abm_hit));
    });
    assemble_cigars(n, stride, ctx->h_cn.p, ctx->h_slots.p, ctx->h_arena.p, arena_n, out_cig_blob1, cig_capacity, out_cig_off1);
    assemble_cigars(n, stride, ctx->h_cn2.p, ctx->h_slots2.p, ctx->h_arena.p, arena_n, out_cig_blob2, cig_capacity, out_cig_off2);
  });
}

// Sizes every workspace of the host entry points for batches of up to n reads (pairs) of up to max_len bases and
// maps a few dummy reads, so that the first real batch pays neither allocations (device arrays, pinned staging)
// nor the loading of the kernels' code object.  Set-up like the index upload; entirely optional.
int abm_ctx_reserve(abm_ctx *ctx, uint64_t n, uint32_t max_len, int paired) {
  return guarded([&] {
    if (!ctx) throw std::invalid_argument("ctx is null");
    if (n == 0) return;
    const uint32_t L = std::min<uint32_t>(std::max<uint32_t>(max_len, 48), abm::kLdsReadLen);
    {
      std::lock_guard<std::mutex> lk(ctx->mu);
      HIPCHK(hipSetDevice(ctx->device));
      const abm::u32 W = words_for(L);
      const uint32_t stride = 4;
      ctx->blob.reserve(n * L); ctx->off.reserve(n + 1);
      ctx->packed.reserve(n * 4 * W); ctx->lens.reserve(n); ctx->order.reserve(n); ctx->cls.reserve(n); ctx->class33.reserve(33);
      ctx->status.reserve(1); ctx->cig_arena_count.reserve(1);
      if (!paired) { ctx->cig.reserve(n * stride); ctx->cig_n.reserve(n); ctx->cig_arena.reserve(std::max<size_t>(1u << 16, 2 * n)); }
      ctx->h_cn.reserve(n); ctx->h_slots.reserve(n * (paired ? kPeHostSlotOps : stride));
      // (what se_device asks for: growing any buffer later frees the old one, and hipFree / hipHostFree wait for the whole
      // device -- i.e. for the other context's mapping kernel -- with the runtime's lock held)
      ctx->h_arena.reserve(std::max<size_t>({ctx->arena_want, size_t(1) << 16, static_cast<size_t>(paired ? 4 * n : 2 * n)}));
      if (!paired) {
        ctx->res.reserve(n); ctx->h_res.reserve(n);
        if (ctx->sam_on) { ctx->h_sam.reserve(static_cast<size_t>(n) * sam_stride_for(ctx, L, stride)); ctx->h_sam_len.reserve(n); }
        // (slices of at least 4096 reads; smaller ones make these buffers grow, which only tests do)
        const size_t ns = n / 4096 + 2;
        ctx->slice_id.reserve(n); ctx->slice_left.reserve(ns); ctx->slice_hist.reserve(abm::order_sliced_hist_words(static_cast<abm::u32>(ns)));
        ctx->slice_first_d.reserve(ns + 1); ctx->h_slice_first.reserve(ns + 1); ctx->h_slice_done.reserve(ns);
      }
      else {
        ctx->blob2.reserve(n * L); ctx->off2.reserve(n + 1);
        ctx->packed2.reserve(n * 4 * W); ctx->lens2.reserve(n);
        ctx->need_big.reserve(n); ctx->subset.reserve(n); ctx->subset_count.reserve(1);
        // (the phase split's hand-over area and second pair list, sized as pe_device sizes them for a batch of n pairs in the
        // random-PBAT mode's eight lists per pair: growing them mid-run frees the old ones, and hipFree waits for the whole
        // device -- the split's first end-to-end run lost half its rate to exactly that, profiles/r05_pe_e2e_regrowth.log)
        ctx->hand_hdr.reserve(n * 8 * 2); ctx->hand_count.reserve(1);
        {
          const abm::u32 scap = std::min<abm::u32>(16384, std::max<abm::u32>(abm::kPeTier1Cap, ctx->pe_scap ? ctx->pe_scap : abm::kPeTier1Cap));
          const size_t hand_cap = ctx->hand_want ? std::max<size_t>(ctx->hand_want, 64) : std::min<size_t>(std::max<size_t>(n * (scap > abm::kPeTier1Cap ? 256 : 64), size_t(1) << 16), 0xFFFFFF00u);
          ctx->hand_pos.reserve(hand_cap); ctx->hand_d.reserve(hand_cap);
        }
        ctx->subset_b.reserve(n); ctx->subset_count_b.reserve(1); ctx->class33_b.reserve(33);
        ctx->lens.reserve(n); ctx->order.reserve(n); ctx->cls.reserve(n); ctx->class33.reserve(33); ctx->next_read.reserve(64); ctx->work.reserve(32);
        ctx->h_pe_out.reserve(n * 5);
        {  // tier 2's workspaces for batches of n pairs (a full grid of waves from a few thousand pairs on)
          const int waves = pe_tier2_waves(ctx, L, 0.1);
          if (waves > 0) pe_tier2_reserve(ctx, static_cast<size_t>(std::min<uint64_t>(static_cast<uint64_t>(waves), std::max<uint64_t>(n, 64))));
        }
        ctx->h_slots.reserve(n * kPeHostSlotOps);
        ctx->h_cn2.reserve(n); ctx->h_slots2.reserve(n * kPeHostSlotOps);
      }
    }
    // a handful of reads through the real entry point: loads the code object, sizes the launch-shape caches
    const uint64_t m = 8;
    std::string blob(m * L, 'A');
    for (uint64_t i = 0; i < blob.size(); ++i) blob[i] = "ACGT"[(i * 7 + i / 3) & 3];
    std::vector<uint64_t> off(m + 1);
    for (uint64_t i = 0; i <= m; ++i) off[i] = i * L;
    std::vector<abm_hit> hits(m), h2(m);
    std::vector<abm_pair> pairs(m);
    std::vector<uint32_t> cig(m * (L + 2)), cig2(m * (L + 2));
    std::vector<uint64_t> co(m + 1), co2(m + 1);
    abm_params par;
    abm_default_params(&par);
    int rc;
    if (!paired) rc = abm_map_se_batch(ctx, ABM_SE_T_RICH, &par, m, blob.data(), off.data(), hits.data(), cig.data(), cig.size(), co.data());
    else rc = abm_map_pe_batch(ctx, ABM_PE_NORMAL, &par, m, blob.data(), off.data(), blob.data(), off.data(), pairs.data(), hits.data(),
                               h2.data(), cig.data(), co.data(), cig2.data(), co2.data(), cig.size());
    if (rc != 0) throw std::runtime_error(g_error);
    {  // the dummy reads are no work of the caller's: the tallies start from zero
      std::lock_guard<std::mutex> lk(ctx->mu);
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemset(ctx->work.p, 0, 32 * sizeof(unsigned long long)));
    }
  });
}

// CIGARs of the context's last device call that were longer than their slot (count > cig_stride): each lies
// whole in the arena, beginning at the index its slot's first word holds.
int abm_ctx_long_cigars(abm_ctx *ctx, uint32_t *out_ops, uint64_t capacity, uint64_t *n_ops) {
  return guarded([&] {
    if (!ctx || !n_ops) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipEventSynchronize(ctx->last_done));
    uint32_t cnt = 0;
    if (ctx->cig_arena_count.p) HIPCHK(hipMemcpy(&cnt, ctx->cig_arena_count.p, 4, hipMemcpyDeviceToHost));
    cnt = static_cast<uint32_t>(std::min<uint64_t>(cnt, ctx->cig_arena.cap));
    *n_ops = cnt;
    if (cnt > capacity) throw std::length_error("capacity too small for the long CIGARs");
    if (cnt && out_ops) HIPCHK(hipMemcpy(out_ops, ctx->cig_arena.p, cnt * 4ull, hipMemcpyDeviceToHost));
  });
}

int abm_stats_allreduce(abm_ctx *const *ctxs, int n_ctx, uint64_t *const *counters) {
  return guarded([&] {
    if (!ctxs || !counters || n_ctx <= 0) throw std::invalid_argument("bad arguments");
    if (n_ctx == 1) return;  // a single GPU already holds the total
    // One communicator, stream and 18 x u64 buffer per participating GPU of this process, created on the first call
    // for a set of devices and kept for the process's lifetime (ncclCommInitAll costs hundreds of milliseconds; a
    // service that maps run after run pays it once).
    struct Group { std::vector<ncclComm_t> comms; std::vector<hipStream_t> streams; std::vector<unsigned long long *> bufs; };
    static std::mutex gmu;
    static std::map<std::vector<int>, Group> groups;
    std::vector<int> devs(n_ctx);
    for (int k = 0; k < n_ctx; ++k) { if (!ctxs[k]) throw std::invalid_argument("null context"); devs[k] = ctxs[k]->device; }
    {
      // Contexts that share a device (replicas of the sharding on one GPU: `abismal-amd map -devices 0,0`, tests on a
      // one-GPU box) cannot be two ranks of one communicator -- RCCL refuses a device listed twice -- so their counters
      // are summed here on the host; the collective is for distinct devices.
      std::vector<int> sorted(devs);
      std::sort(sorted.begin(), sorted.end());
      if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
        unsigned long long total[18] = {0};
        for (int k = 0; k < n_ctx; ++k) for (int j = 0; j < 18; ++j) total[j] += counters[k][j];
        for (int k = 0; k < n_ctx; ++k) for (int j = 0; j < 18; ++j) counters[k][j] = total[j];
        return;
      }
    }
    auto nccl_check = [](ncclResult_t r, const char *what) {
      if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
    };
    std::lock_guard<std::mutex> lk(gmu);
    auto it = groups.find(devs);
    if (it == groups.end()) {
      Group g;
      g.comms.resize(n_ctx);
      nccl_check(ncclCommInitAll(g.comms.data(), n_ctx, devs.data()), "ncclCommInitAll");
      g.streams.assign(n_ctx, nullptr);
      g.bufs.assign(n_ctx, nullptr);
      try {
        for (int k = 0; k < n_ctx; ++k) {
          HIPCHK(hipSetDevice(devs[k]));
          HIPCHK(hipStreamCreate(&g.streams[k]));
          HIPCHK(hipMalloc(&g.bufs[k], 18 * sizeof(unsigned long long)));
        }
      }
      catch (...) {
        for (int k = 0; k < n_ctx; ++k) { if (g.bufs[k]) (void)hipFree(g.bufs[k]); if (g.streams[k]) (void)hipStreamDestroy(g.streams[k]); ncclCommDestroy(g.comms[k]); }
        throw;
      }
      it = groups.emplace(devs, std::move(g)).first;
    }
    Group &g = it->second;
    for (int k = 0; k < n_ctx; ++k) {
      HIPCHK(hipSetDevice(devs[k]));
      HIPCHK(hipMemcpyAsync(g.bufs[k], counters[k], 18 * sizeof(unsigned long long), hipMemcpyHostToDevice, g.streams[k]));
    }
    nccl_check(ncclGroupStart(), "ncclGroupStart");
    for (int k = 0; k < n_ctx; ++k)
      nccl_check(ncclAllReduce(g.bufs[k], g.bufs[k], 18, ncclUint64, ncclSum, g.comms[k], g.streams[k]), "ncclAllReduce");
    nccl_check(ncclGroupEnd(), "ncclGroupEnd");
    for (int k = 0; k < n_ctx; ++k) {
      HIPCHK(hipSetDevice(devs[k]));
      HIPCHK(hipMemcpyAsync(counters[k], g.bufs[k], 18 * sizeof(unsigned long long), hipMemcpyDeviceToHost, g.streams[k]));
      HIPCHK(hipStreamSynchronize(g.streams[k]));
    }
  });
}

}  // extern "C"
