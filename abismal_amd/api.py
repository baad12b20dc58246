"""ctypes binding of include/abismal_amd.h (names and argument meaning as there)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

SE_T_RICH, SE_A_RICH, SE_RANDOM = 0, 1, 2
PE_NORMAL, PE_PBAT, PE_RANDOM = 0, 1, 2

HIT_DTYPE = np.dtype([("diffs", "<i2"), ("flags", "<u2"), ("pos", "<u4")])
PAIR_DTYPE = np.dtype([("aln_score", "<i2"), ("reserved", "<i2"), ("r1", HIT_DTYPE), ("r2", HIT_DTYPE)])

EXPORTED_SYMBOLS = [
    "abm_last_error", "abm_default_params", "abm_index_open", "abm_index_close",
    "abm_index_max_candidates", "abm_index_n_chroms", "abm_index_chrom_name",
    "abm_index_chrom_starts", "abm_index_bytes", "abm_index_build", "abm_index_build_targets", "abm_index_build_opts", "abm_index_window", "abm_ctx_create", "abm_ctx_reserve", "abm_ctx_destroy",
    "abm_map_se_batch", "abm_map_se_batch_sliced", "abm_ctx_slice_results", "abm_map_se_device", "abm_map_pe_batch", "abm_map_pe_device",
    "abm_max_read_length", "abm_ctx_reads_too_long", "abm_ctx_filter_on_planes", "abm_ctx_long_cigars", "abm_ctx_take_work", "abm_ctx_set_phase_stamps", "abm_ctx_set_read_cycles", "abm_ctx_set_timing", "abm_ctx_take_kernel_time", "abm_ctx_take_kernel_times", "abm_ctx_take_work_tiers", "abm_stats_allreduce",
    "abm_device_count", "abm_host_alloc", "abm_host_free", "abm_index_set_seed_extension", "abm_index_set_max_candidates", "abm_index_set_direct_narrowing", "abm_ctx_seed_extension", "abm_ctx_rebuild_seed_extension", "abm_device_numa_node",
    "abm_ctx_set_pe_split", "abm_ctx_pe_split_stats", "abm_ctx_pe_timed_launches", "abm_ctx_set_pair_phases", "abm_device_memory", "abm_ctx_pe_footprint", "abm_index_set_seed_extension_cap", "abm_ctx_pinned_bytes", "abm_ctx_set_sam_tails", "abm_ctx_slice_sam_tails",
    "abm_index_set_window_records", "abm_ctx_window_records",
]


class AbismalAmdError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("max_candidates", C.c_uint32), ("valid_frac", C.c_double), ("min_frag", C.c_uint32),
                ("max_frag", C.c_uint32), ("allow_ambig", C.c_int32)]

    def __init__(self, max_candidates=0, valid_frac=0.1, min_frag=32, max_frag=3000, allow_ambig=0):
        super().__init__(max_candidates, valid_frac, min_frag, max_frag, allow_ambig)


def lib_path() -> str:
    # (ABISMAL_AMD_LIB: another build of the library, for the A/B scripts under scripts/)
    return os.environ.get("ABISMAL_AMD_LIB") or os.path.join(_HERE, "libabismal_amd.so")


_lib = None


def load_library():
    """dlopen the product library; there is no fallback when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch-ROCm bundles its own HIP runtime; when both are going to live in one process the
        # first one loaded must be torch's (same SONAME), so bring it in first if it is installed
        import torch  # noqa: F401
    except Exception:
        pass
    p = lib_path()
    if not os.path.exists(p):
        raise AbismalAmdError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(p)
    lib.abm_last_error.restype = C.c_char_p
    lib.abm_index_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
    lib.abm_index_close.argtypes = [C.c_void_p]
    lib.abm_index_max_candidates.argtypes = [C.c_void_p]
    lib.abm_index_max_candidates.restype = C.c_uint32
    lib.abm_index_n_chroms.argtypes = [C.c_void_p]
    lib.abm_index_n_chroms.restype = C.c_uint32
    lib.abm_index_chrom_name.argtypes = [C.c_void_p, C.c_uint32]
    lib.abm_index_chrom_name.restype = C.c_char_p
    lib.abm_index_chrom_starts.argtypes = [C.c_void_p]
    lib.abm_index_chrom_starts.restype = C.POINTER(C.c_uint32)
    lib.abm_index_bytes.argtypes = [C.c_void_p]
    lib.abm_index_bytes.restype = C.c_uint64
    lib.abm_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    lib.abm_index_build_targets.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]
    lib.abm_index_build_opts.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
    lib.abm_index_window.argtypes = [C.c_void_p]
    if hasattr(lib, "abm_index_set_seed_extension"):  # (absent from older builds loaded through ABISMAL_AMD_LIB)
        lib.abm_index_set_seed_extension.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.abm_ctx_seed_extension.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.abm_index_window.restype = C.c_uint32
    lib.abm_ctx_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.abm_ctx_destroy.argtypes = [C.c_void_p]
    lib.abm_ctx_reserve.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]
    lib.abm_max_read_length.restype = C.c_uint32
    lib.abm_ctx_reads_too_long.argtypes = [C.c_void_p]
    lib.abm_ctx_reads_too_long.restype = C.c_uint64
    lib.abm_ctx_filter_on_planes.argtypes = [C.c_void_p]
    lib.abm_ctx_filter_on_planes.restype = C.c_int
    lib.abm_ctx_take_work.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.abm_ctx_long_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.abm_ctx_set_timing.argtypes = [C.c_void_p, C.c_int]
    lib.abm_ctx_set_phase_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.abm_ctx_set_read_cycles.argtypes = [C.c_void_p, C.c_void_p]
    lib.abm_ctx_take_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    lib.abm_ctx_take_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint64)]
    lib.abm_ctx_take_work_tiers.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    vp = C.c_void_p
    lib.abm_map_se_batch.argtypes = [vp, C.c_int, C.POINTER(Params), C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp]
    lib.abm_map_se_device.argtypes = [vp, C.c_int, C.POINTER(Params), C.c_uint64, vp, vp, C.c_uint32, vp, vp,
                                      C.c_uint32, vp, vp, vp]
    lib.abm_map_pe_batch.argtypes = [vp, C.c_int, C.POINTER(Params), C.c_uint64, vp, vp, vp, vp, vp, vp, vp,
                                     vp, vp, vp, vp, C.c_uint64]
    lib.abm_map_pe_device.argtypes = [vp, C.c_int, C.POINTER(Params), C.c_uint64, vp, vp, vp, vp, C.c_uint32,
                                      vp, vp, vp, vp, vp, C.c_uint32, vp, vp, vp, vp]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise AbismalAmdError(load_library().abm_last_error().decode(errors="replace"))


def blob_and_offsets(reads):
    """list[str|bytes] -> (uint8 blob, uint64 offsets[n+1]) as the C ABI takes them."""
    bs = [r if isinstance(r, bytes) else r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return blob, off


def index_build(fasta, out, threads=0, targets=None, window=20):
    """abm_index_build / abm_index_build_opts (`abismal idx [-A targets]`; window 12 = the reference's --enable-short)."""
    threads = threads or (os.cpu_count() or 1)
    if targets or window != 20:
        _check(load_library().abm_index_build_opts(os.fsencode(fasta), os.fsencode(targets or ""), window, os.fsencode(out), threads))
    else:
        _check(load_library().abm_index_build(os.fsencode(fasta), os.fsencode(out), threads))


# letters of the seed-extension tables that Index() asks for when its caller does not say: None = the library's
# choice from the index's size (none for small genomes).  The GPU test suite sets (2, 1) so that its small genomes
# run the table path of the seed passes, and names (0, 0) where it wants the bisection-only path.
DEFAULT_SEED_EXTENSION = None
# read length Index() asks window records for (abm_index_set_window_records) when its caller does not say: None = the
# library's default (no records).  The GPU test suite sets 172 -- the record-fed kernels are what an hg38-scale run of the
# CLI or of bench.py uses -- and names 0 where it wants the filter on the bit planes.
DEFAULT_WINDOW_RECORDS = None


class Index:
    """abm_index_open / abm_index_close."""

    def __init__(self, path, seed_extension=None, window_records=None):
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.abm_index_open(os.fsencode(path), C.byref(h)))
        self.handle = h
        ext = seed_extension if seed_extension is not None else DEFAULT_SEED_EXTENSION
        if ext is not None:
            _check(self._lib.abm_index_set_seed_extension(h, int(ext[0]), int(ext[1])))
        wrec = window_records if window_records is not None else DEFAULT_WINDOW_RECORDS
        if wrec is not None:
            self.set_window_records(wrec)
        n = self._lib.abm_index_n_chroms(h)
        self.chrom_names = [self._lib.abm_index_chrom_name(h, i).decode() for i in range(n)]
        st = self._lib.abm_index_chrom_starts(h)
        self.chrom_starts = np.array([st[i] for i in range(n + 1)], dtype=np.uint32)
        self.max_candidates = self._lib.abm_index_max_candidates(h)
        self.window = self._lib.abm_index_window(h)
        self.device_bytes = self._lib.abm_index_bytes(h)

    def set_window_records(self, max_read_len):
        """abm_index_set_window_records: the index carries its candidates' windows for reads up to this length (0: none)"""
        self._lib.abm_index_set_window_records.argtypes = [C.c_void_p, C.c_int]
        _check(self._lib.abm_index_set_window_records(self.handle, int(max_read_len)))

    def set_seed_extension_cap(self, letters2, letters3):
        """abm_index_set_seed_extension_cap: the automatic table depths, capped (before the first context is created)"""
        self._lib.abm_index_set_seed_extension_cap.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _check(self._lib.abm_index_set_seed_extension_cap(self.handle, int(letters2), int(letters3)))

    def set_direct_narrowing(self, min_entries):
        """abm_index_set_direct_narrowing: smallest range the paired-end calls narrow directly (0 = never)."""
        self._lib.abm_index_set_direct_narrowing.argtypes = [C.c_void_p, C.c_uint32]
        _check(self._lib.abm_index_set_direct_narrowing(self.handle, int(min_entries)))

    def close(self):
        if self.handle:
            self._lib.abm_index_close(self.handle)
            self.handle = None


class Context:
    """abm_ctx_create / abm_ctx_destroy + the batch entry points."""

    def __init__(self, index: Index, device: int = 0):
        self._lib = load_library()
        self.index = index
        h = C.c_void_p()
        _check(self._lib.abm_ctx_create(index.handle, device, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self._lib.abm_ctx_destroy(self.handle)
            self.handle = None

    def seed_extension(self):
        """(letters of the 2-letter table, letters of the 3-letter tables, bytes) resident on this context's device"""
        a, b, n = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(self._lib.abm_ctx_seed_extension(self.handle, C.byref(a), C.byref(b), C.byref(n)))
        return int(a.value), int(b.value), int(n.value)

    def rebuild_seed_extension(self, max_candidates=0):
        """abm_ctx_rebuild_seed_extension: the device's tables rebuilt for another max_candidates (set-up call)"""
        self._lib.abm_ctx_rebuild_seed_extension.argtypes = [C.c_void_p, C.c_uint32]
        _check(self._lib.abm_ctx_rebuild_seed_extension(self.handle, max_candidates))

    def map_se(self, reads, mode=SE_T_RICH, params=None):
        """abm_map_se_batch.  Returns (hits[HIT_DTYPE], cigar_blob[u32], cigar_off[u64])."""
        params = params or Params()
        blob, off = blob_and_offsets(reads)
        n = len(off) - 1
        res = np.zeros(n, dtype=HIT_DTYPE)
        max_len = int((off[1:] - off[:-1]).max()) if n else 0
        cap = max(1, n * (max_len + 2))
        cig = np.zeros(cap, dtype=np.uint32)
        cig_off = np.zeros(n + 1, dtype=np.uint64)
        _check(self._lib.abm_map_se_batch(self.handle, mode, C.byref(params), n, blob.ctypes.data,
                                          off.ctypes.data, res.ctypes.data, cig.ctypes.data, cap,
                                          cig_off.ctypes.data))
        return res, cig[: int(cig_off[-1])], cig_off

    def map_se_sliced(self, reads, slice_first, mode=SE_T_RICH, params=None):
        """abm_map_se_batch_sliced + abm_ctx_slice_results: the batch's results taken slice by slice as the kernel
        completes them.  slice_first: n_slices + 1 read indices (the last one len(reads)).  Returns (hits, cigar_blob,
        cigar_off) laid out as map_se's for the reads from slice_first[0] on, and the order the slices arrived in."""
        params = params or Params()
        blob, off = blob_and_offsets(reads)
        n = len(off) - 1
        first = np.ascontiguousarray(slice_first, dtype=np.uint64)
        n_slices = len(first) - 1
        res = np.zeros(n, dtype=HIT_DTYPE)
        per_slice, arrived = {}, []
        errors = []

        def on_done(_user, s):
            try:
                lo, hi = int(first[s]), int(first[s + 1])
                m = hi - lo
                h = np.zeros(max(m, 1), dtype=HIT_DTYPE)
                co = np.zeros(m + 1, dtype=np.uint64)
                cg = np.zeros(max(1, 4 * m), dtype=np.uint32)
                rc = self._lib.abm_ctx_slice_results(self.handle, lo, hi, h.ctypes.data, cg.ctypes.data, len(cg), co.ctypes.data)
                if rc == -2:  # ABM_ERR_CAPACITY: the needed size is in co[m]
                    cg = np.zeros(int(co[m]), dtype=np.uint32)
                    rc = self._lib.abm_ctx_slice_results(self.handle, lo, hi, h.ctypes.data, cg.ctypes.data, len(cg), co.ctypes.data)
                _check(rc)
                if s in per_slice:
                    raise AbismalAmdError(f"slice {s} handed over twice")
                per_slice[s] = (h[:m].copy(), cg[: int(co[m])].copy(), co.copy())
                arrived.append(int(s))
            except Exception as e:  # (never let an exception cross the C frame)
                errors.append(e)

        cb_type = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)
        cb = cb_type(on_done)
        self._lib.abm_map_se_batch_sliced.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                      C.c_uint32, C.c_void_p, cb_type, C.c_void_p]
        self._lib.abm_ctx_slice_results.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        _check(self._lib.abm_map_se_batch_sliced(self.handle, mode, C.addressof(params), n, blob.ctypes.data, off.ctypes.data,
                                                 n_slices, first.ctypes.data, cb, None))
        if errors:
            raise errors[0]
        if sorted(arrived) != list(range(n_slices)):
            raise AbismalAmdError("not every slice was handed over")
        cig_parts, cig_off = [], np.zeros(n + 1, dtype=np.uint64)
        at = 0
        for s in range(n_slices):
            lo, hi = int(first[s]), int(first[s + 1])
            h, cg, co = per_slice[s]
            res[lo:hi] = h
            cig_off[lo:hi + 1] = co + at
            at += int(co[-1])
            cig_parts.append(cg)
        cig_off[: int(first[0])] = 0
        cig = np.concatenate(cig_parts) if cig_parts else np.zeros(0, dtype=np.uint32)
        return res, cig, cig_off, arrived

    def map_se_device(self, mode, params, n, d_blob, d_off, max_len, d_res, d_cig, cig_stride, d_cig_n,
                      d_status, stream=0):
        """abm_map_se_device: all d_* are integer device addresses (e.g. tensor.data_ptr())."""
        _check(self._lib.abm_map_se_device(self.handle, mode, C.byref(params), n, d_blob, d_off, max_len,
                                           d_res, d_cig, cig_stride, d_cig_n, d_status, stream))

    def map_pe(self, reads1, reads2, mode=PE_NORMAL, params=None):
        """abm_map_pe_batch.  Returns (pairs, se1, se2, (cig1, off1), (cig2, off2))."""
        params = params or Params()
        b1, o1 = blob_and_offsets(reads1)
        b2, o2 = blob_and_offsets(reads2)
        n = len(o1) - 1
        if len(o2) - 1 != n:
            raise ValueError("paired-end batch sizes differ")
        pairs = np.zeros(n, dtype=PAIR_DTYPE)
        se1 = np.zeros(n, dtype=HIT_DTYPE)
        se2 = np.zeros(n, dtype=HIT_DTYPE)
        ml = max(int((o1[1:] - o1[:-1]).max()) if n else 0, int((o2[1:] - o2[:-1]).max()) if n else 0)
        cap = max(1, n * (ml + 2))
        c1 = np.zeros(cap, dtype=np.uint32)
        c2 = np.zeros(cap, dtype=np.uint32)
        co1 = np.zeros(n + 1, dtype=np.uint64)
        co2 = np.zeros(n + 1, dtype=np.uint64)
        _check(self._lib.abm_map_pe_batch(self.handle, mode, C.byref(params), n, b1.ctypes.data, o1.ctypes.data,
                                          b2.ctypes.data, o2.ctypes.data, pairs.ctypes.data, se1.ctypes.data,
                                          se2.ctypes.data, c1.ctypes.data, co1.ctypes.data, c2.ctypes.data,
                                          co2.ctypes.data, cap))
        return pairs, se1, se2, (c1[: int(co1[-1])], co1), (c2[: int(co2[-1])], co2)

    def reads_too_long(self):
        return int(self._lib.abm_ctx_reads_too_long(self.handle))

    def window_records(self):
        """abm_ctx_window_records: longest read the context's window records serve (0: none)"""
        self._lib.abm_ctx_window_records.argtypes = [C.c_void_p]
        self._lib.abm_ctx_window_records.restype = C.c_uint32
        return int(self._lib.abm_ctx_window_records(self.handle))

    def filter_on_planes(self):
        return bool(self._lib.abm_ctx_filter_on_planes(self.handle))

    def long_cigars(self):
        """abm_ctx_long_cigars: the arena (u32 ops) of CIGARs longer than their slot from the last device call."""
        n = C.c_uint64()
        _check(self._lib.abm_ctx_long_cigars(self.handle, None, 1 << 40, C.byref(n)))
        out = np.zeros(max(1, int(n.value)), dtype=np.uint32)
        _check(self._lib.abm_ctx_long_cigars(self.handle, out.ctypes.data, len(out), C.byref(n)))
        return out[: int(n.value)]

    def set_timing(self, on=True):
        _check(self._lib.abm_ctx_set_timing(self.handle, int(on)))

    def take_kernel_time(self):
        """(launches, total_ms) of the mapping kernel since the last call (HIP events)."""
        n, ms = C.c_uint64(), C.c_double()
        _check(self._lib.abm_ctx_take_kernel_time(self.handle, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def take_kernel_times(self, capacity=256):
        """per-launch mapping-kernel durations (ms) since the last call, in launch order"""
        buf, n = (C.c_double * capacity)(), C.c_uint64()
        _check(self._lib.abm_ctx_take_kernel_times(self.handle, buf, capacity, C.byref(n)))
        return [float(buf[k]) for k in range(min(capacity, int(n.value)))]

    def take_work_tiers(self):
        """paired-end tallies per tier (see include/abismal_amd.h)"""
        out = (C.c_uint64 * 32)()
        _check(self._lib.abm_ctx_take_work_tiers(self.handle, out))
        keys = ["seed_offsets", "search_probes", "candidates", "index_run_lines", "set_updates", "alignments",
                "cyc_probe_narrow", "cyc_gather_hamming", "cyc_replay", "cyc_se_fallback", "cyc_total",
                "window_cache_hits", "cyc_sort_unique", "cyc_score_pairable", "cyc_mate", "cyc_best_single"]
        return [dict(zip(keys, [int(x) for x in out[16 * t:16 * t + 16]])) for t in range(2)]

    def take_work(self):
        out = (C.c_uint64 * 16)()
        _check(self._lib.abm_ctx_take_work(self.handle, out))
        keys = ["seed_offsets", "search_probes", "candidates", "read_words", "set_updates", "alignments"]
        d = dict(zip(keys, [int(x) for x in out[:6]]))
        d["window_cache_hits"] = int(out[11])
        d["single_job_reads"] = int(out[12])  # single-end reads whose set held one alignable entry (scored by the traceback run)
        if out[10]:  # diagnostic (stamped) kernel only
            d["light_filter_steps"], d["fifo_updates"], d["filter_steps"] = int(out[13]), int(out[14]), int(out[15])
        if out[10]:
            d["phase_cycles"] = dict(zip(["probe_narrow", "gather_hamming", "replay", "align", "total"],
                                         [int(x) for x in out[6:11]]))
        return d

    def set_pe_split(self, split=-1, seed_cap=0, hand_entries=0):
        """abm_ctx_set_pe_split: how paired-end batches are launched (seed / mate kernels, or one kernel per pair)"""
        self._lib.abm_ctx_set_pe_split.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64]
        _check(self._lib.abm_ctx_set_pe_split(self.handle, int(split), int(seed_cap), int(hand_entries)))

    def pe_split_stats(self):
        """abm_ctx_pe_split_stats: pairs by route since the last call, and the last batch's hand-over entries"""
        out = (C.c_uint64 * 4)()
        self._lib.abm_ctx_pe_split_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _check(self._lib.abm_ctx_pe_split_stats(self.handle, out))
        return {"mated_from_lds": int(out[0]), "mapped_whole": int(out[1]), "mated_from_device_memory": int(out[2]),
                "hand_over_entries_last_batch": int(out[3])}

    def pe_timed_launches(self):
        self._lib.abm_ctx_pe_timed_launches.argtypes = [C.c_void_p]
        self._lib.abm_ctx_pe_timed_launches.restype = C.c_uint32
        return int(self._lib.abm_ctx_pe_timed_launches(self.handle))

    def set_pair_phases(self, d_ptr):
        self._lib.abm_ctx_set_pair_phases.argtypes = [C.c_void_p, C.c_void_p]
        _check(self._lib.abm_ctx_set_pair_phases(self.handle, d_ptr))

    def set_read_cycles(self, d_ptr):
        _check(self._lib.abm_ctx_set_read_cycles(self.handle, d_ptr))

    def set_phase_stamps(self, on=True):
        _check(self._lib.abm_ctx_set_phase_stamps(self.handle, int(on)))
