#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native abismal mapping path.

Metric (BASELINE.json): mapped reads/sec, whole node, 100 bp single-end reads on
an hg38-scale index.  hg38 itself is not available offline, so the workload is a
synthetic hg38-*shaped* genome (24 chromosomes, interspersed repeat families,
low-complexity tracts, N gaps; size --genome-mbp, default 3100 Mbp) indexed by
the product's own `abm_index_build`, and reads drawn from it the way
`abismal sim` draws them (uniform position and strand, 1 % mutations split over
substitution/insertion/deletion, 98 % bisulfite conversion).

One "step" = one pass of the hot path (pack + map kernels, via
abm_map_se_device) over one batch of --reads reads already resident in HBM.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher in the environment this script starts the N ranks
itself: the parent only parses the arguments and spawns N fresh worker
processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment)
before it has made any GPU call, forwards rank 0's JSON line and exits with the
workers' status.  Under `python -m torch.distributed.run --nproc-per-node N`
the same workers run directly (WORLD_SIZE must then equal --gpus).  Reads shard
across ranks with the index replicated per GPU; the only collective is the
end-of-run RCCL all-reduce of the mapping statistics.
"""
import argparse
import os

# Paired-end steps of several (context, stream) slots overlap on the device (tier 2 ends in a few pairs that keep
# single waves busy for seconds); the HIP runtime multiplexes streams onto 4 hardware queues unless told otherwise,
# and must be told before it starts.  Measured at hg38 scale, 1 M pairs per step: 4 queues / 3 slots 1.8 M reads/s,
# 16 queues / 12 slots 3.0 M reads/s (profiles/r02_exp_pe_hw_queues.log); round 3's kernels: 16 slots 4.2-4.4 M against 4.1-4.2 M
# with 12 (profiles/r03_exp_pe_slots.log); round 5's: 24 slots 6.4-6.6 M against 6.2 M with 16 (profiles/r05_exp_pe_slots.log;
# paired runs build no seed-extension tables any more, so 24 contexts fit).  Single-end is unaffected.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HG38_FRACTIONS = [248, 242, 198, 190, 181, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59,
                  64, 47, 51, 156, 57]  # chr1..22, X, Y in Mbp


def physical_cores():
    """cores of the box (first SMT sibling of each), from sysfs; falls back to the thread count"""
    try:
        seen = set()
        base = "/sys/devices/system/cpu"
        for d in os.listdir(base):
            if d.startswith("cpu") and d[3:].isdigit():
                p = os.path.join(base, d, "topology", "thread_siblings_list")
                if os.path.exists(p):
                    seen.add(open(p).read().strip())
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_quota():
    """CPUs the container's CFS quota gives this process (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), or None: on a pod of
    a shared node, threads beyond it do not run more, they are throttled (abm_cli.cpp: CpuQuota)"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / p if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


# ----------------------------------------------------------------------------- genome
def synth_genome_fasta(path, total_mbp, seed, device):
    """hg38-shaped synthetic genome written as FASTA (generated with torch on the GPU)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    total = int(total_mbp * 1_000_000)
    sizes = [max(200_000, int(total * f / sum(HG38_FRACTIONS))) for f in HG38_FRACTIONS]
    lut = torch.tensor([ord(c) for c in "A" * 75 + "C" * 53 + "G" * 52 + "T" * 76], dtype=torch.uint8, device=device)
    acgt = torch.tensor([ord(c) for c in "ACGT"], dtype=torch.uint8, device=device)

    def rand_seq(n):
        return lut[torch.randint(0, 256, (n,), generator=g, device=device)]

    # interspersed repeat families, hg38-like proportions:
    # (consensus length, piece length, fraction of genome, per-copy divergence range)
    families = [(300, 300, 0.10, (0.05, 0.20)),    # Alu-like SINEs
                (6000, 900, 0.15, (0.05, 0.30)),   # L1-like LINE fragments
                (2000, 400, 0.04, (0.15, 0.35))]   # older elements
    cons = [rand_seq(c) for c, _, _, _ in families]
    sat_master = rand_seq(171)                     # alpha-satellite-like monomer
    with open(path, "wb") as f:
        for ci, n in enumerate(sizes):
            seq = rand_seq(n)
            for (clen, plen, frac, (dlo, dhi)), con in zip(families, cons):
                k = int(n * frac / plen)
                if k == 0 or n <= plen + 1:
                    continue
                at = torch.randint(0, n - plen, (k,), generator=g, device=device)
                src0 = torch.randint(0, clen - plen + 1, (k,), generator=g, device=device)
                ar = torch.arange(plen, device=device)
                piece = con[(src0[:, None] + ar[None, :])]
                div = dlo + (dhi - dlo) * torch.rand((k, 1), generator=g, device=device)
                mut = torch.rand((k, plen), generator=g, device=device) < div
                rnd = acgt[torch.randint(0, 4, (k, plen), generator=g, device=device)]
                piece = torch.where(mut, rnd, piece)
                seq[(at[:, None] + ar[None, :]).reshape(-1)] = piece.reshape(-1)
            # centromeric satellite arrays (~2 %): tandem 171-mers, array consensus 20 % off
            # the master, copies 2 % off their array's consensus
            sat_len = int(n * 0.02)
            if sat_len > 20000:
                n_arr = max(1, sat_len // 150000)
                alen = (sat_len // n_arr) // 171 * 171
                for a0 in torch.randint(0, n - alen, (n_arr,), generator=g, device=device).tolist():
                    m0 = torch.rand((171,), generator=g, device=device) < 0.20
                    acons = torch.where(m0, acgt[torch.randint(0, 4, (171,), generator=g, device=device)], sat_master)
                    arr = acons.repeat(alen // 171)
                    m1 = torch.rand((alen,), generator=g, device=device) < 0.02
                    arr = torch.where(m1, acgt[torch.randint(0, 4, (alen,), generator=g, device=device)], arr)
                    seq[a0:a0 + alen] = arr
            # microsatellites / low complexity (~1 %): motifs of 1-4 bases, tract length
            # geometric with mean ~30 bp (12..250)
            n_tr = max(1, int(n * 0.01 / 30))
            if n > 1000:
                at = torch.randint(0, n - 256, (n_tr,), generator=g, device=device)
                ml = torch.randint(1, 5, (n_tr,), generator=g, device=device)
                tl = (12 - 30.0 * torch.log(torch.rand((n_tr,), generator=g, device=device).clamp_min(1e-9))).clamp(12, 250).long()
                motif = lut[torch.randint(0, 256, (n_tr, 4), generator=g, device=device)]
                ar = torch.arange(256, device=device)
                tract = torch.gather(motif, 1, ar[None, :] % ml[:, None])
                keep = ar[None, :] < tl[:, None]
                dst = (at[:, None] + ar[None, :])[keep]
                seq[dst] = tract[keep]
            # one long N gap (centromere-like) and a short N run per chromosome
            if n > 4_000_000:
                seq[n // 3: n // 3 + min(1_000_000, n // 50)] = ord("N")
                seq[n // 2: n // 2 + 100] = ord("N")
            host = seq.cpu().numpy()
            f.write(f">chrS{ci + 1}\n".encode())
            w = 100
            full = (n // w) * w
            lines = np.empty((full // w, w + 1), dtype=np.uint8)
            lines[:, :w] = host[:full].reshape(-1, w)
            lines[:, w] = 10
            f.write(lines.tobytes())
            if full < n:
                f.write(host[full:].tobytes() + b"\n")


def read_index_genome(path):
    """Parse an AbismalIndex file far enough to get the chromosome table and the
    4-bit genome (layout: src/AbismalIndex.cpp:1037-1072)."""
    with open(path, "rb") as f:
        assert f.read(12) == b"AbismalIndex"
        f.read(12)
        (n_chroms,) = struct.unpack("<I", f.read(4))
        names = []
        for _ in range(n_chroms):
            (ln,) = struct.unpack("<I", f.read(4))
            names.append(f.read(ln).decode())
        starts = np.frombuffer(f.read(4 * (n_chroms + 1)), dtype=np.uint32)
        n_words = (int(starts[-1]) + 15) // 16
        genome = np.fromfile(f, dtype=np.uint64, count=n_words)
    return names, starts, genome


# ------------------------------------------------------------------------------ reads
def sample_pairs(genome_words, starts, n, L, seed, device, mut=0.01, bis=0.98, frag=(150, 500), chunk=500_000):
    """Paired-end sampler: read 1 = first L bases of a converted fragment, read 2 = first L bases of its
    reverse complement (src/simreads.cpp:113-133); substitutions only (indels are covered by the SE sampler)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gw = torch.from_numpy(genome_words.view(np.int64)).to(device)
    dec = torch.tensor([ord(c) for c in "NACNGNNNTNNNNNNN"], dtype=torch.uint8, device=device)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip("ACGTN", "TGCAN"):
        comp[ord(a)] = ord(b)
    acgt = torch.tensor([ord(c) for c in "ACGT"], dtype=torch.uint8, device=device)
    G = int(starts[-1])
    fmax = max(frag[1], L)
    r1 = torch.empty((n, L), dtype=torch.uint8, device=device)
    r2 = torch.empty((n, L), dtype=torch.uint8, device=device)
    ar = torch.arange(fmax, device=device)
    arL = torch.arange(L, device=device)
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        flen = torch.randint(max(frag[0], L), fmax + 1, (m,), generator=g, device=device)
        pos = torch.randint(0, G - fmax - 1, (m,), generator=g, device=device)
        for _ in range(8):
            idx = pos[:, None] + ar[None, :]
            nib = (gw[idx >> 4] >> ((idx & 15) << 2)) & 15
            bad = ((nib == 0) & (ar[None, :] < flen[:, None])).any(1)
            nb = int(bad.sum())
            if nb == 0:
                break
            pos[bad] = torch.randint(0, G - fmax - 1, (nb,), generator=g, device=device)
        fr = dec[nib]
        minus = torch.rand((m,), generator=g, device=device) < 0.5
        # fragment on the minus strand = reverse complement of the window [0, flen)
        ridx = (flen[:, None] - 1 - ar[None, :]).clamp(min=0)
        rc = comp[torch.gather(fr, 1, ridx).long()]
        fr = torch.where(minus[:, None], rc, fr)
        mutm = torch.rand((m, fmax), generator=g, device=device) < mut
        fr = torch.where(mutm, acgt[torch.randint(0, 4, (m, fmax), generator=g, device=device)], fr)
        conv = (fr == ord("C")) & (torch.rand((m, fmax), generator=g, device=device) < bis)
        fr = torch.where(conv, torch.full_like(fr, ord("T")), fr)
        r1[a:a + m] = fr[:, :L]
        tail = (flen[:, None] - 1 - arL[None, :]).clamp(min=0)
        r2[a:a + m] = comp[torch.gather(fr, 1, tail).long()]
    return r1.reshape(-1), r2.reshape(-1)


def sample_reads(genome_words, starts, n, L, seed, device, mut=0.01, bis=0.98, chunk=1_000_000, pbat_frac=0.0):
    """simreads-equivalent sampler on the GPU: returns uint8 blob [n*L] of ASCII reads.  pbat_frac of the reads come
    from the PBAT strand (A-rich: the reverse complement of a T-rich read), as `sim -R` mixes them (src/simreads.cpp)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    gw = torch.from_numpy(genome_words.view(np.int64)).to(device)
    dec = torch.tensor([ord(c) for c in "NACNGNNNTNNNNNNN"], dtype=torch.uint8, device=device)
    comp = torch.zeros(256, dtype=torch.uint8, device=device)
    for a, b in zip("ACGTN", "TGCAN"):
        comp[ord(a)] = ord(b)
    acgt = torch.tensor([ord(c) for c in "ACGT"], dtype=torch.uint8, device=device)
    G = int(starts[-1])
    span = L + 16
    out = torch.empty((n, L), dtype=torch.uint8, device=device)
    ar = torch.arange(span, device=device)
    for a in range(0, n, chunk):
        m = min(chunk, n - a)
        pos = torch.randint(0, G - span, (m,), generator=g, device=device)
        for _ in range(8):  # fragments touching an N run are redrawn (keeps the batch rectangular)
            idx = pos[:, None] + ar[None, :]
            nib = (gw[idx >> 4] >> ((idx & 15) << 2)) & 15
            bad = (nib == 0).any(1)
            nb = int(bad.sum())
            if nb == 0:
                break
            pos[bad] = torch.randint(0, G - span, (nb,), generator=g, device=device)
        frag = dec[nib]
        minus = torch.rand((m,), generator=g, device=device) < 0.5
        rc = comp[frag.flip(1).long()]
        frag = torch.where(minus[:, None], rc, frag)
        # mutations: step 1 normally, 2 after a deletion, 0 at an insertion
        u = torch.rand((m, L), generator=g, device=device)
        kind = torch.randint(0, 3, (m, L), generator=g, device=device)
        is_mut = u < mut
        step = torch.ones((m, L), dtype=torch.int64, device=device)
        step[is_mut & (kind == 1)] = 0
        step[is_mut & (kind == 2)] = 2
        src = (torch.cumsum(step, 1) - step).clamp_(0, span - 1)
        reads = torch.gather(frag, 1, src)
        rnd = acgt[torch.randint(0, 4, (m, L), generator=g, device=device)]
        reads = torch.where(is_mut & (kind != 2), rnd, reads)
        conv = (reads == ord("C")) & (torch.rand((m, L), generator=g, device=device) < bis)
        reads = torch.where(conv, torch.full_like(reads, ord("T")), reads)
        if pbat_frac > 0:
            pbat = torch.rand((m,), generator=g, device=device) < pbat_frac
            reads = torch.where(pbat[:, None], comp[reads.flip(1).long()], reads)
        out[a:a + m] = reads
    # ReadLoader rules (src/abismal.cpp:187-195): <44 informative bases -> skipped.
    # Such reads (N-gap overlaps) are kept as all-N records of length L so that the
    # batch stays rectangular; the mapper treats them exactly like the reference
    # treats a read it cannot seed (no hit).  Count them for the report.
    n_skipped = int(((out != ord("N")).sum(1) < 44).sum())
    return out.reshape(-1), n_skipped


def lookup_traffic(kind, genome_mbp, n, L):
    """L2->fabric read requests per launch (step) from a separate rocprofv3 --pmc pass over this same command
    (profiles/r*_traffic*.json, newest round first); reported only when the workload matches -- kind ("se_trich",
    "se_random", "pe"), genome size, reads (pairs) per step and read length."""
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None, None
    for tf in sorted((f for f in os.listdir(pdir) if "_traffic" in f and f.endswith(".json")), reverse=True):
        t = json.load(open(os.path.join(pdir, tf)))
        wl = t.get("workload", {})
        if (wl.get("kind", "se_trich"), wl.get("genome_mbp"), wl.get("reads"), wl.get("read_len")) == (kind, genome_mbp, n, L):
            return t["hbm_read_bytes_per_launch"], (
                f"profiles/{tf}: NOT measured in this run -- a separate rocprofv3 --pmc pass of the same command "
                f"(build {t.get('build', 'n/a')}); TCC_EA0_RDREQ_sum x 128 B = read requests the L2s sent to the "
                "fabric, Infinity-Cache hits included, so an upper bound on HBM bytes")
    return None, None


def run_pe(args, A, ctx, index, genome_words, starts, dev, world, rank, barrier):
    """Paired-end measurement (abm_map_pe_device); same timing protocol as the SE path."""
    import torch
    from abismal_amd.dist import reduce_stats, ranks_and_rates
    n, L = args.reads, args.read_len
    b1, b2 = sample_pairs(genome_words, starts, n, L, 2000 + rank, dev)
    off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    stride = 16
    params = A.Params()
    import ctypes as C
    lib = A.load_library()

    # Batches are independent, so consecutive steps go to alternating (context, stream) slots, as
    # the CLI's mapper threads do: the few pairs whose 32768-entry candidate sets keep one wave
    # busy long after the rest of a batch is done then overlap with the next batch.
    class Slot:
        def __init__(self, c, st):
            self.ctx, self.tstream = c, st
            self.stream = st.cuda_stream if st is not None else torch.cuda.current_stream().cuda_stream
            self.pairs = torch.zeros((n, 5), dtype=torch.int32, device=dev)
            self.se1 = torch.zeros((n, 2), dtype=torch.int32, device=dev)
            self.se2 = torch.zeros((n, 2), dtype=torch.int32, device=dev)
            self.c1 = torch.zeros((n, stride), dtype=torch.int32, device=dev)
            self.c2 = torch.zeros((n, stride), dtype=torch.int32, device=dev)
            self.n1 = torch.zeros((n,), dtype=torch.int32, device=dev)
            self.n2 = torch.zeros((n,), dtype=torch.int32, device=dev)
            self.status = torch.zeros((1,), dtype=torch.int32, device=dev)

    slots = [Slot(ctx, None)]
    for _ in range(1, max(1, args.streams)):
        slots.append(Slot(A.Context(index, dev.index or 0), torch.cuda.Stream(device=dev)))
    pairs, se1, se2, status = slots[0].pairs, slots[0].se1, slots[0].se2, slots[0].status
    issued = [0]

    def step(slot=None):
        z = slot or slots[issued[0] % len(slots)]
        issued[0] += 1
        rc = lib.abm_map_pe_device(z.ctx.handle, A.PE_NORMAL, C.byref(params), n, b1.data_ptr(), off.data_ptr(),
                                   b2.data_ptr(), off.data_ptr(), L, z.pairs.data_ptr(), z.se1.data_ptr(),
                                   z.se2.data_ptr(), z.c1.data_ptr(), z.c2.data_ptr(), stride, z.n1.data_ptr(),
                                   z.n2.data_ptr(), z.status.data_ptr(), z.stream)
        if rc != 0:
            raise RuntimeError(lib.abm_last_error().decode())

    for z in slots:  # every slot's workspaces are sized before the timed region
        step(z)
    for _ in range(max(0, args.warmup - len(slots))):
        step()
    torch.cuda.synchronize()
    for z in slots:
        z.ctx.take_work_tiers()
        z.ctx.set_timing(True)
    barrier()
    issued[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # per-launch durations (HIP events on the launch streams) of the timed steps, by kernel: a paired-end call launches
    # seed, mate (LDS lists), mate (lists in device memory) and whole-pair kernels -- or tier 1 and tier 2 when unsplit
    per_call = max(1, slots[0].ctx.pe_timed_launches())
    launch_names = ["seed", "mate_lds_lists", "mate_device_lists", "whole_pairs"] if per_call == 4 else ["tier1", "tier2"]
    launch_ms = [[] for _ in range(per_call)]
    split_routes = {}
    for z in slots:
        times = z.ctx.take_kernel_times(capacity=4096)
        z.ctx.set_timing(False)
        for k in range(per_call):
            launch_ms[k] += times[k::per_call]
        for k, v in z.ctx.pe_split_stats().items():
            split_routes[k] = (split_routes.get(k, 0) + v) if k != "hand_over_entries_last_batch" else max(split_routes.get(k, 0), v)
        z.ctx.take_work_tiers()
    tier_ms = [sum(launch_ms[:per_call // 2], []), sum(launch_ms[per_call // 2:], [])]
    # exact work tallies: the production kernels keep none (they cost the seed kernel registers), so ONE more step goes
    # through the diagnostic kernels on rank 0 -- tallies, per-launch durations alone on the device, phase shares
    tier_work = [dict(), dict()]
    diag = None
    if rank == 0:
        ctx.take_work_tiers()
        ctx.set_timing(True)
        ctx.set_phase_stamps(True)
        pd = torch.zeros((n,), dtype=torch.int32, device=dev)
        pph = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        ctx.set_read_cycles(pd.data_ptr())
        ctx.set_pair_phases(pph.data_ptr())
        step(slots[0])
        torch.cuda.synchronize()
        ctx.set_phase_stamps(False)
        ctx.set_read_cycles(None)
        ctx.set_pair_phases(None)
        pdh = pd.cpu().numpy().view(np.uint32)
        size, cyc = (pdh >> 16).astype(np.int64), (pdh & 0xFFFF).astype(np.int64) << 20
        hist = []
        pphh = pph.cpu().numpy().view(np.uint32).astype(np.int64) << 10
        phase_names = ["probe_narrow", "gather_hamming", "replay", "sort_unique", "score_pairable", "mate", "best_single", "se_fallback"]
        for lo_, hi_ in ((0, 129), (129, 257), (257, 1025), (1025, 4097), (4097, 16385), (16385, 65536)):
            sel = (size >= lo_) & (size < hi_)
            tot = max(1.0, float(pphh[sel].sum()))
            hist.append({"largest_set": f"{lo_}-{hi_ - 1}", "pairs": int(sel.sum()),
                         "gcycles": round(float(cyc[sel].sum()) / 1e9, 2),
                         "max_mcycles": round(float(cyc[sel].max()) / 1e6, 1) if sel.any() else 0,
                         "phase_share_of_the_mating_kernel": {nm: round(float(pphh[sel][:, k].sum()) / tot, 3) for k, nm in enumerate(phase_names)}})
        # the costliest pairs one by one: what a launch's tail is made of
        top = np.argsort(-cyc)[:8]
        slowest = [{"largest_set": int(size[i]), "mcycles": round(float(cyc[i]) / 1e6, 1),
                    "phase_mcycles": {nm: round(float(pphh[i, k]) / 1e6, 1) for k, nm in enumerate(phase_names)}} for i in top]
        times = ctx.take_kernel_times()
        ctx.set_timing(False)
        tiers = ctx.take_work_tiers()
        tier_work = [dict(t) for t in tiers]
        diag = {"kernel_ms": dict(zip(launch_names, [round(t, 2) for t in times])), "by_set_size": hist, "slowest_pairs": slowest, "tiers": []}
        for t in tiers:
            tot = max(1, t["cyc_total"])
            diag["tiers"].append({
                "counts": {k: t[k] for k in ("seed_offsets", "search_probes", "candidates", "set_updates", "alignments")},
                "share": {k[4:]: round(t[k] / tot, 3) for k in t if k.startswith("cyc_") and k != "cyc_total"}})
    concordant = pairs[:, 2] != 0
    fallback = (~concordant).unsqueeze(1) & torch.stack([se1[:, 1] != 0, se2[:, 1] != 0], 1)
    stats = torch.tensor([n, int(concordant.sum()), int(fallback.sum()), 0, 0, 0], dtype=torch.int64, device=dev)
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ranks_seen, rank_rates = ranks_and_rates(2 * n * args.steps, elapsed)
    reduce_stats(stats, t_el)
    if rank != 0:
        return
    elapsed = float(t_el.item())
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        from tests import oracle_binding as ob
        o = ob.load(build=not os.path.exists(ob.LIB))
        ns = min(n, max(1, args.cpu_sample // 4))
        h1 = [bytes(r) for r in b1[: ns * L].cpu().numpy().reshape(ns, L)]
        h2 = [bytes(r) for r in b2[: ns * L].cpu().numpy().reshape(ns, L)]
        oix = o.index_load(os.path.join(args.workdir, f"g{int(args.genome_mbp)}.idx"))
        quota = cpu_quota()
        cores = os.cpu_count() or 1
        if quota:
            cores = int(min(cores, max(1, round(4 * quota))))
        t0 = time.perf_counter()
        orc = o.map_pe(oix, h1, h2, mode=0, threads=cores)
        t_cpu = time.perf_counter() - t0
        o.index_free(oix)
        op, o_work = orc[0], orc[5]
        gp = pairs[:ns].cpu().numpy().view(np.uint32)
        same = int(((gp[:, 2] == op["r1"]["pos"]) & (gp[:, 4] == op["r2"]["pos"])).sum())
        # everything the boundary hands over -- pair score, both hits, both fallback hits, CIGARs -- for the
        # sample, through the host entry point (long CIGARs patched in), against the oracle
        from tests.test_gpu_pe_parity import compare_pe
        full = "identical"
        try:
            compare_pe(slots[0].ctx.map_pe(h1, h2, mode=A.PE_NORMAL), orc, "bench sample")
        except AssertionError as e:
            full = str(e)[:300]
        phys = physical_cores()
        cpu = {"value": round(2 * ns / t_cpu, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "physical_cores": phys, "cpu_quota_cpus": quota, "per_thread": round(2 * ns / t_cpu / cores, 1),
               "per_core": round(2 * ns / t_cpu / (min(phys, quota) if quota else phys), 1),
               "sample": f"first {ns} pairs, oracle restatement (-O3 -DNDEBUG), {cores} threads, {t_cpu:.1f}s",
               "pair_positions_identical_to_gpu": f"{same}/{ns}",
               "pairs_hits_fallbacks_cigars_vs_oracle": full}
        nr = max(1, o_work["reads"] // 2)
        strict = {k: o_work[k] / nr for k in ("seed_iters", "search_probes", "candidates", "words", "aligns", "aligns_tb")}
    # roofline (HBM-bound, like the single-end kernel): algorithmic bytes per PAIR by SURVEY 8(d)'s formula from the
    # kernels' own tallies (tier 1 + tier 2: a pair redone by tier 2 counts twice, as the work was done twice) and,
    # strictly, from the oracle's counters on the sample.  The tiers of the (context, stream) slots overlap in time,
    # so the rate is bytes of all timed steps / wall time of the timed region; per-launch durations are listed beside.
    bw_band = 2 * int(0.1 * L) + 1
    nw = (L + 15) // 16

    def pair_bytes(S, P, C, W, Aln, Cw=None):
        return 2 * L + S * 16 + P * 4.5 + C * 4 + (W + (C if Cw is None else Cw)) * 8 + Aln * (L + bw_band) / 2 + 36 + 16

    done_pairs = n * args.steps
    tally_pairs = n  # (the tallies are one diagnostic step's)
    tw = {k: tier_work[0].get(k, 0) + tier_work[1].get(k, 0) for k in tier_work[0]}
    fetched = tw["candidates"] - tw["window_cache_hits"]
    k_bytes = pair_bytes(tw["seed_offsets"] / tally_pairs, tw["search_probes"] / tally_pairs, tw["candidates"] / tally_pairs,
                         fetched * nw / tally_pairs, tw["alignments"] / tally_pairs, fetched / tally_pairs)
    s_bytes = None
    if cpu is not None:
        s_bytes = pair_bytes(strict["seed_iters"], strict["search_probes"], strict["candidates"], strict["words"],
                             strict["aligns"] + strict["aligns_tb"])
    use = s_bytes if s_bytes is not None else k_bytes
    achieved = use * done_pairs / elapsed / 1e9
    pe_traffic, pe_traffic_source = lookup_traffic("pe", int(args.genome_mbp), n, L)  # per step: tier 1 + tier 2 launches
    # 128-byte lines a pair asks for, by source, from the kernels' own tallies (both tiers; a pair redone by tier 2 counts
    # in both): what each item costs at line granularity -- a fetched candidate window is one line (two copies of the bit
    # planes keep it inside one), a seed offset two (its two seed-extension entries, or two counter pairs), a narrowing
    # probe two (index entry + genome letter), the index-entry runs what they span, an alignment its window
    # ((L + band) / 2 bytes at a random offset) -- against the counters' total when a PMC pass is on file
    per_aln = ((L + bw_band) / 2 + 127) // 128 + 1
    lines_by_source = {"candidate_windows": fetched / tally_pairs, "seed_offset_lookups": 2 * tw["seed_offsets"] / tally_pairs,
                       "narrowing_probes": 2 * tw["search_probes"] / tally_pairs, "index_entry_runs": tw["index_run_lines"] / tally_pairs,
                       "alignment_windows": per_aln * tw["alignments"] / tally_pairs, "reads_in_results_out": 2 * ((4 * nw * 8 + 127) // 128) + 2}
    lines_by_source = {k: round(v, 1) for k, v in lines_by_source.items()}
    lines_by_source["accounted"] = round(sum(lines_by_source.values()), 1)
    if pe_traffic:
        lines_by_source["counters_total"] = round(pe_traffic / 128 / n, 1)
        lines_by_source["not_accounted"] = round(lines_by_source["counters_total"] - lines_by_source["accounted"], 1)
        lines_by_source["not_accounted_is"] = ("by phase (counter passes with phases switched off, profiles/r04_pe_pmc_phases.log) 99 % of tier 1's requests are "
                                               "the seed passes', half of them the sensitive passes': lines the tallies count once and the L2s, turned over "
                                               "every few microseconds by the window traffic, fetch again; tier 2 adds its per-wave lists, heap, sort buffer and "
                                               "best_single log in global memory (DESIGN.md 4.3)")
    tier_lines = [{k: round(v / tally_pairs, 1) for k, v in (("candidate_windows", tier_work[t].get("candidates", 0) - tier_work[t].get("window_cache_hits", 0)),
                                                            ("seed_offset_lookups", 2 * tier_work[t].get("seed_offsets", 0)),
                                                            ("narrowing_probes", 2 * tier_work[t].get("search_probes", 0)),
                                                            ("index_entry_runs", tier_work[t].get("index_run_lines", 0)),
                                                            ("alignment_windows", per_aln * tier_work[t].get("alignments", 0)))} for t in range(2)]
    roofline = {"bound": "hbm", "kernel": "map_pe_kernel (tier 1 + tier 2)", "achieved": round(achieved, 2), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "basis": "strict" if s_bytes is not None else "kernel_tally",
                "alg_bytes_per_pair_strict": round(s_bytes, 1) if s_bytes is not None else None,
                "alg_bytes_per_pair_kernel_tally": round(k_bytes, 1),
                "frac_kernel_tally": round(k_bytes * done_pairs / elapsed / 1e9 / 8000.0, 5),
                "denominator": "wall time of the timed region (kernels of %d slots overlap)" % len(slots),
                "tier1_ms_per_launch": round(sum(tier_ms[0]) / max(1, len(launch_ms[0])), 2),
                "tier2_ms_per_launch": round(sum(tier_ms[1]) / max(1, len(launch_ms[-1])), 2),
                "ms_per_launch": {nm: round(sum(v) / max(1, len(v)), 2) for nm, v in zip(launch_names, launch_ms)},
                "pairs_by_route": split_routes,
                "lines_per_pair_by_source": lines_by_source, "lines_per_pair_by_source_tier1": tier_lines[0],
                "lines_per_pair_by_source_tier2": tier_lines[1],
                "traffic": pe_traffic, "traffic_source": pe_traffic_source,
                "traffic_over_algorithmic": round(pe_traffic / (use * n), 2) if pe_traffic else None,
                "work_per_pair": {k: round(v / tally_pairs, 2) for k, v in tw.items() if not k.startswith("cyc_")},
                "tier2_share_of_candidates": round(tier_work[1].get("candidates", 0) / max(1, tw["candidates"]), 3),
                "strict_counts_per_pair": {k: round(v, 2) for k, v in strict.items()} if cpu is not None else None}
    n_slots = len(slots)
    ext_pe = ctx.seed_extension()
    seed_tables_pe = {"letters_2": ext_pe[0], "letters_3": ext_pe[1], "gb": round(ext_pe[2] / 1e9, 2)}
    e2e = None
    if not args.no_e2e:
        # SURVEY 8(d)'s window for config 3: product sim (2 x L, fragments 150-500) -> two FASTQ files -> abismal-amd map ->
        # SAM on tmpfs; this process lets go of the GPU first (every slot's tier-2 workspaces are ~10 GB)
        kstat, n_pairs, n_conc, n_single = int(status.item()), int(stats[0]), int(stats[1]), int(stats[2])
        for z in slots[1:]:
            z.ctx.close()
        del slots, pairs, se1, se2, status, b1, b2
        ctx.close()
        index.close()
        torch.cuda.empty_cache()
        tag = f"g{int(args.genome_mbp)}"
        try:
            e2e = run_e2e(args, os.path.join(args.workdir, tag + ".idx"), os.path.join(args.workdir, tag + ".fa"), L, gpus=world, kind="pe")
        except Exception as exc:
            e2e = {"error": f"{type(exc).__name__}: {exc}"[:1500]}
        log(f"e2e (paired-end): {e2e}")
    else:
        kstat, n_pairs, n_conc, n_single = int(status.item()), int(stats[0]), int(stats[1]), int(stats[2])
    pe_value = 2 * n * args.steps * world / elapsed
    print(json.dumps({
        "metric": "mapped reads/sec (whole node), paired-end", "value": round(2 * n * args.steps * world / elapsed, 1),
        "unit": "reads/s", "n_gpus": world, "ranks_seen": ranks_seen, "per_rank_reads_per_s": rank_rates,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"synthetic hg38-shaped genome {args.genome_mbp:g} Mbp, {n} pairs x 2x{L} bp per GPU per step",
                   "streams": n_slots,
                   "end_to_end": ({"reads_per_s": e2e.get("value"), "over_kernel": round(e2e["value"] / pe_value, 4) if e2e.get("value") else None,
                                   "seconds_of_each_run": e2e.get("seconds_of_each_run"), "reads": e2e.get("reads"),
                                   "sam_body_md5_equals_oracle_cli": (e2e.get("parity") or {}).get("identical")} if isinstance(e2e, dict) else None)},
        "roofline": roofline, "cpu_baseline": cpu,
        "e2e_reads_per_s": e2e.get("value") if isinstance(e2e, dict) else None,
        "e2e_over_kernel": round(e2e["value"] / pe_value, 4) if isinstance(e2e, dict) and e2e.get("value") else None,
        "e2e": e2e, "kernel_status": kstat, "phase_stamps": diag, "seed_extension_tables": seed_tables_pe,
        "mapping": {"pairs": n_pairs, "concordant": n_conc, "ends_mapped_single": n_single}}), flush=True)



# ---------------------------------------------------------------------------------- e2e
def run_e2e(args, idx, fasta, L, gpus=1, kind="se"):
    """SURVEY.md section 8(d)'s window on `gpus` GPUs: FASTQ on disk -> SAM on disk through the product CLI
    (`abismal-amd map -gpus N`, a fresh child process), timed from the first batch submitted to the last SAM byte
    written (index load/upload excluded and reported).  The FASTQ comes from the product's own `sim` (md5-pinned
    restatement of `abismal sim`) on the bench genome with section 8(d)'s flags for the configuration -- kind "se":
    config 2/4 (100 bp single-end), "random": config 5 (`sim -single -R`, 150 bp, fragments 150-500; `map -R`), "pe":
    config 3 (2 x 150, fragments 150-500) -- --e2e-reads reads (pairs), N times over for N GPUs; the MEDIAN of three
    runs is the value.  With N > 1 the run writes N ordered part files (-out-parts N: one tmpfs file takes 6.5 GB/s =
    40 M reads/s of SAM text from any number of writers, profiles/r04_sink_probe.log).  A prefix of the input also goes
    through the oracle's CLI and the SAM bodies must be identical.  Single-end, config 2: the same input once more
    with -virtual-gpus N (no mapping call: every read gets a made-up hit) says what the host pipeline around the
    mapper can carry on this box, and a gzip-compressed copy of a prefix says what compressed input costs."""
    import hashlib
    import shutil
    import statistics
    import subprocess
    cli = os.path.join(ROOT, "abismal_amd", "abismal-amd")
    if not os.path.exists(fasta):  # (an earlier run without this leg removed it: the generator is deterministic)
        import torch
        synth_genome_fasta(fasta, args.genome_mbp, 1234, torch.device("cuda", 0))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else args.workdir
    wd = os.path.join(shm, f"abismal_e2e_{os.getpid()}")
    os.makedirs(wd, exist_ok=True)
    out = {}
    try:
        n = args.e2e_reads
        t0 = time.time()
        sim_flags = {"se": ["-single"], "random": ["-single", "-R", "-min-fraglen", "150", "-max-fraglen", "500"],
                     "pe": ["-min-fraglen", "150", "-max-fraglen", "500"]}[kind]
        map_flags = ["-R"] if kind == "random" else []
        subprocess.run([cli, "sim"] + sim_flags + ["-seed", "1", "-n", str(n), "-l", str(L), "-m", "0.01", "-b", "0.98",
                        "-o", os.path.join(wd, "reads"), fasta], check=True, stdout=subprocess.DEVNULL)
        t_sim = time.time() - t0
        fq1 = os.path.join(wd, "reads_1.fq")
        fqs1 = [fq1] + ([os.path.join(wd, "reads_2.fq")] if kind == "pe" else [])
        fq, copies = fq1, 1
        if gpus > 1 and kind != "pe":
            # N GPUs map N times the reads: the same FASTQ N times over (fewer if the tmpfs cannot hold input + SAM)
            per_copy = os.path.getsize(fq1) * 1.8
            room = shutil.disk_usage(wd).free * 0.8
            copies = int(max(1, min(gpus, room // per_copy)))
            if copies > 1:
                fq = os.path.join(wd, "reads_n.fq")
                with open(fq, "wb") as fo:
                    for _ in range(copies):
                        with open(fq1, "rb") as fi:
                            shutil.copyfileobj(fi, fo, 1 << 24)
        fqs = [fq] + fqs1[1:]
        if kind == "pe" and args.e2e_copies > 1:
            # a paired-end batch ends in a tail of a second or two that only other batches hide: the run is
            # timed on the pair of FASTQ files --e2e-copies times over (sim makes 150 k pairs a second on one thread)
            copies = args.e2e_copies
            fqs = []
            for k, src in enumerate(fqs1):
                dstp = os.path.join(wd, f"reads_x_{k + 1}.fq")
                with open(dstp, "wb") as fo:
                    for _ in range(copies):
                        with open(src, "rb") as fi:
                            shutil.copyfileobj(fi, fo, 1 << 24)
                fqs.append(dstp)
            fq = fqs[0]
        sam, tj = os.path.join(wd, "out.sam"), os.path.join(wd, "timing.json")
        gflag = ["-gpus", str(gpus)] + map_flags
        parts = gpus if gpus > 1 else 1
        pflag = ["-out-parts", str(parts)] if parts > 1 else []
        sam_files = [sam] if parts == 1 else [f"{sam}.part{k:03d}" for k in range(parts)]
        runs = []
        for rep in range(3):
            r = subprocess.run([cli, "map"] + gflag + pflag + ["-i", idx, "-o", sam, "-s", os.path.join(wd, "out.stats"), "-timing", tj] + fqs,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                return {"error": r.stderr[-2000:]}
            runs.append(json.load(open(tj)))
        secs = sorted(t["seconds"] for t in runs)
        med = [t for t in runs if t["seconds"] == secs[1]][0]
        # what the host side alone can carry around `gpus` GPUs: the same command with virtual GPUs (no device, no mapping
        # call), into the same kind of sink and into /dev/null, at several host-thread counts
        def host_ceiling_rows(src, vgpus, vparts, thread_counts=None, reps=3):
            """-virtual-gpus runs of `src` (no device, no mapping call): SAM into tmpfs (one file, or part files) and into /dev/null,
            at the run's default worker count and at -t 128 (both clamped to the container's CPU quota)"""
            rows = []
            vflag = ["-out-parts", str(vparts)] if vparts > 1 else []
            for sink in (os.path.join(wd, "ceil.sam"), "/dev/null"):
                for th in (thread_counts or sorted({med["host_threads"], min(os.cpu_count() or 1, 128)})):
                    got = []
                    for rep in range(reps):
                        r = subprocess.run([cli, "map", "-virtual-gpus", str(vgpus)] + vflag + ["-t", str(th), "-i", idx, "-o", sink, "-timing", tj, src],
                                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                        if r.returncode == 0:
                            got.append(json.load(open(tj)))
                        for f in [sink] + [f"{sink}.part{k:03d}" for k in range(vparts)]:
                            if f != "/dev/null" and os.path.exists(f):
                                os.remove(f)
                    if got:
                        t = sorted(got, key=lambda x: x["seconds"])[len(got) // 2]  # the median run (every run's time is listed)
                        rows.append({"virtual_gpus": vgpus, "statistic": f"median of {len(got)} runs", "sink": (f"{vparts} tmpfs part files" if vparts > 1 else "one tmpfs file") if sink != "/dev/null" else "/dev/null",
                                     "host_threads": t["host_threads"], "asked_threads": th, "reads": t["reads"],
                                     "reads_per_s": round(t["reads"] / t["seconds"], 1), "seconds_of_each_run": [round(x["seconds"], 3) for x in got],
                                     "busy_s": {k: round(v, 3) for k, v in t["busy_s"].items()}, "cpu_s": t.get("cpu_s"),
                                     "cpu_quota_cpus": t.get("cpu_quota_cpus"), "throttled_s": t.get("throttled_s")})
            return rows

        ceiling = []
        if kind == "se" and gpus > 1:
            ceiling = host_ceiling_rows(fq, gpus, parts)
        # compressed input (single-end, one GPU): a prefix of the FASTQ as one gzip member (what `gzip` writes: inflated by
        # one thread, as the reference's reader does, src/abismal.cpp:150-209) 
        gz = None
        if kind == "se" and gpus == 1 and args.e2e_gz_reads > 0:
            import zlib
            ngz = min(n, args.e2e_gz_reads)
            gzp = os.path.join(wd, "prefix.fq.gz")
            t0 = time.time()
            co = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(fq1, "rb") as fi, open(gzp, "wb") as fo:
                left = 4 * ngz
                for line in fi:
                    if left == 0:
                        break
                    fo.write(co.compress(line))
                    left -= 1
                fo.write(co.flush())
            t_gz = time.time() - t0
            r = subprocess.run([cli, "map", "-gpus", "1", "-i", idx, "-o", os.path.join(wd, "gz.sam"), "-timing", tj, gzp],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if r.returncode == 0:
                t = json.load(open(tj))
                gz = {"value": round(t["reads"] / t["seconds"], 1), "unit": "reads/s", "reads": t["reads"], "seconds": round(t["seconds"], 3),
                      "input": f"the first {ngz} reads as ONE gzip member (level 1, {os.path.getsize(gzp)} bytes, written in {t_gz:.0f}s)",
                      "note": "single-member gzip is inflated by one thread per file (zlib), as in the reference"}
            else:
                gz = {"error": r.stderr[-500:]}
            for f in (gzp, os.path.join(wd, "gz.sam")):
                if os.path.exists(f):
                    os.remove(f)
            # ... and the whole FASTQ as BGZF (what bgzip writes: independent blocks of up to 64 KB), which the host workers
            # inflate side by side
            import struct
            from concurrent.futures import ThreadPoolExecutor
            bzp = os.path.join(wd, "reads.fq.bgz")

            def bgzf_block(d):
                co = zlib.compressobj(1, zlib.DEFLATED, -15)
                z = co.compress(d) + co.flush()
                return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(z) + 25) + z +
                        struct.pack("<II", zlib.crc32(d), len(d)))

            t0 = time.time()
            with open(fq1, "rb") as fi, open(bzp, "wb") as fo, ThreadPoolExecutor(max_workers=16) as pool:
                while True:
                    big = fi.read(64 << 20)
                    if not big:
                        break
                    for blk in pool.map(bgzf_block, [big[k:k + 0xff00] for k in range(0, len(big), 0xff00)]):
                        fo.write(blk)
                fo.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
            t_bz = time.time() - t0
            bz_runs = []
            for rep in range(2):
                r = subprocess.run([cli, "map", "-gpus", "1", "-i", idx, "-o", os.path.join(wd, "bz.sam"), "-timing", tj, bzp],
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                if r.returncode != 0:
                    gz["bgzf"] = {"error": r.stderr[-500:]}
                    break
                bz_runs.append(json.load(open(tj)))
            if bz_runs:
                t = min(bz_runs, key=lambda x: x["seconds"])
                def body_bytes(path):  # file size without the header (whose @PG line holds the command line)
                    with open(path, "rb") as f:
                        head = 0
                        for line in f:
                            if not line.startswith(b"@"):
                                break
                            head += len(line)
                    return os.path.getsize(path) - head

                same = body_bytes(os.path.join(wd, "bz.sam")) == body_bytes(sam)
                gz["bgzf"] = {"value": round(t["reads"] / t["seconds"], 1), "unit": "reads/s", "reads": t["reads"], "seconds": round(t["seconds"], 3),
                              "seconds_of_each_run": [round(x["seconds"], 3) for x in bz_runs], "host_threads": t["host_threads"],
                              "input": f"the whole FASTQ as BGZF (level 1, {os.path.getsize(bzp)} bytes, written in {t_bz:.0f}s)",
                              "sam_body_size_equals_plain_run": same, "cpu_s": t.get("cpu_s"),
                              "note": "BGZF blocks are inflated by the host workers side by side (16 CPUs of quota on the measured box: zlib inflate is most of them)"}
            for f in (bzp, os.path.join(wd, "bz.sam")):
                if os.path.exists(f):
                    os.remove(f)
        # one GPU: the same pipeline on a longer input (the FASTQ four times over) -- a 10 M-read run is a few batches
        # long, so it mostly measures how well the first batch's start and the last batch's output are hidden
        sustained = None
        if args.e2e_copies > 1 and gpus == 1 and kind == "se":
            big = os.path.join(wd, "reads_x.fq")
            with open(big, "wb") as fo:
                for _ in range(args.e2e_copies):
                    with open(fq1, "rb") as fi:
                        shutil.copyfileobj(fi, fo, 1 << 24)
            r = subprocess.run([cli, "map", "-i", idx, "-o", os.path.join(wd, "big.sam"), "-timing", tj, big],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if r.returncode == 0:
                t = json.load(open(tj))
                sustained = {"value": round(t["reads"] / t["seconds"], 1), "unit": "reads/s", "reads": t["reads"],
                             "seconds": round(t["seconds"], 3), "input": f"the same FASTQ {args.e2e_copies} times over",
                             "batch_reads": t["batch_reads"]}
            # what the host side alone can carry: on the long input (a 10 M-read run of the pipeline alone lasts 0.2 s), around one
            # virtual GPU as this run has, and around eight with a part file each (what a node's run would ask of the host)
            ceiling = host_ceiling_rows(big, 1, 1) + host_ceiling_rows(big, 8, 8, thread_counts=[med["host_threads"]])
            for f in (big, os.path.join(wd, "big.sam")):
                if os.path.exists(f):
                    os.remove(f)
        out = {"value": round(med["reads"] / med["seconds"], 1), "unit": "reads/s", "gpus": gpus, "statistic": "median of three runs",
               "sustained": sustained, "gzip_input": gz,
               "window": "first batch submitted -> last SAM byte written (abismal-amd map -gpus N" + (f" -out-parts {parts}" if parts > 1 else "") +
                         (" -R" if kind == "random" else "") + ", plain FASTQ in, SAM text out, tmpfs)",
               "reads": med["reads"], "input": f"{n} product-sim " + ("pairs" if kind == "pe" else "reads") + (f", {copies} times over" if copies > 1 else "") +
                                               " (sim " + " ".join(sim_flags) + f" -l {L})",
               "seconds": round(med["seconds"], 3), "seconds_of_each_run": [round(t["seconds"], 3) for t in runs],
               "index_load_s": round(med["index_load_s"], 2), "host_prepare_s": round(med.get("host_prepare_s", 0.0), 2),
               "fastq_bytes": sum(os.path.getsize(f) for f in fqs), "sam_bytes": sum(os.path.getsize(f) for f in sam_files), "sim_s": round(t_sim, 1),
               "cli": {k: med[k] for k in ("gpus", "mappers_per_gpu", "host_threads", "numa_nodes", "pinned", "out_parts", "batch_reads", "batches_per_gpu", "reads_per_gpu") if k in med},
               "busy_s": {k: round(v, 3) for k, v in med["busy_s"].items()}, "cpu_s": med.get("cpu_s"),
               "host_ceiling": ceiling or None,
               "host_ceiling_reads_per_s": max([c["reads_per_s"] for c in ceiling if c["sink"] != "/dev/null" and c["virtual_gpus"] == gpus], default=None),
               "host_ceiling_reads_per_s_dev_null": max([c["reads_per_s"] for c in ceiling if c["sink"] == "/dev/null" and c["virtual_gpus"] == gpus], default=None),
               "host_ceiling_reads_per_s_8_virtual_gpus_8_parts": max([c["reads_per_s"] for c in ceiling if c["sink"] != "/dev/null" and c["virtual_gpus"] == 8], default=None),
               "host_ceiling_note": ("abismal-amd map -virtual-gpus N on the same input: count, cut, parse, deal, format and write at full rate, every "
                                     "read given a made-up hit instead of the mapping call; on the FASTQ --e2e-copies times over (40 M reads), SAM into one tmpfs file "
                                     "and into /dev/null at the default and at 128 host threads (both clamped to the container's CPU quota), and around 8 "
                                     "virtual GPUs with 8 part files; scripts/r04_host_ceiling.py sweeps 40 M-read runs (profiles/r04_host_ceiling.log)") if ceiling else None}
        # parity on a prefix: product CLI vs oracle CLI, SAM body (everything but the @PG line) byte for byte
        nchk = min(n, args.e2e_check)
        if nchk > 0:
            from tests import oracle_binding as ob
            if not os.path.exists(ob.CLI):
                ob.build_oracle()
            pfqs = []
            for k, src in enumerate(fqs):
                pfq = os.path.join(wd, f"prefix_{k + 1}.fq")
                with open(src, "rb") as fi, open(pfq, "wb") as fo:
                    for j, line in enumerate(fi):
                        if j >= 4 * nchk:
                            break
                        fo.write(line)
                pfqs.append(pfq)

            def body_md5(path, limit=None):
                h, k = hashlib.md5(), 0
                with open(path, "rb") as f:
                    for line in f:
                        if line.startswith(b"@PG"):
                            continue
                        if limit is not None and k >= limit:
                            break
                        h.update(line)
                        k += 1
                return h.hexdigest(), k

            subprocess.run([cli, "map"] + map_flags + ["-i", idx, "-o", os.path.join(wd, "p_gpu.sam")] + pfqs, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t0 = time.time()
            subprocess.run([ob.CLI, "map"] + map_flags + ["-t", str(int(min(os.cpu_count() or 1, max(1, round(4 * (cpu_quota() or 1e9)))))), "-i", idx, "-o", os.path.join(wd, "p_oracle.sam")] + pfqs,
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            t_or = time.time() - t0
            m_g, k_g = body_md5(os.path.join(wd, "p_gpu.sam"))
            m_o, k_o = body_md5(os.path.join(wd, "p_oracle.sam"))
            m_f, _ = body_md5(sam_files[0], limit=k_g)  # the full run starts with the very same records
            out["parity"] = {"prefix_" + ("pairs" if kind == "pe" else "reads"): nchk, "sam_lines": k_g, "md5_product": m_g, "md5_oracle_cli": m_o,
                             "md5_full_run_prefix": m_f, "identical": bool(m_g == m_o == m_f and k_g == k_o),
                             "oracle_cli_s": round(t_or, 1)}
    finally:
        shutil.rmtree(wd, ignore_errors=True)
    return out


# --------------------------------------------------------------------------- launcher
def launch_ranks(n_ranks):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: spawns one fresh worker
    process per GPU and waits.  Nothing here touches the GPU (no torch import, no HIP call): a
    process that has initialised the GPU must never be replaced or forked on this pool."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ABM_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[bench] rank {procs.index(p)} exited with {code}; stopping the others", file=sys.stderr, flush=True)
                    for q in live:
                        q.terminate()  # exact PIDs we started
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dist_dry_run(args, rank, world):
    """--dist-dry-run: the launcher, the rendezvous and the statistics collective with the gloo backend
    and made-up counters -- what the CPU test suite can exercise of the N > 1 path without a GPU."""
    import torch
    import torch.distributed as dist
    from abismal_amd.dist import reduce_stats, ranks_and_rates, shard_bounds
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n = args.reads
    lo, hi = shard_bounds(n * world, rank, world)
    ids = torch.arange(lo, hi, dtype=torch.int64)
    stats = torch.stack([torch.tensor(hi - lo), (ids % 3 == 0).sum(), (ids % 7 == 0).sum(), (ids % 11 == 0).sum(),
                         ids.sum(), (ids * 100).sum()]).to(torch.int64)
    elapsed = 0.5 + 0.25 * rank
    t_el = torch.tensor([elapsed], dtype=torch.float64)
    ranks_seen, rates = ranks_and_rates(n * args.steps, elapsed)
    reduce_stats(stats, t_el)
    if rank == 0:
        print(json.dumps({"metric": "dry run of the multi-GPU plumbing (no mapping)", "value": n * args.steps * world / float(t_el),
                          "unit": "reads/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
                          "warmup": args.warmup, "per_rank_reads_per_s": rates, "dry_run": True, "backend": "gloo",
                          "mapping": {"total": int(stats[0]), "unique": int(stats[1]), "ambiguous": int(stats[2]),
                                      "unseedable": int(stats[3]), "edits": int(stats[4]), "bases": int(stats[5])}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; --pe: one per slot, so that the slots overlap)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 1; --pe: one per slot)")
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("ABM_BENCH_GENOME_MBP", 3100)))
    ap.add_argument("--reads", type=int, default=int(os.environ.get("ABM_BENCH_READS", 10_000_000)),
                    help="reads per step per GPU")
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--mode", choices=["trich", "arich", "random"], default="trich",
                    help="single-end conversion mode: T-rich (default), A-rich (-A) or random PBAT (-R; BASELINE config 5 with --read-len 150)")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("ABM_BENCH_CPU_SAMPLE", 1_000_000)))
    ap.add_argument("--pe", action="store_true",
                    help="paired-end variant (BASELINE config 3): 2 x --read-len pairs from 150-500 bp fragments; "
                         "not the headline metric -- prints its own JSON line")
    ap.add_argument("--streams", type=int, default=24,
                    help="--pe only: consecutive steps alternate over this many (context, stream) slots")
    ap.add_argument("--distinct-batches", type=int, default=4,
                    help="SE: number of different synthetic batches the steps cycle through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phase-stamps", action="store_true",
                    help="after the timed region, run one extra step of the diagnostic kernel and report phase shares "
                         "(single-end: on by default at N=1, see --no-stage-split)")
    ap.add_argument("--no-stage-split", action="store_true", help="single-end: skip the per-stage roofline rows")
    ap.add_argument("--workdir", default=os.environ.get("ABM_BENCH_DIR", "/tmp/abismal_bench"))
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end leg (product sim -> FASTQ -> abismal-amd map -> SAM on tmpfs) at N=1")
    ap.add_argument("--e2e-reads", type=int, default=int(os.environ.get("ABM_BENCH_E2E_READS", 10_000_000)))
    ap.add_argument("--e2e-copies", type=int, default=int(os.environ.get("ABM_BENCH_E2E_COPIES", 4)),
                    help="also time the CLI on the e2e FASTQ concatenated this many times (0/1 = skip)")
    ap.add_argument("--e2e-check", type=int, default=int(os.environ.get("ABM_BENCH_E2E_CHECK", 500_000)),
                    help="reads of the FASTQ prefix mapped by the oracle CLI too (SAM body md5 must agree)")
    ap.add_argument("--e2e-gz-reads", type=int, default=int(os.environ.get("ABM_BENCH_E2E_GZ_READS", 2_000_000)),
                    help="reads of the e2e FASTQ's prefix that also run as gzip-compressed input (0 = skip)")
    ap.add_argument("--seed-ext", default=os.environ.get("ABM_BENCH_SEED_EXT", ""),
                    help="letters of the seed-extension tables as 'a,b' (default: the library's choice from the index's size)")
    ap.add_argument("--window-records", type=int, default=int(os.environ.get("ABM_BENCH_WINDOW_RECORDS", -1)),
                    help="read length the index's window records are built for (abm_index_set_window_records); 0 = none, "
                         "-1 = this run's read length, as `abismal-amd map` takes it from its input")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the embedded runs of BASELINE configs 3 (paired-end 2 x 150) and 5 (150 bp random PBAT) at N=1")
    ap.add_argument("--dist-dry-run", action="store_true",
                    help="exercise launcher + rendezvous + statistics reduce over gloo with made-up counters (no GPU)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = args.streams if args.pe else 3
    if args.warmup is None:
        args.warmup = args.streams if args.pe else 1

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (this process never touches the GPU)
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
                         "pass --gpus equal to --nproc-per-node (or run plain `python bench.py --gpus N`)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dist_dry_run:
        return dist_dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    import abismal_amd as A

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the mapping path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    os.makedirs(args.workdir, exist_ok=True)
    tag = f"g{int(args.genome_mbp)}"
    fasta = os.path.join(args.workdir, tag + ".fa")
    idx = os.path.join(args.workdir, tag + ".idx")
    t_build = 0.0
    # the index file is shared: rank 0 builds it (once per box), the others wait for the FILE -- no rank
    # sits in a collective while another works for minutes
    if rank == 0 and not os.path.exists(idx):
        t0 = time.time()
        synth_genome_fasta(fasta, args.genome_mbp, 1234, dev)
        log(f"synthetic genome written in {time.time() - t0:.1f}s")
        t0 = time.time()
        # (threads: what the container's CPU quota pays for -- 256 runnable threads on a 16-CPU pod are throttled, not faster)
        A.index_build(fasta, idx + ".tmp", int(min(os.cpu_count() or 1, max(1, round(cpu_quota() or 1e9)))))
        os.replace(idx + ".tmp", idx)  # atomic: a waiting rank never sees a partial file
        t_build = time.time() - t0
        log(f"index built in {t_build:.1f}s ({os.path.getsize(idx) / 1e9:.2f} GB)")
        if args.no_e2e and not os.environ.get("ABM_BENCH_KEEP_FASTA"):
            os.remove(fasta)  # (the end-to-end leg simulates its FASTQ from this file)
    t_wait = time.time()
    while not os.path.exists(idx):
        if time.time() - t_wait > 3600:
            raise SystemExit(f"bench.py: rank {rank} waited an hour for {idx}")
        time.sleep(0.5)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    def barrier():
        if world > 1:
            dist.barrier()

    t0 = time.time()
    ext_arg = None
    if args.seed_ext:
        ext_arg = tuple(int(x) for x in args.seed_ext.split(","))
    wrec_for = args.window_records if args.window_records >= 0 else args.read_len
    index = A.Index(idx, seed_extension=ext_arg, window_records=wrec_for)
    if os.environ.get("ABM_BENCH_DIRECT_MIN"):  # (experiments: the smallest range the kernels narrow directly; default 128)
        index.set_direct_narrowing(int(os.environ["ABM_BENCH_DIRECT_MIN"]))
    if args.pe and ext_arg is None:
        index.set_seed_extension_cap(0, 0)  # (as `abismal-amd map` does for pairs: as fast without tables since the pair kernels narrow every big range directly)
    ctx = A.Context(index, local_rank)
    on_planes = ctx.filter_on_planes()
    filter_genome = "bit planes (cooperative window loads)" if on_planes else "nibble array (one lane per window)"
    if on_planes and ctx.window_records() >= args.read_len:
        filter_genome = f"window records for reads up to {ctx.window_records()} bases (one lane per candidate; derived from the bit planes)"
    t_load = time.time() - t0
    index_gb = round(index.device_bytes / 1e9, 2)
    ext = ctx.seed_extension()
    wrec_serves = ctx.window_records()
    seed_tables = {"letters_2": ext[0], "letters_3": ext[1], "gb": round(ext[2] / 1e9, 2),
                   "window_records_serve_reads_up_to": wrec_serves}
    log(f"index loaded + uploaded to HBM in {t_load:.1f}s ({index.device_bytes / 1e9:.2f} GB resident; "
        f"seed-extension tables {ext[0]}+{ext[1]} letters, {ext[2] / 1e9:.1f} GB; window records for reads up to {wrec_serves} bases)")

    names, starts, genome_words = read_index_genome(idx)
    n, L = args.reads, args.read_len
    se_mode = {"trich": A.SE_T_RICH, "arich": A.SE_A_RICH, "random": A.SE_RANDOM}[args.mode]
    mode_name = {"trich": "T-rich mode", "arich": "A-rich mode (-A)", "random": "random-PBAT mode (-R: both conversions, half the reads from the PBAT strand)"}[args.mode]
    if args.pe:
        return run_pe(args, A, ctx, index, genome_words, starts, dev, world, rank, barrier)
    t0 = time.time()
    # every step maps a batch of its own (as consecutive batches of a run would be): a step that
    # re-mapped the previous step's reads would find their index lines still in the last-level cache
    n_batches = max(1, min(args.steps + args.warmup, args.distinct_batches))
    blobs, n_skipped = [], 0
    for k in range(n_batches):
        bk, sk = sample_reads(genome_words, starts, n, L, 1000 + rank + 7919 * k, dev, pbat_frac=0.5 if se_mode == A.SE_RANDOM else 0.0)
        blobs.append(bk)
        n_skipped = sk if k == 0 else n_skipped
    blob = blobs[0]
    off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    log(f"{n_batches} x {n} reads sampled on GPU in {time.time() - t0:.1f}s ({n_skipped} unseedable in the first)")
    del genome_words

    stride = 16
    res = torch.zeros((n, 2), dtype=torch.int32, device=dev)  # abm_hit = 8 bytes
    cig = torch.zeros((n, stride), dtype=torch.int32, device=dev)
    cig_n = torch.zeros((n,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    params = A.Params()
    stream = torch.cuda.current_stream().cuda_stream

    issued = [0]

    def step():
        bk = blobs[issued[0] % n_batches]
        issued[0] += 1
        ctx.map_se_device(se_mode, params, n, bk.data_ptr(), off.data_ptr(), L, res.data_ptr(),
                          cig.data_ptr(), stride, cig_n.data_ptr(), status.data_ptr(), stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.take_work()
    ctx.set_timing(True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    per_launch_ms = ctx.take_kernel_times()
    launches, kernel_ms = len(per_launch_ms), float(sum(per_launch_ms))
    work = ctx.take_work()
    ctx.set_timing(False)
    long_arena = ctx.long_cigars()  # CIGARs of the last step that outgrew their device slot
    blob = blobs[(issued[0] - 1) % n_batches]  # the batch whose results the output buffers hold
    # mapping statistics (six counters, src/abismal.cpp:865-895) reduced over ranks
    pos = res[:, 1]
    flags = (res[:, 0] >> 16) & 0xFFFF
    diffs = res[:, 0] & 0xFFFF
    mapped = pos != 0
    ambig = mapped & ((flags & 0x100) != 0)
    uniq = mapped & ~ambig
    ops = cig.clamp(min=0)
    ref_consuming = ((ops & 15) == 0) | ((ops & 15) == 2)
    valid_op = torch.arange(stride, device=dev)[None, :] < cig_n[:, None]
    bases = ((ops >> 4) * (ref_consuming & valid_op)).sum(1)
    stats = torch.tensor([n, int(uniq.sum()), int(ambig.sum()), n_skipped, int(diffs[uniq].sum()),
                          int(bases[uniq].sum())], dtype=torch.int64, device=dev)
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    from abismal_amd.dist import reduce_stats, ranks_and_rates
    ranks_seen, rank_rates = ranks_and_rates(n * args.steps, elapsed)
    reduce_stats(stats, t_el)  # the path's single collective: RCCL sum of the counters (+ max of the time)
    elapsed = float(t_el.item())
    st_host = int(status.item())

    if world > 1:  # (that was the run's one collective: every rank lets go of its communicator before rank 0 starts the CLI)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    total_reads = n * args.steps * world
    value = total_reads / elapsed
    # ---- roofline of the dominant kernel (map_se_kernel), HBM-bound ------------------------------
    # algorithmic bytes per read:
    #   L + S*16 + P*4.5 + C*4 + (W+C)*8 + A*(L+bw)/2 + 8*(1+ops/2)
    # S seed offsets probed, P narrowing probes, C candidates compared, W read words compared,
    # A alignments, ops CIGAR ops written.  Two sets of counts are reported:
    #  * "kernel tally": what the kernel itself did in the timed launches -- every candidate's full
    #    window (no early exit: W = ceil(L/16) per candidate), minus the candidates served by the
    #    per-call position cache (no window fetched for them);
    #  * "strict": the counts of the reference algorithm -- the oracle's counters on the cpu-baseline
    #    sample of this very batch, full_compare's early exit included (W = words touched before the
    #    running sum passes the cutoff).  `achieved`/`frac` use the strict figure when it is available.
    # The production kernel keeps no work tallies (they cost it registers); counts and phase shares come from ONE more
    # step through the diagnostic build of the kernel (tallies + s_memtime stamps), after the timed region, on a batch
    # the timed steps mapped as well.  Shares are applied to the real kernel's duration.
    phases = None
    ctx.take_work()
    ctx.set_phase_stamps(True)
    step()
    torch.cuda.synchronize()
    work = ctx.take_work()
    ctx.set_phase_stamps(False)
    blob = blobs[(issued[0] - 1) % n_batches]  # (the output buffers now hold this step's results: what the oracle is compared with)
    long_arena = ctx.long_cigars()
    pc = work.get("phase_cycles")
    if pc and not args.no_stage_split:
        phases = {k: round(v / max(1, pc["total"]), 4) for k, v in pc.items() if k != "total"}
        # (filter steps of 128 candidates per read, those holding at most 64, set updates applied as runs of ties)
        phases["filter_steps_per_read"] = round(work.get("filter_steps", 0) / n, 2)
        phases["light_filter_steps_per_read"] = round(work.get("light_filter_steps", 0) / n, 2)
        phases["tie_run_updates_per_read"] = round(work.get("fifo_updates", 0) / n, 2)
    per_launch = {k: float(v) for k, v in work.items() if not isinstance(v, dict)}
    bw_band = 2 * int(0.1 * L) + 1
    nwords = (L + 15) // 16
    ops_per_read = float(cig_n[mapped].sum().item()) / n

    def alg_bytes_per_read(S, P, C, W, Aln, ops, C_windows=None):
        Cw = C if C_windows is None else C_windows  # candidates whose genome window was fetched
        stage = {"probe_narrow": L + S * 16 + P * 4.5,
                 "filter": C * 4 + (W + Cw) * 8,
                 "align": Aln * (L + bw_band) / 2 + 8 * (1 + ops / 2)}
        return sum(stage.values()), stage

    fetched = per_launch["candidates"] - per_launch["window_cache_hits"]
    # 8-byte words a fetched window costs: the read's words on the nibble array; four 16-byte blocks of the bit planes
    # per group of four lanes (reads up to 192 bases), the blocks the window has per group of eight
    words_per_window = nwords if not on_planes else (8 if L <= 192 else 2 * ((L + 63) // 64 + 1))
    k_bytes, k_stage = alg_bytes_per_read(per_launch["seed_offsets"] / n, per_launch["search_probes"] / n,
                                          per_launch["candidates"] / n, fetched * words_per_window / n,
                                          per_launch["alignments"] / n, ops_per_read, C_windows=fetched / n)
    n_long_cigars = int((cig_n > stride).sum().item())
    avg_ms = kernel_ms / max(1, launches)
    # L2->fabric read requests per launch come from a separate rocprofv3 --pmc pass over this same command
    # (profiles/r*_traffic.json, newest round first); reported only when the workload matches
    traffic, traffic_source = lookup_traffic("se_" + args.mode, int(args.genome_mbp), n, L)

    cpu, strict = None, None
    if not args.no_cpu_baseline and world == 1:  # reported at N=1 only
        from tests import oracle_binding as ob  # cpu_baseline leg: the oracle is the thing timed here
        o = ob.load(build=not os.path.exists(ob.LIB))
        ns = min(n, args.cpu_sample)
        host_reads = blob[: ns * L].cpu().numpy().reshape(ns, L)
        seqs = [bytes(r) for r in host_reads]
        seqs = [b"" if s.count(b"N") > L - 44 else s for s in seqs]
        oix = o.index_load(idx)
        # threads: every hardware thread -- unless the container's CPU quota is smaller, in which case four threads per CPU
        # of quota (measured best on the 16-CPU pod: 64 threads 49 k reads/s, 256 threads 38 k: the rest is throttling)
        quota = cpu_quota()
        cores = os.cpu_count() or 1
        if quota:
            cores = int(min(cores, max(1, round(4 * quota))))
        t0 = time.perf_counter()
        o_res, o_cig, o_cn, o_work = o.map_se(oix, seqs, mode=int(se_mode), threads=cores, cig_stride=L + 2)
        t_cpu = time.perf_counter() - t0
        # full comparison on the sample: position, then diffs + flags, then the CIGAR op for op
        g_res = res[:ns].cpu().numpy().view(np.uint32)
        g_cn = cig_n[:ns].cpu().numpy().view(np.uint32)
        g_cig = cig[:ns].cpu().numpy().view(np.uint32)
        g_pos, g_diffs, g_flags = g_res[:, 1], (g_res[:, 0] & 0xFFFF).astype(np.int16), (g_res[:, 0] >> 16).astype(np.uint16)
        same_pos = g_pos == o_res["pos"]
        hit = o_res["pos"] != 0
        same_df = same_pos & (~hit | ((g_diffs == o_res["diffs"]) & (g_flags == o_res["flags"])))
        k = np.arange(stride, dtype=np.uint32)[None, :]
        in_slot = g_cn <= stride
        ops_equal = ((g_cig == o_cig[:, :stride]) | (k >= np.minimum(g_cn, stride)[:, None])).all(1)
        same_cig = same_df & (~hit | ((g_cn == o_cn) & ops_equal))
        # CIGARs longer than the device slot lie in the launch's arena, at the index the slot's first word holds
        long_ids = np.nonzero(hit & ~in_slot)[0]
        long_ok = 0
        for i in long_ids:
            at, k = int(g_cig[i, 0]), int(g_cn[i])
            full = long_arena[at:at + k].tolist() if at + k <= len(long_arena) else None
            good = bool(same_df[i]) and full == o_cig[i, :int(o_cn[i])].tolist()
            same_cig[i] = good
            long_ok += int(good)
        # the same restatement on fewer threads (a fifth of the sample each): whether all hardware threads of a
        # two-socket box are the fairest stand-in for `abismal -t <all cores>`; per-thread and per-core rates beside
        phys = physical_cores()
        sweep = [{"threads": cores, "reads_per_s": round(ns / t_cpu, 1), "per_thread": round(ns / t_cpu / cores, 1), "sample_reads": ns}]
        ns_sw = max(1, ns // 5)
        for th in sorted(({int(max(1, round(quota))), os.cpu_count() or 1} if quota else {max(1, phys // 2), phys}) - {cores}):
            t0 = time.perf_counter()
            o.map_se(oix, seqs[:ns_sw], mode=int(se_mode), threads=th, cig_stride=L + 2)
            dt = time.perf_counter() - t0
            sweep.append({"threads": th, "reads_per_s": round(ns_sw / dt, 1), "per_thread": round(ns_sw / dt / th, 1), "sample_reads": ns_sw})
        best = max(sweep, key=lambda s: s["reads_per_s"])
        o.index_free(oix)
        cpu = {"value": round(ns / t_cpu, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "physical_cores": phys, "cpu_quota_cpus": quota, "per_thread": round(ns / t_cpu / cores, 1),
               # per core: of the cores the process can actually use (its quota, if it has one)
               "per_core": round(ns / t_cpu / (min(phys, quota) if quota else phys), 1),
               "thread_sweep": sorted(sweep, key=lambda s: s["threads"]), "best_of_sweep": best,
               "sample": f"first {ns} reads of rank 0's batch, oracle restatement (-O3 -DNDEBUG), {cores} threads, {t_cpu:.1f}s",
               "positions_identical_to_gpu": f"{int(same_pos.sum())}/{ns}",
               "pos_diffs_flags_identical_to_gpu": f"{int(same_df.sum())}/{ns}",
               "pos_diffs_flags_cigar_identical_to_gpu": f"{int(same_cig.sum())}/{ns}",
               "long_cigars_checked_through_the_arena": f"{long_ok}/{len(long_ids)}"}
        nr = max(1, o_work["reads"])
        o_ops = float(o_cn[hit].sum()) / nr
        s_bytes, s_stage = alg_bytes_per_read(o_work["seed_iters"] / nr, o_work["search_probes"] / nr, o_work["candidates"] / nr,
                                              o_work["words"] / nr, (o_work["aligns"] + o_work["aligns_tb"]) / nr, o_ops)
        strict = {"bytes_per_read": s_bytes, "stage": s_stage,
                  "counts_per_read": {"S": round(o_work["seed_iters"] / nr, 2), "P": round(o_work["search_probes"] / nr, 2),
                                      "C": round(o_work["candidates"] / nr, 2), "W": round(o_work["words"] / nr, 2),
                                      "A": round((o_work["aligns"] + o_work["aligns_tb"]) / nr, 2), "ops": round(o_ops, 3)},
                  "words_per_candidate": round(o_work["words"] / max(1, o_work["candidates"]), 3)}

    basis = "strict" if strict else "kernel_tally"
    use_bytes, use_stage = (strict["bytes_per_read"], strict["stage"]) if strict else (k_bytes, k_stage)
    achieved = use_bytes * n / (avg_ms * 1e-3) / 1e9
    k_achieved = k_bytes * n / (avg_ms * 1e-3) / 1e9
    stages = None
    if phases:
        share = {"probe_narrow": phases.get("probe_narrow", 0.0), "filter": phases.get("gather_hamming", 0.0) + phases.get("replay", 0.0),
                 "align": phases.get("align", 0.0)}
        stages = []
        for name in ("probe_narrow", "filter", "align"):
            ms = share[name] * avg_ms
            gbps = use_stage[name] * n / (ms * 1e-3) / 1e9 if ms > 0 else None
            stages.append({"stage": name, "time_share": round(share[name], 4), "ms": round(ms, 2),
                           "alg_bytes_per_read": round(use_stage[name], 1), "kernel_tally_bytes_per_read": round(k_stage[name], 1),
                           "achieved_GBps": round(gbps, 1) if gbps else None, "frac": round(gbps / 8000.0, 5) if gbps else None})
    roofline = {"bound": "hbm", "kernel": "map_se_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "basis": basis,
                "traffic": traffic, "traffic_source": traffic_source,
                "traffic_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                "avg_kernel_ms": round(avg_ms, 3),
                "kernel_ms_per_launch": [round(x, 1) for x in per_launch_ms],
                "alg_bytes_per_read": round(use_bytes, 1),
                "alg_bytes_per_read_strict": round(strict["bytes_per_read"], 1) if strict else None,
                "alg_bytes_per_read_kernel_tally": round(k_bytes, 1),
                "frac_strict": round(strict["bytes_per_read"] * n / (avg_ms * 1e-3) / 1e9 / 8000.0, 5) if strict else None,
                "frac_kernel_tally": round(k_achieved / 8000.0, 5),
                "strict_counts_per_read": strict["counts_per_read"] if strict else None,
                "strict_words_per_candidate": strict["words_per_candidate"] if strict else None,
                "strict_note": ("S,P,C,W,A,ops per read = the oracle's counters (reference algorithm, early-exit full_compare) on the "
                                "cpu_baseline sample of this batch; bytes/read x reads per launch / avg kernel time") if strict else None,
                "stages": stages,
                "stage_shares_source": "one extra step of the s_memtime-stamped diagnostic kernel after the timed region; "
                                       "shares applied to avg_kernel_ms (filter = window gather + Hamming + ordered replay)" if stages else None,
                "gathers_per_s": round((per_launch["seed_offsets"] * 2 + per_launch["candidates"] * 2 +
                                        per_launch["search_probes"] * 2) / (avg_ms * 1e-3), 0)}

    e2e = None
    if not args.no_e2e and args.mode in ("trich", "random"):  # (the other ranks have left by now: the CLI child drives all `world` GPUs itself)
        # free this process's HBM first: the CLI is a process of its own on the same GPU
        del blobs, blob, res, cig, cig_n
        ctx.close()
        index.close()
        torch.cuda.empty_cache()
        try:
            e2e = run_e2e(args, idx, fasta, L, gpus=world, kind="random" if args.mode == "random" else "se")
        except Exception as exc:  # the line's `value` has been measured: a failing end-to-end leg must not lose it
            e2e = {"error": f"{type(exc).__name__}: {exc}"[:1500]}
        log(f"e2e: {e2e}")

    # the other single-GPU configurations of BASELINE.json, each by a fresh child process of this script once this
    # process has let go of the GPU (their own JSON lines, embedded): config 3 (paired-end 2 x 150) and config 5
    # (150 bp single-end, random PBAT) -- shortened so that the default run stays within minutes
    other = None
    if world == 1 and args.mode == "trich" and not args.no_other_configs:
        import subprocess
        if e2e is None:
            del blobs, blob, res, cig, cig_n
            ctx.close()
            index.close()
            torch.cuda.empty_cache()
        other = {}
        common = [sys.executable, os.path.abspath(__file__), "--no-other-configs", "--genome-mbp", str(args.genome_mbp),
                  "--workdir", args.workdir] + (["--no-e2e"] if args.no_e2e else [])
        for key, extra in (("config3_paired_end_2x150", ["--pe", "--reads", "1000000", "--read-len", "150", "--steps", "24", "--warmup", "24", "--cpu-sample", "200000",
                                                         "--e2e-reads", "2000000", "--e2e-check", "20000", "--e2e-copies", "4"]),
                           ("config5_random_pbat_150", ["--mode", "random", "--read-len", "150", "--reads", "4000000", "--steps", "3", "--warmup", "1",
                                                        "--cpu-sample", "200000", "--e2e-reads", "4000000", "--e2e-check", "200000", "--e2e-copies", "1",
                                                        "--e2e-gz-reads", "0"])):
            t0 = time.time()
            r = subprocess.run(common + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and rows:
                other[key] = json.loads(rows[-1])
                other[key]["wall_s"] = round(time.time() - t0, 1)
            else:
                other[key] = {"error": r.stderr[-1500:]}
            log(f"{key}: {other[key].get('value')} {other[key].get('unit', '')} in {time.time() - t0:.0f}s")

    # what BASELINE's metric defines -- FASTQ on disk -> SAM on disk -- for this configuration and the other two, inside
    # `config` (the driver's record keeps `config`, `roofline` and `cpu_baseline` whole and drops other top-level keys)
    def e2e_summary(row, kernel_value):
        if not isinstance(row, dict) or not row.get("value"):
            return {"error": (row or {}).get("error", "not run")[:200]} if isinstance(row, dict) else None
        out = {"reads_per_s": row["value"], "over_kernel": round(row["value"] / kernel_value, 4) if kernel_value else None,
               "statistic": row.get("statistic"), "seconds_of_each_run": row.get("seconds_of_each_run"), "reads": row.get("reads"),
               "sam_body_md5_equals_oracle_cli": (row.get("parity") or {}).get("identical")}
        for k in ("host_ceiling_reads_per_s", "host_ceiling_reads_per_s_dev_null", "host_ceiling_reads_per_s_8_virtual_gpus_8_parts"):
            if row.get(k) is not None:
                out[k] = row[k]
        gz = row.get("gzip_input")
        if isinstance(gz, dict):
            out["gzip_input_reads_per_s"] = gz.get("value")
            out["bgzf_input_reads_per_s"] = (gz.get("bgzf") or {}).get("value")
        if isinstance(row.get("sustained"), dict):
            out["sustained_reads_per_s"] = row["sustained"].get("value")
        return out

    other_summary = None
    if other:
        other_summary = {}
        for key, row in other.items():
            if "error" in row and "value" not in row:
                other_summary[key] = {"error": row["error"][-300:]}
                continue
            other_summary[key] = {"value": row.get("value"), "unit": row.get("unit"), "ms_per_step": row.get("ms_per_step"),
                                  "workload": (row.get("config") or {}).get("workload"),
                                  "roofline_frac": (row.get("roofline") or {}).get("frac"),
                                  "roofline_basis": (row.get("roofline") or {}).get("basis"),
                                  "traffic_over_algorithmic": (row.get("roofline") or {}).get("traffic_over_algorithmic"),
                                  "ms_per_launch": (row.get("roofline") or {}).get("ms_per_launch"),
                                  "end_to_end": e2e_summary(row.get("e2e"), row.get("value")),
                                  "parity_sample": {k: v for k, v in (row.get("cpu_baseline") or {}).items() if "identical" in k or "vs_oracle" in k} or None}
    line = {
        "metric": f"mapped reads/sec (whole node), {L} bp SE on hg38-scale index" + ("" if args.mode == "trich" else f", {mode_name}"),
        "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "ranks_seen": ranks_seen,
        "per_rank_reads_per_s": rank_rates, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"synthetic hg38-shaped genome {args.genome_mbp:g} Mbp (hg38 unavailable offline), "
                               f"{n} sim-like reads x {L} bp SE per GPU per step, {mode_name}",
                   "reads_per_step_per_gpu": n, "read_len": L, "index_gb": index_gb,
                   "parallelism": f"reads sharded over {world} GPU(s), index replicated",
                   "end_to_end": e2e_summary(e2e, value) if e2e is not None else None,
                   "other_configs": other_summary},
        "roofline": roofline, "cpu_baseline": cpu,
        # SURVEY 8(d)'s metric proper -- FASTQ on disk -> SAM on disk through `abismal-amd map -gpus N` -- beside `value`
        # (the HBM-resident kernel-loop rate the measurement contract defines; at N > 1 `value` is N independent loops
        # added up, e2e is ONE process driving N GPUs through the host pipeline)
        "e2e_reads_per_s": e2e.get("value") if isinstance(e2e, dict) else None,
        "e2e_over_kernel": round(e2e["value"] / value, 4) if isinstance(e2e, dict) and e2e.get("value") else None,
        "e2e": e2e, "other_configs": other,
        "mapping": {"total": int(stats[0]), "unique": int(stats[1]), "ambiguous": int(stats[2]),
                    "unseedable": int(stats[3]), "edits": int(stats[4]), "bases": int(stats[5])},
        "work_per_read": {k: round(v / n, 2) for k, v in per_launch.items()},
        "phase_shares_diagnostic": phases,
        "kernel_status": st_host,
        "filter_genome": filter_genome,
        "seed_extension_tables": seed_tables,
        "long_cigars": {"slot_ops": stride, "reads_beyond_slot": n_long_cigars, "returned_through": "per-launch arena (abm_ctx_long_cigars)"},
        "index_build_s": round(t_build, 1), "index_upload_s": round(t_load, 1),
    }
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
