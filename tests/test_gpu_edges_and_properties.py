"""GPU: edge cases of the C ABI and size-independent properties of the mapping path."""
import os

import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(trex_index):
    import abismal_amd as A
    ix = A.Index(trex_index)
    c = A.Context(ix, 0)
    yield c
    c.close()
    ix.close()


@pytest.fixture(scope="module")
def reads(oracle, workdir):
    prefix = os.path.join(workdir, "edge_reads")
    oracle.simulate(os.path.join(GOLD, "tRex1.fa"), prefix, 20000, single_end=True, seed=3)
    return ob.read_fastq_like_readloader(prefix + "_1.fq")[1]


def cigars(cig, off):
    return [tuple(cig[int(off[i]):int(off[i + 1])].tolist()) for i in range(len(off) - 1)]


def test_empty_batch_and_all_skipped(ctx):
    res, cig, off = ctx.map_se([])
    assert len(res) == 0 and off.tolist() == [0]
    res, cig, off = ctx.map_se(["", "", ""])
    assert (res["pos"] == 0).all() and off.tolist() == [0, 0, 0, 0]
    pairs, se1, se2, (c1, o1), (c2, o2) = ctx.map_pe(["", ""], ["", ""])
    assert (pairs["r1"]["pos"] == 0).all() and (se1["pos"] == 0).all() and (se2["pos"] == 0).all()


def _reads_from_genome(trex_index, lengths, seed, mut=0.02, indel_every=0):
    import bench
    names, starts, gw = bench.read_index_genome(trex_index)
    dec = np.frombuffer(b"NACNGNNNTNNNNNNN", dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(seed)
    reads = []
    for k, L in enumerate(lengths):
        ch = 1 + k % 2
        p = int(rng.integers(int(starts[ch]) + 1000, int(starts[ch + 1]) - L - 1000))
        idx = np.arange(p, p + L + 64)
        s = dec[(gw[idx >> 4] >> ((idx & 15).astype(np.uint64) << np.uint64(2))) & np.uint64(15)].copy()
        if indel_every:  # a few small deletions and insertions along the read
            pieces, at = [], 0
            while at < len(s):
                step = int(rng.integers(indel_every // 2, indel_every * 2))
                pieces.append(s[at:at + step])
                at += step
                if rng.random() < 0.5:
                    at += int(rng.integers(1, 3))
                else:
                    pieces.append(acgt[rng.integers(0, 4, int(rng.integers(1, 3)))])
            s = np.concatenate(pieces)
        s = s[:L].copy()
        if k % 3 == 2:  # every third read from the other strand
            s = np.frombuffer(bytes(s).translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1], dtype=np.uint8).copy()
        s[s == ord("C")] = ord("T")
        m = rng.random(len(s)) < mut
        s[m] = acgt[rng.integers(0, 4, int(m.sum()))]
        reads.append(bytes(s).decode().replace("N", "A"))
    return reads


def test_long_reads(ctx, oracle, trex_index):
    """Reads of any length the reference takes (below 32767 bases, src/abismal.cpp:179-185) map like any other: up to
    1024 bases in the batch's ordinary launch, longer ones in the long-read launch (traceback table in global memory),
    mixed freely in one batch -- positions, edit distances, flags and CIGARs equal the oracle's."""
    import abismal_amd as A
    from tests.test_gpu_se_parity import compare_se
    assert A.load_library().abm_max_read_length() == 32766
    lengths = [300, 513, 800, 1024, 1024, 700, 100, 1500, 1025, 64, 2000, 2048, 3000, 5000, 10000, 10000, 1100, 99, 16383, 2500]
    # (a small indel every few thousand bases: the reference's band is 61 wide, so a long read may drift by 30 at most)
    reads = _reads_from_genome(trex_index, lengths, seed=9, indel_every=3000)
    before = ctx.reads_too_long()
    res, cig, off = ctx.map_se(reads)
    assert ctx.reads_too_long() == before
    oix = oracle.index_load(trex_index)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, threads=8)
        compare_se(res, cig, off, o_res, o_cig, o_n, reads, "long reads, T-rich")
        assert (res["pos"] != 0).sum() >= len(reads) - 4
        mapped_long = [i for i in range(len(reads)) if len(reads[i]) >= 5000 and res["pos"][i] != 0]
        assert len(mapped_long) >= 2, "the fixture's long reads must map"
        assert max(int(off[i + 1] - off[i]) for i in mapped_long) > 4, "long reads must carry indels (CIGARs through the arena)"
        # random-PBAT mode: both conversions tried on every read
        res2, cig2, off2 = ctx.map_se(reads, mode=2)
        o2 = oracle.map_se(oix, reads, mode=2, threads=8)
        compare_se(res2, cig2, off2, o2[0], o2[1], o2[2], reads, "long reads, random PBAT")
    finally:
        oracle.index_free(oix)
    with pytest.raises(A.AbismalAmdError):
        ctx.map_se(["ACGT" * 30], mode=7)


def test_reads_beyond_16383_bases(ctx, oracle, trex_index):
    """Beyond 16383 bases a perfect alignment's score (2 L) no longer fits the reference's 16-bit score_t
    (src/AbismalAlign.hpp:35): the long-read launch narrows every score to 16 bits exactly as that arithmetic does, so
    results still equal the restatement's, whatever they are worth.  32766 bases is the longest read the reference
    takes; one base more is an error there and comes back unmapped and counted here."""
    from tests.test_gpu_se_parity import compare_se
    reads = _reads_from_genome(trex_index, [20000, 32766, 17000, 150], seed=10, mut=0.01)
    res, cig, off = ctx.map_se(reads)
    oix = oracle.index_load(trex_index)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, threads=4)
    finally:
        oracle.index_free(oix)
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, "reads beyond 16383 bases")
    before = ctx.reads_too_long()
    res, cig, off = ctx.map_se([reads[3], reads[1] + "A", reads[3]])
    assert ctx.reads_too_long() - before == 1 and res["pos"][1] == 0 and res["pos"][0] == res["pos"][2] != 0


def _pairs_from_genome(trex_index, shapes, seed, indel_every=2500):
    """(len1, len2, fragment) triples -> (reads1, reads2): read 1 = the first len1 bases of a converted, mutated fragment,
    read 2 = the reverse complement of its last len2 bases (what `sim` makes, src/simreads.cpp:113-133)."""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    frags = _reads_from_genome(trex_index, [f for _, _, f in shapes], seed=seed, indel_every=indel_every)
    r1 = [fr[:a] for fr, (a, _, _) in zip(frags, shapes)]
    r2 = [fr[len(fr) - b:].encode().translate(comp)[::-1].decode() for fr, (_, b, _) in zip(frags, shapes)]
    return r1, r2


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pairs_with_a_long_end(ctx, oracle, trex_index, mode):
    """Pairs whose ends are longer than the pair kernels' 1024 bases (up to the reference's own limit, 32766,
    src/abismal.cpp:179-185) map like any other -- in a launch of their own after the batch's two tiers, with the read
    data and the traceback tables in global memory and 16-bit scores like the reference's (src/AbismalAlign.hpp:35) --
    mixed freely with ordinary pairs: pair, both fallback hits and both CIGARs equal the oracle's in the three
    paired-end modes (normal, PBAT, random PBAT)."""
    import abismal_amd as A
    from tests.test_gpu_pe_parity import compare_pe
    shapes = [(150, 150, 400), (1500, 150, 1700), (150, 1500, 1650), (100, 100, 300), (1025, 1025, 1300), (5000, 5000, 6000),
              (2000, 3000, 4000), (150, 150, 500), (20000, 300, 20400), (17000, 17000, 18000), (1024, 1024, 1500), (10000, 64, 10100)]
    r1, r2 = _pairs_from_genome(trex_index, shapes, seed=31)
    if mode == 1:    # PBAT: read 1 is the A-rich one
        r1, r2 = r2, r1
    elif mode == 2:  # random PBAT: either
        r1, r2 = [a if k % 2 else b for k, (a, b) in enumerate(zip(r1, r2))], [b if k % 2 else a for k, (a, b) in enumerate(zip(r1, r2))]
    p = A.Params(max_frag=40000)
    before = ctx.reads_too_long()
    gpu = ctx.map_pe(r1, r2, mode=mode, params=p)
    assert ctx.reads_too_long() == before
    oix = oracle.index_load(trex_index)
    try:
        orc = oracle.map_pe(oix, r1, r2, mode=mode, max_frag=40000, threads=8)
    finally:
        oracle.index_free(oix)
    compare_pe(gpu, orc, f"pairs with long ends, mode {mode}")
    mapped = gpu[0]["r1"]["pos"] != 0
    long_ones = [k for k, (a, b, _) in enumerate(shapes) if max(a, b) > 1024]
    assert sum(bool(mapped[k]) for k in long_ones) >= len(long_ones) - 2, "the fixture's long pairs must map concordantly"
    # an end beyond the reference's limit: unmapped, counted, the rest of the batch unaffected
    too = ["ACGT" * 8192 + "AC"]  # 32770 bases
    gpu2 = ctx.map_pe(r1[:2] + too, r2[:2] + [r2[0]], mode=mode, params=p)
    assert ctx.reads_too_long() == before + 1 and gpu2[0]["r1"]["pos"][2] == 0
    assert gpu2[0]["r1"]["pos"][0] == gpu[0]["r1"]["pos"][0] and gpu2[0]["r1"]["pos"][1] == gpu[0]["r1"]["pos"][1]


def test_unseedable_and_ragged_inputs(ctx, oracle, trex_index, reads):
    rng = np.random.default_rng(0)
    weird = ["N" * 60, "A" * 44, "ACGT" * 11, "T" * 300, "", "NNNNACGTNNNN" * 8, "GATTACA" * 40]
    ragged = [r[: int(rng.integers(44, len(r) + 1))] if r and len(r) >= 47 else r for r in reads[:3000]]
    # (reads of 44-46 bases included: what they see past their end comes from the reads before them, in order)
    batch = weird + ragged
    res, cig, off = ctx.map_se(batch)
    oix = oracle.index_load(trex_index)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, [b if len(b) >= 44 else "" for b in batch], threads=8)
    finally:
        oracle.index_free(oix)
    from tests.test_gpu_se_parity import compare_se
    compare_se(res, cig, off, o_res, o_cig, o_n, batch, "ragged batch")


def test_batch_order_and_size_do_not_matter(ctx, reads):
    """Results are per read: permuting the batch, splitting it, or running it twice changes nothing
    (the only cross-read state in the reference is buffer sizing, src/abismal.cpp:1549)."""
    # reads of 44-46 bases are the exception: they see what EARLIER reads left in the reference's reused
    # buffers, so their results depend on the order by design (test_ghost_reads)
    base = [r for r in reads if not (44 <= len(r) <= 46)]
    res, cig, off = ctx.map_se(base)
    res2, cig2, off2 = ctx.map_se(base)
    assert res.tobytes() == res2.tobytes() and cig.tobytes() == cig2.tobytes()
    perm = np.random.default_rng(1).permutation(len(base))
    resp, cigp, offp = ctx.map_se([base[i] for i in perm])
    c0, cp = cigars(cig, off), cigars(cigp, offp)
    for k, i in enumerate(perm):
        assert res[i]["pos"] == resp[k]["pos"]
        if res[i]["pos"]:
            assert res[i].tobytes() == resp[k].tobytes() and c0[i] == cp[k]
    half = len(base) // 2
    ra, ca, oa = ctx.map_se(base[:half])
    rb, cb, ob_ = ctx.map_se(base[half:])
    both = np.concatenate([ra, rb])
    m = res["pos"] != 0
    assert (both["pos"] == res["pos"]).all() and (both[m] == res[m]).all()
    assert cigars(ca, oa) + cigars(cb, ob_) == c0


def test_exact_reads_map_back_to_their_origin(ctx, trex_index):
    """Reads cut from the genome without errors (fully converted) must come back unique at their
    origin with NM 0 and an all-match CIGAR -- or be flagged ambiguous, never somewhere else."""
    import bench
    names, starts, gw = bench.read_index_genome(trex_index)
    dec = np.frombuffer(b"NACNGNNNTNNNNNNN", dtype=np.uint8)
    rng = np.random.default_rng(5)
    L = 100
    pos = rng.integers(int(starts[1]) + 10100, int(starts[2]) - 200, 4000)
    out, keep = [], []
    for p in pos:
        idx = np.arange(p, p + L)
        nib = ((gw[idx >> 4] >> ((idx & 15).astype(np.uint64) * np.uint64(4))) & np.uint64(15)).astype(np.int64)
        s = dec[nib]
        if (s == ord("N")).any():
            continue
        out.append(bytes(s).replace(b"C", b"T"))
        keep.append(int(p))
    res, cig, off = ctx.map_se(out)
    cg = cigars(cig, off)
    n_unique = 0
    for i, p in enumerate(keep):
        assert res[i]["pos"] != 0, "an error-free read must seed (>= 44 bp guarantees a hit)"
        if not (int(res[i]["flags"]) & 0x100):
            assert int(res[i]["pos"]) == p and int(res[i]["diffs"]) == 0 and cg[i] == ((L << 4),)
            n_unique += 1
    assert n_unique > 0.9 * len(keep)


def test_device_entry_point_matches_host_entry_point(ctx, reads):
    import torch
    import abismal_amd as A
    base = [r for r in reads[:5000] if len(r) == 100]
    res, cig, off = ctx.map_se(base)
    dev = torch.device("cuda", 0)
    n, L = len(base), 100
    blob = torch.tensor(np.frombuffer("".join(base).encode(), dtype=np.uint8), device=dev)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    stride = 102
    d_res = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    d_cig = torch.zeros((n, stride), dtype=torch.int32, device=dev)
    d_n = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.map_se_device(A.SE_T_RICH, A.Params(), n, blob.data_ptr(), offs.data_ptr(), L, d_res.data_ptr(),
                      d_cig.data_ptr(), stride, d_n.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(d_st.item()) == 0
    got = d_res.cpu().numpy().view(A.HIT_DTYPE).reshape(-1)
    m = res["pos"] != 0
    assert (got["pos"] == res["pos"]).all() and (got[m] == res[m]).all()
    cn = d_n.cpu().numpy()
    dc = d_cig.cpu().numpy().view(np.uint32)
    assert [tuple(dc[i, :cn[i]].tolist()) for i in range(n)] == cigars(cig, off)


def test_long_cigars_come_back_through_the_arena(ctx, reads):
    """Device entry point with 2-op slots: every CIGAR with more ops lies whole in the context's arena, at the
    index its slot's first word holds (abm_ctx_long_cigars) -- nothing is truncated, nothing is mapped twice."""
    import torch
    import abismal_amd as A
    base = [r for r in reads[:6000] if len(r) == 100]
    res, cig, off = ctx.map_se(base)
    want = cigars(cig, off)
    dev = torch.device("cuda", 0)
    n, L, stride = len(base), 100, 2
    blob = torch.tensor(np.frombuffer("".join(base).encode(), dtype=np.uint8), device=dev)
    offs = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
    d_res = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    d_cig = torch.zeros((n, stride), dtype=torch.int32, device=dev)
    d_n = torch.zeros(n, dtype=torch.int32, device=dev)
    d_st = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.map_se_device(A.SE_T_RICH, A.Params(), n, blob.data_ptr(), offs.data_ptr(), L, d_res.data_ptr(),
                      d_cig.data_ptr(), stride, d_n.data_ptr(), d_st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(d_st.item()) == 0
    arena = ctx.long_cigars()
    cn = d_n.cpu().numpy()
    dc = d_cig.cpu().numpy().view(np.uint32)
    got = []
    for i in range(n):
        k = int(cn[i])
        got.append(tuple(dc[i, :k].tolist()) if k <= stride else tuple(arena[int(dc[i, 0]):int(dc[i, 0]) + k].tolist()))
    assert got == want
    # (the arena may also hold CIGARs of reads whose alignment was discarded afterwards)
    assert sum(1 for c in want if len(c) > stride) > 100 and len(arena) >= sum(len(c) for c in want if len(c) > stride)


def test_small_cigar_capacity_is_reported_not_overrun(ctx, reads):
    """abm_map_se_batch with too little room for the CIGARs returns ABM_ERR_CAPACITY (-2) and the
    same call with enough room succeeds (the CLI relies on this to start with a small buffer)."""
    import ctypes as C
    import abismal_amd.api as A
    sub = reads[:2000]
    sub = [r if isinstance(r, bytes) else r.encode() for r in sub]
    blob = np.frombuffer(b"".join(sub), dtype=np.uint8).copy()
    off = np.zeros(len(sub) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in sub])
    n = len(sub)
    res = np.zeros(n, dtype=A.HIT_DTYPE)
    co = np.zeros(n + 1, dtype=np.uint64)
    lib, p = A.load_library(), A.Params()
    small = np.zeros(16, dtype=np.uint32)
    rc = lib.abm_map_se_batch(ctx.handle, 0, C.byref(p), n, blob.ctypes.data, off.ctypes.data, res.ctypes.data,
                              small.ctypes.data, len(small), co.ctypes.data)
    assert rc == -2, (rc, lib.abm_last_error())
    big = np.zeros(n * 102, dtype=np.uint32)
    rc = lib.abm_map_se_batch(ctx.handle, 0, C.byref(p), n, blob.ctypes.data, off.ctypes.data, res.ctypes.data,
                              big.ctypes.data, len(big), co.ctypes.data)
    assert rc == 0, lib.abm_last_error()
    mapped = int((res["pos"] != 0).sum())
    assert mapped > 0.8 * n and int(co[-1]) >= mapped


@pytest.mark.parametrize("mode,maxc", [(0, 0), (0, 5), (1, 5), (2, 5), (2, 100)])
def test_ghost_reads(oracle, tmp_path_factory, mode, maxc):
    """Reads of 44-46 bases hash and extend seeds PAST their end, into what the reads before them left
    in the reference's reused buffers (src/abismal.cpp:1163-1194, :1302-1308, :1377-1386): the result
    depends on the preceding reads, in input order, exactly as at -t 1.  Half the batch is such reads,
    between reads of many other lengths; a small -c forces the over-long extension to happen often."""
    import abismal_amd as A
    from tests import synth
    from tests.test_gpu_se_parity import compare_se
    wd = tmp_path_factory.mktemp("ghost")
    fa, idx = str(wd / "rep.fa"), str(wd / "rep.idx")
    synth.repeat_rich_genome(fa, seed=12, n_chroms=2, chrom_len=800_000)
    A.index_build(fa, idx, 8)
    rng = np.random.default_rng(40 + mode + maxc)
    long_reads = synth.trim_like_readloader(synth.mutated_reads(fa, 4000, 120, seed=3, mut=0.02, pbat_frac=0.5 if mode else 0.0))
    reads = []
    for r in long_reads:
        u = rng.random()
        if u < 0.5 and len(r) >= 47:
            r = r[: int(rng.integers(44, 47))]
        elif u < 0.7 and len(r) >= 60:
            r = r[: int(rng.integers(47, len(r) + 1))]
        elif u < 0.75:
            r = ""
        reads.append(r)
    assert sum(1 for r in reads if 44 <= len(r) <= 46) > 1000
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    oix = oracle.index_load(idx)
    try:
        o_res, o_cig, o_n, work = oracle.map_se(oix, reads, mode=mode, max_candidates=maxc, threads=1)
        res, cig, off = ctx.map_se(reads, mode=mode, params=A.Params(max_candidates=maxc))
        compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"ghost reads mode {mode} -c {maxc}")
        # paired-end: the same for both ends' buffers
        from tests.test_gpu_pe_parity import compare_pe
        r1, r2 = reads[:1500], reads[1500:3000]
        orc = oracle.map_pe(oix, r1, r2, mode=mode, max_candidates=maxc, threads=1)
        gpu = ctx.map_pe(r1, r2, mode=mode, params=A.Params(max_candidates=maxc))
        compare_pe(gpu, orc, f"ghost pairs mode {mode} -c {maxc}")
        # ... and with the pair kernels narrowing every range above -c directly: the offsets that start beyond a
        # short read's end must still go through the letter loop and its ghost bits
        ix.set_direct_narrowing(1)
        gpu = ctx.map_pe(r1, r2, mode=mode, params=A.Params(max_candidates=maxc))
        compare_pe(gpu, orc, f"ghost pairs mode {mode} -c {maxc}, direct narrowing from 1 entry")
    finally:
        oracle.index_free(oix)
        ctx.close()
        ix.close()


@pytest.mark.parametrize("max_len", [150, 192, 193, 240, 300, 448, 449])
def test_read_lengths_around_the_filter_lane_groups(ctx, oracle, trex_index, max_len):
    """The cooperative filter shares a candidate's window among 4 lanes up to 192 bases and among 8 up to 448
    (64 bases of bit planes per lane); longer batches take the one-lane-per-window path.  Same results either side."""
    import bench
    from tests.test_gpu_se_parity import compare_se
    names, starts, gw = bench.read_index_genome(trex_index)
    dec = np.frombuffer(b"NACNGNNNTNNNNNNN", dtype=np.uint8)
    rng = np.random.default_rng(max_len)
    reads = []
    for k in range(400):
        L = max_len if k == 0 else int(rng.integers(60, max_len + 1))
        p = int(rng.integers(int(starts[1]) + 20000, int(starts[2]) - 3000))
        idx = np.arange(p, p + L)
        s = dec[(gw[idx >> 4] >> ((idx & 15).astype(np.uint64) << np.uint64(2))) & np.uint64(15)].copy()
        s[s == ord("C")] = ord("T")
        mut = rng.random(L) < 0.03
        s[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
        reads.append(bytes(s).decode().replace("N", "A"))
    res, cig, off = ctx.map_se(reads)
    oix = oracle.index_load(trex_index)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, threads=4)
    finally:
        oracle.index_free(oix)
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"reads up to {max_len} bases")
    assert (res["pos"] != 0).mean() > 0.8


def test_windows_that_reach_into_an_n_run(oracle, workdir):
    """Candidates whose genome window overlaps a long N run (blank nibbles: they match nothing, not even the padding
    of a read's last word) are redone on the nibble array by the bit-plane filter.  Reads cut right at the edges of
    an N run, at every distance 0..24 from it, and on both strands."""
    import abismal_amd as A
    from tests import synth
    from tests.test_gpu_se_parity import compare_se
    fa = os.path.join(workdir, "nrun.fa")
    synth.repeat_rich_genome(fa, seed=11, n_chroms=2, chrom_len=400_000)
    idx = os.path.join(workdir, "nrun.idx")
    oracle.index_build(fa, idx, threads=4)
    chroms = [np.frombuffer(rec.split(b"\n", 1)[1].replace(b"\n", b"").upper(), dtype=np.uint8)
              for rec in open(fa, "rb").read().split(b">")[1:]]
    reads = []
    for ch in chroms:
        mid = len(ch) // 2
        for L in (100, 97, 150, 300):  # (300: a window shared by eight lanes)
            for k in range(25):
                for seg in (ch[mid - L - k: mid - k], ch[mid + 3000 + k: mid + 3000 + k + L], ch[50 + k: 50 + k + L] if ch[0] == ord("N") else ch[k: k + L]):
                    s = seg.copy()
                    s[s == ord("C")] = ord("T")
                    reads.append(bytes(s).decode())
                    reads.append(bytes(synth.COMP[seg[::-1]]).decode().replace("C", "T"))
    index = A.Index(idx)
    ctx = A.Context(index, 0)
    assert ctx.filter_on_planes()  # N runs (long ones stay blank nibbles) do not take the genome off the bit planes
    res, cig, off = ctx.map_se(reads)
    oix = oracle.index_load(idx)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, threads=4)
    finally:
        oracle.index_free(oix)
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, "reads at the edges of N runs")
    assert (res["pos"] != 0).mean() > 0.5
    # the same for pairs: end 1 cut at the edge of the N run, end 2 from 150-400 bases further out, opposite strand
    from tests.test_gpu_pe_parity import compare_pe
    r1, r2 = [], []
    rng = np.random.default_rng(4)
    for ch in chroms:
        mid = len(ch) // 2
        for L in (100, 150):
            for k in range(25):
                gap = int(rng.integers(150, 400))
                a, b = ch[mid - L - k: mid - k], ch[mid - k - gap - L: mid - k - gap]         # left of the run
                c, d = ch[mid + 3000 + k: mid + 3000 + k + L], ch[mid + 3000 + k + gap: mid + 3000 + k + gap + L]  # right of it
                for fwd, rev in ((b, a), (c, d)):
                    r1.append(bytes(fwd).decode().replace("C", "T"))
                    r2.append(bytes(synth.COMP[rev[::-1]]).decode().replace("G", "A"))
    gpu = ctx.map_pe(r1, r2)
    oix = oracle.index_load(idx)
    try:
        orc = oracle.map_pe(oix, r1, r2, mode=0)
    finally:
        oracle.index_free(oix)
    compare_pe(gpu, orc, "pairs at the edges of N runs")
