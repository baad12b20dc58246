"""GPU parity of the window records (abm_index_set_window_records; abm_ext.hip build_window_records): whatever read
length the records were built for -- none (windows gathered from the bit planes, one random line per candidate as
check_hits' gathers are, src/abismal.cpp:1124-1150), 108 / 140 / 172 bases (records of 3 / 4 / 5 blocks) -- results
equal the oracle's read for read: batches the records serve (the record-fed kernel: windows addressed by entry
number, index entries read only for candidates within the cutoff), batches with a read too long for them (the
bit-plane kernel), ragged batches, reads at the edges of N runs (redone on the nibble array), the short-read index."""
import os

import numpy as np
import pytest

from tests.test_gpu_se_parity import compare_se

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rep(oracle, workdir):
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "rep_wrec.fa")
    synth.repeat_rich_genome(fa, seed=5)
    idx = os.path.join(workdir, "rep_wrec.idx")
    A.index_build(fa, idx, 8)
    oix = oracle.index_load(idx)
    yield fa, idx, oix
    oracle.index_free(oix)


@pytest.mark.parametrize("built_for,serves", [(0, 0), (100, 108), (108, 108), (109, 140), (150, 172), (400, 172)])
def test_se_any_record_size_equals_the_oracle(oracle, rep, built_for, serves):
    import abismal_amd as A
    from tests import synth
    fa, idx, oix = rep
    ix = A.Index(idx, window_records=built_for)
    ctx = A.Context(ix, 0)
    try:
        assert ctx.window_records() == serves
        for mode, L in ((0, 100), (1, 108), (2, 150), (0, 66), (0, 46), (0, 128), (0, 172), (0, 180)):
            reads = synth.trim_like_readloader(synth.mutated_reads(fa, 3000, L, seed=11 + L, pbat_frac=0.5 if mode == 2 else 0.0))
            # (44-46 bases: such reads hash past their end, into what the reads before them left behind -- in input order)
            o_res, o_cig, o_n, work = oracle.map_se(oix, reads, mode=mode, threads=8 if L > 46 else 1)
            assert work["candidates"] > 0
            res, cig, off = ctx.map_se(reads, mode=mode)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"records for {built_for} mode {mode} L {L}")
        # ragged: every length from the shortest mapped read up, in one batch the records serve and one they do not
        rng = np.random.default_rng(3)
        for top in (108, 131):
            pool = synth.mutated_reads(fa, 3000, top, seed=500 + top)
            reads = synth.trim_like_readloader([r[: int(rng.integers(30, top + 1))] for r in pool[:-1]] + [pool[-1]])
            o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=0, threads=1)
            res, cig, off = ctx.map_se(reads, mode=0)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"records for {built_for}, ragged up to {top}")
    finally:
        ctx.close()
        ix.close()


def test_records_beside_an_n_run(oracle, workdir):
    """A record holds code 0 where the genome has an N, as the bit planes do; a candidate within the cutoff there is redone
    on the nibble array, and one beyond the cutoff on the planes is beyond it there too (an N matches nothing)."""
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "nrun_wrec.fa")
    synth.repeat_rich_genome(fa, seed=12, n_chroms=2, chrom_len=400_000)
    idx = os.path.join(workdir, "nrun_wrec.idx")
    oracle.index_build(fa, idx, threads=4)
    chroms = [np.frombuffer(rec.split(b"\n", 1)[1].replace(b"\n", b"").upper(), dtype=np.uint8)
              for rec in open(fa, "rb").read().split(b">")[1:]]
    reads = []
    for ch in chroms:
        mid = len(ch) // 2
        for L in (100, 97, 150):
            for k in range(25):
                for seg in (ch[mid - L - k: mid - k], ch[mid + 3000 + k: mid + 3000 + k + L], ch[50 + k: 50 + k + L] if ch[0] == ord("N") else ch[k: k + L]):
                    s = seg.copy()
                    s[s == ord("C")] = ord("T")
                    reads.append(bytes(s).decode())
                    reads.append(bytes(synth.COMP[seg[::-1]]).decode().replace("C", "T"))
    oix = oracle.index_load(idx)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, threads=4)
    finally:
        oracle.index_free(oix)
    for built_for in (172, 0):
        index = A.Index(idx, window_records=built_for)
        ctx = A.Context(index, 0)
        try:
            assert ctx.filter_on_planes() and (ctx.window_records() != 0) == (built_for != 0)
            res, cig, off = ctx.map_se(reads)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"reads at the edges of N runs, records for {built_for}")
            assert (res["pos"] != 0).mean() > 0.5
        finally:
            ctx.close()
            index.close()


def test_records_with_the_short_read_index(oracle, workdir):
    """window 12 (the reference's --enable-short): reads from 36 bases, whose specific pass covers max(12, L / 2) offsets"""
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "short_wrec.fa")
    synth.repeat_rich_genome(fa, seed=21, n_chroms=2, chrom_len=300_000)
    idx = os.path.join(workdir, "short_wrec.idx")
    A.index_build(fa, idx, 8, window=12)
    oix = oracle.index_load(idx)
    try:
        for built_for in (108, 0):
            index = A.Index(idx, window_records=built_for)
            ctx = A.Context(index, 0)
            try:
                for L in (36, 40, 50, 75):
                    reads = [r if len(r) >= 36 else "" for r in (x.decode().strip("N") for x in synth.mutated_reads(fa, 2000, L, seed=70 + L))]
                    # (threads=1: reads of 36-38 bases hash past their end, into what the reads before them left behind)
                    o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=0, threads=1)
                    res, cig, off = ctx.map_se(reads, mode=0)
                    compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"short-read index, L {L}, records for {built_for}")
            finally:
                ctx.close()
                index.close()
    finally:
        oracle.index_free(oix)
