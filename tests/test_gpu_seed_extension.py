"""GPU parity of the seed-extension tables (abm_ext.hip): whatever number of letters the tables answer for -- none
(bisection from the counters, the reference's find_candidates / find_candidates_three, src/abismal.cpp:1163-1259),
a few, or the hg38-scale maximum of the 2-letter table -- results equal the oracle's, read for read, on the
repeat-rich genome whose buckets need narrowing; and a call whose max_candidates differs from the one the tables
were built for runs without them (never stale answers, never a rebuild inside a mapping call); the explicit rebuild."""
import os

import pytest

from tests.test_gpu_se_parity import compare_se
from tests.test_gpu_pe_parity import compare_pe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rep(oracle, workdir):
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "rep_ext.fa")
    synth.repeat_rich_genome(fa)
    idx = os.path.join(workdir, "rep_ext.idx")
    A.index_build(fa, idx, 8)
    oix = oracle.index_load(idx)
    yield fa, idx, oix
    oracle.index_free(oix)


@pytest.mark.parametrize("letters", [(0, 0), (1, 1), (3, 2), (7, 1), (2, 4)])
def test_se_any_table_depth_equals_the_oracle(oracle, rep, letters):
    import abismal_amd as A
    from tests import synth
    fa, idx, oix = rep
    ix = A.Index(idx, seed_extension=letters)
    ctx = A.Context(ix, 0)
    try:
        got = ctx.seed_extension()
        assert got[:2] == letters or letters == (0, 0) and got[:2] == (0, 0), got
        for mode, L in ((0, 100), (2, 150), (0, 66), (0, 48)):  # (48: reads too short for the tables take the counters)
            reads = synth.trim_like_readloader(synth.mutated_reads(fa, 4000, L, seed=7 + L, pbat_frac=0.5 if mode == 2 else 0.0))
            o_res, o_cig, o_n, work = oracle.map_se(oix, reads, mode=mode, threads=8)
            assert work["search_probes"] > 0
            res, cig, off = ctx.map_se(reads, mode=mode)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"tables {letters} mode {mode} L {L}")
    finally:
        ctx.close()
        ix.close()


def test_max_candidates_other_than_the_tables(oracle, rep):
    import abismal_amd as A
    from tests import synth
    fa, idx, oix = rep
    ix = A.Index(idx, seed_extension=(3, 2))
    ctx = A.Context(ix, 0)
    try:
        reads = synth.trim_like_readloader(synth.mutated_reads(fa, 4000, 100, seed=99))
        for c, rebuild in ((20, False), (500, True), (100, False), (5, True), (20, False)):
            # (a call never rebuilds the tables itself; with tables built for another value it bisects from
            # the counters, and abm_ctx_rebuild_seed_extension is the set-up call that rebuilds them)
            if rebuild:
                ctx.rebuild_seed_extension(c)
            p = A.Params(max_candidates=c)
            o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=0, threads=8, max_candidates=c)
            res, cig, off = ctx.map_se(reads, mode=0, params=p)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"-c {c}, tables {'rebuilt' if rebuild else 'of another value: bypassed'}")
        # a second context on the device: a differing max_candidates now runs without tables (nothing is rebuilt under it)
        ctx2 = A.Context(ix, 0)
        try:
            p = A.Params(max_candidates=33)
            o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=0, threads=8, max_candidates=33)
            res, cig, off = ctx2.map_se(reads, mode=0, params=p)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, "second context, -c 33, no tables")
            with pytest.raises(Exception):
                ctx2.rebuild_seed_extension(33)  # not beside another context
        finally:
            ctx2.close()
    finally:
        ctx.close()
        ix.close()


def test_pe_with_and_without_tables(oracle, rep):
    import abismal_amd as A
    from tests import synth
    fa, idx, oix = rep
    r1, r2 = synth.mutated_pairs(fa, 3000, 100, seed=3)
    r1, r2 = synth.trim_like_readloader(r1), synth.trim_like_readloader(r2)  # (the boundary takes reads as ReadLoader hands them over)
    orc = oracle.map_pe(oix, r1, r2, mode=0, threads=8)
    for letters in ((0, 0), (3, 2)):
        ix = A.Index(idx, seed_extension=letters)
        ctx = A.Context(ix, 0)
        try:
            compare_pe(ctx.map_pe(r1, r2, mode=0), orc, f"PE tables {letters}")
        finally:
            ctx.close()
            ix.close()


@pytest.mark.parametrize("min_entries", [0, 16, 64, 100000])
def test_pe_direct_narrowing_any_threshold(oracle, rep, min_entries):
    """abm_index_set_direct_narrowing: whichever ranges the pair kernels narrow directly (none, nearly all, only
    the big ones, none because the threshold is never reached) -- with and without seed-extension tables in front of
    it -- the pairs, fallback hits and CIGARs equal the oracle's."""
    import abismal_amd as A
    from tests import synth
    fa, idx, oix = rep
    r1, r2 = synth.mutated_pairs(fa, 3000, 100, seed=13)
    r1, r2 = synth.trim_like_readloader(r1), synth.trim_like_readloader(r2)
    orc = oracle.map_pe(oix, r1, r2, mode=0, threads=8)
    for letters in ((0, 0), (3, 2)):
        ix = A.Index(idx, seed_extension=letters)
        ix.set_direct_narrowing(min_entries)
        ctx = A.Context(ix, 0)
        try:
            compare_pe(ctx.map_pe(r1, r2, mode=0), orc, f"PE direct narrowing from {min_entries} entries, tables {letters}")
            if min_entries == 16:  # (also at -c 20: ranges stay big for longer)
                p = A.Params(max_candidates=20)
                orc20 = oracle.map_pe(oix, r1, r2, mode=0, threads=8, max_candidates=20)
                compare_pe(ctx.map_pe(r1, r2, mode=0, params=p), orc20, f"PE direct narrowing from 16 entries, -c 20, tables {letters}")
        finally:
            ctx.close()
            ix.close()


@pytest.mark.parametrize("min_entries", [0, 16, 64, 100000])
def test_se_direct_narrowing_any_threshold(oracle, rep, min_entries):
    """The single-end kernel narrows big ranges directly as well (round 5): whichever ranges it takes that way, with and
    without seed-extension tables in front, with and without window records behind -- reads of 100 and 150 bases in the
    T-rich and random-PBAT modes, and at -c 20, equal the oracle's."""
    import abismal_amd as A
    from tests import synth
    fa, idx, oix = rep
    sets = []
    for mode, L in ((0, 100), (2, 150), (0, 60)):
        reads = synth.trim_like_readloader(synth.mutated_reads(fa, 3000, L, seed=31 + L, pbat_frac=0.5 if mode == 2 else 0.0))
        sets.append((mode, L, reads, oracle.map_se(oix, reads, mode=mode, threads=8)[:3]))
    for letters, wrec in (((0, 0), 172), ((3, 2), 172), ((3, 2), 0)):
        ix = A.Index(idx, seed_extension=letters, window_records=wrec)
        ix.set_direct_narrowing(min_entries)
        ctx = A.Context(ix, 0)
        try:
            for mode, L, reads, (o_res, o_cig, o_n) in sets:
                res, cig, off = ctx.map_se(reads, mode=mode)
                compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"SE direct narrowing from {min_entries} entries, tables {letters}, records {wrec}, mode {mode} L {L}")
            if min_entries == 16:  # (also at -c 20: ranges stay big for longer)
                mode, L, reads, _ = sets[0]
                o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=mode, threads=8, max_candidates=20)
                res, cig, off = ctx.map_se(reads, mode=mode, params=A.Params(max_candidates=20))
                compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"SE direct narrowing from 16 entries, -c 20, tables {letters}, records {wrec}")
        finally:
            ctx.close()
            ix.close()
