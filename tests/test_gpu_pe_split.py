"""GPU parity of the paired-end launch forms (abm_ctx_set_pe_split): seed + mate kernels with lists in LDS, with lists
that outgrow LDS inside the seed kernel's staging area (mated from device memory), with a hand-over area that runs
out of room (those pairs are mapped whole), and seeding and mating in one kernel per pair -- all against the oracle."""
import os

import pytest

from tests import oracle_binding as ob
from tests.test_gpu_pe_parity import compare_pe, sim_pairs

pytestmark = pytest.mark.gpu

FORMS = [
    ("unsplit", dict(split=0)),
    ("split", dict(split=1)),
    ("split_stage_1024", dict(split=1, seed_cap=1024)),
    ("split_stage_16384", dict(split=1, seed_cap=16384, hand_entries=24_000_000)),
    ("split_stage_300", dict(split=1, seed_cap=300)),
    ("split_small_area", dict(split=1, seed_cap=16384, hand_entries=200000)),
]


@pytest.fixture(scope="module")
def repeat_rich(oracle, workdir):
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "rep_pe_split.fa")
    idx = os.path.join(workdir, "rep_pe_split.idx")
    synth.repeat_rich_genome(fa)
    A.index_build(fa, idx, 8)
    out = {}
    oix = oracle.index_load(idx)
    try:
        for mode, L in ((0, 100), (1, 150), (2, 125)):
            r1, r2 = synth.mutated_pairs(fa, 3000, L, seed=11 + L)
            r1, r2 = synth.trim_like_readloader(r1), synth.trim_like_readloader(r2)
            out[mode] = (r1, r2, oracle.map_pe(oix, r1, r2, mode=mode, threads=8))
    finally:
        oracle.index_free(oix)
    return idx, out


@pytest.mark.parametrize("form,kw", FORMS)
def test_repeat_rich_by_launch_form(repeat_rich, form, kw):
    import abismal_amd as A
    idx, data = repeat_rich
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    try:
        ctx.set_pe_split(**kw)
        for mode, (r1, r2, orc) in data.items():
            ctx.pe_split_stats()
            gpu = ctx.map_pe(r1, r2, mode=mode)
            compare_pe(gpu, orc, f"repeat-rich PE, {form}, mode {mode}")
            st = ctx.pe_split_stats()
            routed = st["mated_from_lds"] + st["mapped_whole"] + st["mated_from_device_memory"]
            if kw["split"] == 0:
                assert routed == 0
                continue
            assert routed == len(r1), st
            # the repeat-rich genome grows sets to ~2000 entries: without a staging area those pairs are mapped whole, with
            # one of 16384 entries they are mated from device memory (1024: still whole); a small hand-over area sends
            # pairs the whole-pair way whatever the staging area holds
            if kw.get("seed_cap", 0) >= 16384 and kw.get("hand_entries", 0) >= 1_000_000:
                assert st["mated_from_device_memory"] > 0, st
            elif kw.get("seed_cap", 0) <= 128:
                assert st["mated_from_device_memory"] == 0 and st["mapped_whole"] > 0, st
            if kw.get("hand_entries", 1 << 40) < 1_000_000:
                assert st["hand_over_entries_last_batch"] > kw["hand_entries"] and st["mapped_whole"] > 0, st
    finally:
        ctx.close()
        ix.close()


@pytest.mark.parametrize("form,kw", [FORMS[0], FORMS[2]])
def test_trex_by_launch_form(oracle, trex_index, workdir, form, kw):
    """the reference's own genome, 10 k simulated pairs, random PBAT (four orientation calls per pair)"""
    import abismal_amd as A
    r1, r2 = sim_pairs(oracle, workdir, "split_" + form, random_pbat=True)
    oix = oracle.index_load(trex_index)
    try:
        orc = oracle.map_pe(oix, r1, r2, mode=2, threads=8)
    finally:
        oracle.index_free(oix)
    ix = A.Index(trex_index)
    ctx = A.Context(ix, 0)
    try:
        ctx.set_pe_split(**kw)
        compare_pe(ctx.map_pe(r1, r2, mode=2), orc, f"tRex1 PE random PBAT, {form}")
    finally:
        ctx.close()
        ix.close()
