"""GPU parity: the HIP single-end path (through the C ABI) against the oracle."""
import os

import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def compare_se(res, cig, cig_off, o_res, o_cig, o_cig_n, reads, label=""):
    n = len(reads)
    assert len(res) == n
    bad = []
    for i in range(n):
        a, b = res[i], o_res[i]
        if int(a["pos"]) != int(b["pos"]):
            bad.append((i, "pos", int(a["pos"]), int(b["pos"])))
            continue
        if int(b["pos"]) == 0:
            continue
        if int(a["diffs"]) != int(b["diffs"]) or int(a["flags"]) != int(b["flags"]):
            bad.append((i, "diffs/flags", (int(a["diffs"]), hex(int(a["flags"]))), (int(b["diffs"]), hex(int(b["flags"])))))
            continue
        ca = cig[int(cig_off[i]):int(cig_off[i + 1])].tolist()
        cb = o_cig[i, :int(o_cig_n[i])].tolist()
        if ca != cb:
            bad.append((i, "cigar", ca, cb))
    assert not bad, f"{label}: {len(bad)} of {n} reads differ; first: {bad[:5]}"


@pytest.fixture(scope="module")
def gpu_ctx(trex_index):
    import abismal_amd as A
    ix = A.Index(trex_index)
    ctx = A.Context(ix, 0)
    yield ctx
    ctx.close()
    ix.close()


@pytest.fixture(scope="module")
def trex_se_reads(oracle, workdir):
    prefix = os.path.join(workdir, "reads")
    oracle.simulate(os.path.join(GOLD, "tRex1.fa"), prefix, 10000, single_end=True)
    return ob.read_fastq_like_readloader(prefix + "_1.fq")


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_se_trex_10k(oracle, trex_index, gpu_ctx, trex_se_reads, mode):
    names, reads = trex_se_reads
    oix = oracle.index_load(trex_index)
    try:
        o_res, o_cig, o_cig_n, _ = oracle.map_se(oix, reads, mode=mode, threads=8)
    finally:
        oracle.index_free(oix)
    res, cig, cig_off = gpu_ctx.map_se(reads, mode=mode)
    compare_se(res, cig, cig_off, o_res, o_cig, o_cig_n, reads, f"tRex1 SE mode {mode}")
    mapped = int((res["pos"] != 0).sum())
    seedable = sum(1 for r in reads if r)
    # T-rich reads: nearly all map in T-rich and random mode; read as A-rich they mostly cannot
    if mode in (0, 2):
        assert mapped > 0.85 * seedable
    else:
        assert mapped < 0.5 * seedable


@pytest.fixture(scope="module")
def repeat_setup(oracle, workdir):
    """Repeat-rich genome indexed by the PRODUCT builder (must equal the oracle's)."""
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "rep.fa")
    synth.repeat_rich_genome(fa)
    idx = os.path.join(workdir, "rep.idx")
    A.index_build(fa, idx, 8)
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    oix = oracle.index_load(idx)
    yield fa, idx, ctx, oix
    oracle.index_free(oix)
    ctx.close()
    ix.close()


@pytest.mark.parametrize("mode,L,pbat", [(0, 100, 0.0), (1, 100, 1.0), (2, 150, 0.5), (0, 64, 0.0), (0, 250, 0.0)])
def test_se_repeat_rich(oracle, repeat_setup, mode, L, pbat):
    from tests import synth
    fa, idx, ctx, oix = repeat_setup
    reads = synth.trim_like_readloader(synth.mutated_reads(fa, 6000, L, seed=100 + L + mode, pbat_frac=pbat))
    o_res, o_cig, o_cig_n, work = oracle.map_se(oix, reads, mode=mode, threads=8)
    assert work["search_probes"] > 0, "fixture no longer exercises bucket narrowing"
    res, cig, cig_off = ctx.map_se(reads, mode=mode)
    compare_se(res, cig, cig_off, o_res, o_cig, o_cig_n, reads, f"repeat-rich mode {mode} L {L}")


def test_se_iupac_genome(oracle, workdir):
    """Genome with IUPAC ambiguity codes: a word's mismatch contribution can be negative, so
    admission must follow the reference's running-sum early exit, not the final distance."""
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "iupac.fa")
    synth.repeat_rich_genome(fa, seed=21, n_chroms=2, chrom_len=600_000, iupac=60000)
    idx = os.path.join(workdir, "iupac.idx")
    A.index_build(fa, idx, 8)
    reads = synth.trim_like_readloader(synth.mutated_reads(fa, 6000, 100, seed=5, mut=0.04))
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    assert not ctx.filter_on_planes()  # ambiguity letters: the filter stays on the nibble array
    oix = oracle.index_load(idx)
    try:
        for mode in (0, 2):
            o_res, o_cig, o_cig_n, _ = oracle.map_se(oix, reads, mode=mode, threads=8)
            res, cig, cig_off = ctx.map_se(reads, mode=mode)
            compare_se(res, cig, cig_off, o_res, o_cig, o_cig_n, reads, f"IUPAC genome mode {mode}")
    finally:
        oracle.index_free(oix)
        ctx.close()
        ix.close()
