"""GPU parity: the HIP paired-end path (through the C ABI) against the oracle."""
import os

import numpy as np
import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hit_tuple(h):
    return (int(h["diffs"]), int(h["flags"]), int(h["pos"]))


def compare_pe(gpu, orc, label):
    pairs, se1, se2, (c1, co1), (c2, co2) = gpu
    o_pairs, o_se1, o_se2, (oc1, on1), (oc2, on2), _ = orc
    n = len(pairs)
    bad = []
    for i in range(n):
        gp, op = pairs[i], o_pairs[i]
        g_has, o_has = int(gp["r1"]["pos"]) != 0, int(op["r1"]["pos"]) != 0
        if g_has != o_has:
            bad.append((i, "pair presence", g_has, o_has))
            continue
        cig_g1 = c1[int(co1[i]):int(co1[i + 1])].tolist()
        cig_g2 = c2[int(co2[i]):int(co2[i + 1])].tolist()
        cig_o1 = oc1[i, :int(on1[i])].tolist()
        cig_o2 = oc2[i, :int(on2[i])].tolist()
        if o_has:
            a = (int(gp["aln_score"]), hit_tuple(gp["r1"]), hit_tuple(gp["r2"]))
            b = (int(op["aln_score"]), hit_tuple(op["r1"]), hit_tuple(op["r2"]))
            if a != b:
                bad.append((i, "pair", a, b))
            elif not (int(op["r1"]["flags"]) & 0x100) and (cig_g1 != cig_o1 or cig_g2 != cig_o2):
                bad.append((i, "pair cigar", (cig_g1, cig_g2), (cig_o1, cig_o2)))
        # single-end results (only meaningful when the oracle produced them)
        for which, gs, os_, cg, co in ((1, se1[i], o_se1[i], cig_g1, cig_o1), (2, se2[i], o_se2[i], cig_g2, cig_o2)):
            if int(gs["pos"]) != int(os_["pos"]):
                bad.append((i, f"se{which} pos", hit_tuple(gs), hit_tuple(os_)))
            elif int(os_["pos"]) != 0:
                if hit_tuple(gs) != hit_tuple(os_):
                    bad.append((i, f"se{which}", hit_tuple(gs), hit_tuple(os_)))
                elif not o_has and cg != co:
                    bad.append((i, f"se{which} cigar", cg, co))
    assert not bad, f"{label}: {len(bad)} of {n} pairs differ; first: {bad[:4]}"


@pytest.fixture(scope="module")
def gpu_ctx(trex_index):
    import abismal_amd as A
    ix = A.Index(trex_index)
    ctx = A.Context(ix, 0)
    yield ctx
    ctx.close()
    ix.close()


def sim_pairs(oracle, workdir, tag, n=10000, **kw):
    prefix = os.path.join(workdir, "pe_" + tag)
    oracle.simulate(os.path.join(GOLD, "tRex1.fa"), prefix, n, **kw)
    _, r1 = ob.read_fastq_like_readloader(prefix + "_1.fq")
    _, r2 = ob.read_fastq_like_readloader(prefix + "_2.fq")
    return r1, r2


@pytest.mark.parametrize("tag,simkw,mode", [
    ("normal", {}, 0),
    ("pbat", {"pbat": True}, 1),
    ("rpbat_as_pbat", {"random_pbat": True}, 1),   # the reference's own rpbat test maps with -P
    ("rpbat_random", {"random_pbat": True}, 2),
    ("long150", {"read_len": 150, "min_frag": 150, "max_frag": 500}, 0),
])
def test_pe_trex(oracle, trex_index, gpu_ctx, workdir, tag, simkw, mode):
    r1, r2 = sim_pairs(oracle, workdir, tag, **simkw)
    oix = oracle.index_load(trex_index)
    try:
        orc = oracle.map_pe(oix, r1, r2, mode=mode, threads=8)
    finally:
        oracle.index_free(oix)
    gpu = gpu_ctx.map_pe(r1, r2, mode=mode)
    compare_pe(gpu, orc, f"tRex1 PE {tag}")
    assert int((gpu[0]["r1"]["pos"] != 0).sum()) > 0.3 * len(r1)


@pytest.mark.parametrize("mode,L", [(0, 100), (2, 125)])
def test_pe_repeat_rich(oracle, workdir, mode, L):
    """Repeat-rich genome: candidate sets grow past tier 1 (256) so tier 2 runs too."""
    import abismal_amd as A
    from tests import synth
    fa = os.path.join(workdir, "rep_pe.fa")
    idx = os.path.join(workdir, "rep_pe.idx")
    if not os.path.exists(idx):
        synth.repeat_rich_genome(fa)
        A.index_build(fa, idx, 8)
    r1, r2 = synth.mutated_pairs(fa, 3000, L, seed=7 + L)
    r1, r2 = synth.trim_like_readloader(r1), synth.trim_like_readloader(r2)
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    oix = oracle.index_load(idx)
    try:
        orc = oracle.map_pe(oix, r1, r2, mode=mode, threads=8)
        gpu = ctx.map_pe(r1, r2, mode=mode)
    finally:
        oracle.index_free(oix)
        ctx.close()
        ix.close()
    compare_pe(gpu, orc, f"repeat-rich PE mode {mode} L {L}")
