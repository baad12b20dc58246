"""GPU: reads mapped on an `idx -A targets` index (src/AbismalIndex.cpp:83-123, :206-279: everything outside the
regions is masked before indexing, so reads from outside find nothing and reads from inside map as usual) equal
the oracle's on the same index, single-end and paired-end; and `map -g genome.fa` (index built on the fly,
src/abismal.cpp:2439-2446) writes the same SAM and statistics as `map -i` on the prebuilt index."""
import os
import subprocess

import pytest

from tests import oracle_binding as ob
from tests.test_gpu_se_parity import compare_se
from tests.test_gpu_pe_parity import compare_pe

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FA = os.path.join(ROOT, "tests", "golden", "tRex1.fa")
CLI = os.path.join(ROOT, "abismal_amd", "abismal-amd")


def test_reads_on_a_targets_index(oracle, tmp_path):
    import abismal_amd as A
    names = [ln[1:].split()[0] for ln in open(FA) if ln.startswith(">")]
    tf = tmp_path / "targets.bed"
    tf.write_text(f"{names[0]}\t1000\t200000\n{names[0]}\t300000\t420000\n{names[1]}\t5000\t250000\n")
    idx = str(tmp_path / "t.idx")
    A.index_build(FA, idx, 8, targets=str(tf))
    oracle.simulate(FA, str(tmp_path / "se"), 6000, single_end=True, seed=21)
    oracle.simulate(FA, str(tmp_path / "pe"), 3000, seed=22)
    _, reads = ob.read_fastq_like_readloader(str(tmp_path / "se_1.fq"))
    _, r1 = ob.read_fastq_like_readloader(str(tmp_path / "pe_1.fq"))
    _, r2 = ob.read_fastq_like_readloader(str(tmp_path / "pe_2.fq"))
    oix = oracle.index_load(idx)
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    try:
        for mode in (0, 2):
            o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=mode, threads=8)
            res, cig, off = ctx.map_se(reads, mode=mode)
            compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"targets index, SE mode {mode}")
        frac = float((res["pos"] != 0).mean())
        assert 0.2 < frac < 0.9, f"reads from outside the target regions must not map ({frac:.2f} mapped)"
        compare_pe(ctx.map_pe(r1, r2, mode=0), oracle.map_pe(oix, r1, r2, mode=0, threads=8), "targets index, PE")
    finally:
        oracle.index_free(oix)
        ctx.close()
        ix.close()


def test_map_g_equals_map_i(oracle, tmp_path):
    oracle.simulate(FA, str(tmp_path / "r"), 5000, single_end=True, seed=31)
    fq = str(tmp_path / "r_1.fq")
    subprocess.run([CLI, "idx", FA, str(tmp_path / "t.idx")], check=True)
    out = {}
    for tag, how in (("i", ["-i", str(tmp_path / "t.idx")]), ("g", ["-g", FA])):
        sam, st = tmp_path / f"{tag}.sam", tmp_path / f"{tag}.mstats"
        r = subprocess.run([CLI, "map", "-s", str(st), "-o", str(sam)] + how + [fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        out[tag] = ([ln for ln in open(sam) if not ln.startswith("@PG")], open(st).read())
    assert out["i"] == out["g"] and len(out["i"][0]) > 4000
    assert not os.path.exists(str(tmp_path / "g.sam") + ".tmp.idx"), "the temporary index of -g must be removed"
