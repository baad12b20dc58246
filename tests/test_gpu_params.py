"""The tunables the path reads (src/abismal.cpp:2329-2339, :2448-2452) varied against the oracle:
-c max_candidates, -m valid_frac, -l/-L fragment range, -a allow_ambig -- through the C ABI, and
through the product CLI against the oracle's CLI (SAM body and statistics byte for byte)."""
import os
import subprocess

import pytest

from tests import oracle_binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "abismal_amd", "abismal-amd")


@pytest.fixture(scope="module")
def rep(oracle, tmp_path_factory):
    import abismal_amd as A
    from tests import synth
    wd = tmp_path_factory.mktemp("params")
    fa, idx = str(wd / "rep.fa"), str(wd / "rep.idx")
    synth.repeat_rich_genome(fa)
    A.index_build(fa, idx, 8)
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    oix = oracle.index_load(idx)
    yield {"fa": fa, "idx": idx, "ctx": ctx, "oix": oix, "wd": wd}
    oracle.index_free(oix)
    ctx.close()
    ix.close()


@pytest.mark.parametrize("maxc,frac", [(20, 0.1), (500, 0.1), (0, 0.05), (0, 0.2), (20, 0.2)])
@pytest.mark.parametrize("mode", [0, 2])
def test_se_params(oracle, rep, maxc, frac, mode):
    import abismal_amd as A
    from tests import synth
    from tests.test_gpu_se_parity import compare_se
    reads = synth.trim_like_readloader(synth.mutated_reads(rep["fa"], 4000, 100, seed=31 + maxc, mut=0.04, pbat_frac=0.5 if mode else 0.0))
    o_res, o_cig, o_n, _ = oracle.map_se(rep["oix"], reads, mode=mode, max_candidates=maxc, valid_frac=frac, threads=8)
    res, cig, off = rep["ctx"].map_se(reads, mode=mode, params=A.Params(max_candidates=maxc, valid_frac=frac))
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"SE -c {maxc} -m {frac} mode {mode}")
    assert (res["pos"] != 0).sum() > 1000


@pytest.mark.parametrize("kw", [
    dict(max_candidates=20), dict(max_candidates=500), dict(valid_frac=0.05), dict(valid_frac=0.2),
    dict(min_frag=100, max_frag=400), dict(allow_ambig=1), dict(allow_ambig=1, valid_frac=0.2, min_frag=100, max_frag=400),
])
def test_pe_params(oracle, rep, kw):
    import abismal_amd as A
    from tests import synth
    from tests.test_gpu_pe_parity import compare_pe
    r1, r2 = synth.mutated_pairs(rep["fa"], 2500, 100, seed=77)
    r1, r2 = synth.trim_like_readloader(r1), synth.trim_like_readloader(r2)
    okw = dict(kw)
    if "allow_ambig" in okw:
        okw["allow_ambig"] = bool(okw["allow_ambig"])
    orc = oracle.map_pe(rep["oix"], r1, r2, mode=0, threads=8, **okw)
    gpu = rep["ctx"].map_pe(r1, r2, mode=0, params=A.Params(**kw))
    compare_pe(gpu, orc, f"PE {kw}")


def _write_fastq(path, reads):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            s = r.decode() if isinstance(r, bytes) else r
            f.write(f"@r{i} x\n{s}\n+\n{'I' * len(s)}\n")


@pytest.mark.parametrize("flags", [["-a"], ["-R"], ["-A"], ["-a", "-R"], ["-c", "20", "-m", "0.2"], ["-a", "-j"]])
def test_cli_se_flags_vs_oracle_cli(oracle, rep, flags):
    from tests import synth
    o_cli = ob.CLI
    fq = str(rep["wd"] / "se.fq")
    if not os.path.exists(fq):
        _write_fastq(fq, synth.mutated_reads(rep["fa"], 5000, 100, seed=5, mut=0.03, pbat_frac=0.5))
    tag = "_".join(x.strip("-") for x in flags)
    outs = {}
    for who, exe in (("gpu", CLI), ("oracle", o_cli)):
        sam, st = str(rep["wd"] / f"{who}_{tag}.sam"), str(rep["wd"] / f"{who}_{tag}.stats")
        r = subprocess.run([exe, "map", "-i", rep["idx"], "-o", sam, "-s", st] + flags + [fq], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        outs[who] = ([ln for ln in open(sam) if not ln.startswith("@PG")], open(st).read())
    assert outs["gpu"][0] == outs["oracle"][0], f"SAM differs with {flags}"
    assert outs["gpu"][1] == outs["oracle"][1], f"statistics differ with {flags}"
    body = [ln for ln in outs["gpu"][0] if not ln.startswith("@")]
    assert len(body) > 2000
    if "-a" in flags:
        assert any(int(ln.split("\t")[1]) & 0x100 for ln in body), "no secondary-flagged (ambiguous) record written"


@pytest.mark.parametrize("flags", [["-a"], ["-l", "100", "-L", "400"], ["-R"], ["-a", "-P", "-m", "0.2"]])
def test_cli_pe_flags_vs_oracle_cli(oracle, rep, flags):
    from tests import synth
    f1, f2 = str(rep["wd"] / "pe_1.fq"), str(rep["wd"] / "pe_2.fq")
    if not os.path.exists(f1):
        r1, r2 = synth.mutated_pairs(rep["fa"], 3000, 100, seed=9)
        _write_fastq(f1, r1)
        _write_fastq(f2, r2)
    tag = "pe_" + "_".join(x.strip("-") for x in flags)
    outs = {}
    for who, exe in (("gpu", CLI), ("oracle", ob.CLI)):
        sam, st = str(rep["wd"] / f"{who}_{tag}.sam"), str(rep["wd"] / f"{who}_{tag}.stats")
        r = subprocess.run([exe, "map", "-i", rep["idx"], "-o", sam, "-s", st] + flags + [f1, f2], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        outs[who] = ([ln for ln in open(sam) if not ln.startswith("@PG")], open(st).read())
    assert outs["gpu"][0] == outs["oracle"][0], f"SAM differs with {flags}"
    assert outs["gpu"][1] == outs["oracle"][1], f"statistics differ with {flags}"


@pytest.mark.parametrize("mode", [0, 2])
def test_short_read_index(oracle, rep, mode):
    """An index built with window 12 (the reference's --enable-short): reads from 36 bases on are mapped,
    the specific-pass limits follow the window (src/abismal.cpp:212-213, :1302-1305), results == oracle --
    including the reads short enough to hash past their end (36-38 bases here)."""
    import numpy as np
    import abismal_amd as A
    from tests import synth
    from tests.test_gpu_se_parity import compare_se
    from tests.test_gpu_pe_parity import compare_pe
    idx = str(rep["wd"] / "rep12.idx")
    if not os.path.exists(idx):
        A.index_build(rep["fa"], idx, 8, window=12)
    rng = np.random.default_rng(5 + mode)
    long_reads = synth.trim_like_readloader(synth.mutated_reads(rep["fa"], 5000, 100, seed=21, mut=0.02, pbat_frac=0.5 if mode else 0.0))
    reads = []
    for r in long_reads:
        if len(r) >= 60 and rng.random() < 0.8:
            r = r[: int(rng.integers(30, 61))]
        reads.append(r if len(r) >= 36 else "")
    ix = A.Index(idx)
    ctx = A.Context(ix, 0)
    oix = oracle.index_load(idx)
    try:
        o_res, o_cig, o_n, _ = oracle.map_se(oix, reads, mode=mode, threads=1)
        res, cig, off = ctx.map_se(reads, mode=mode)
        compare_se(res, cig, off, o_res, o_cig, o_n, reads, f"window-12 index, mode {mode}")
        short = [i for i, r in enumerate(reads) if 36 <= len(r) < 44]
        assert len(short) > 300 and sum(1 for i in short if res[i]["pos"] != 0) > 100, "reads below 44 bases must map with this index"
        r1, r2 = reads[:2000], reads[2000:4000]
        compare_pe(ctx.map_pe(r1, r2, mode=mode), oracle.map_pe(oix, r1, r2, mode=mode, threads=1), "window-12 pairs")
    finally:
        oracle.index_free(oix)
        ctx.close()
        ix.close()
