"""GPU vs oracle at scale, inside the test suite: a 400 Mbp hg38-shaped genome (the bench's generator:
repeat families, satellites, microsatellites, N gaps), full comparison of hits AND CIGARs.
 * BASELINE config 2/4 shape: 200 k x 100 bp single-end T-rich;
 * BASELINE config 5: 100 k x 150 bp single-end random-PBAT (-R: both conversions);
 * BASELINE config 3: 50 k pairs 2 x 150 bp."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MBP = float(os.environ.get("ABM_TEST_SCALE_MBP", 400))
COMP = bytes.maketrans(b"ACGT", b"TGCA")


@pytest.fixture(scope="module")
def big(oracle, tmp_path_factory):
    import torch
    import abismal_amd as A
    import bench
    wd = tmp_path_factory.mktemp("scale")
    dev = torch.device("cuda", 0)
    fa, idx = str(wd / "g.fa"), str(wd / "g.idx")
    bench.synth_genome_fasta(fa, MBP, 4321, dev)
    A.index_build(fa, idx, os.cpu_count() or 8)
    os.remove(fa)
    _, starts, gw = bench.read_index_genome(idx)
    ix = A.Index(idx, seed_extension=(-1, -1))  # the library's own choice for a genome of this size (4 + 3 letters)
    ctx = A.Context(ix, 0)
    assert ctx.seed_extension()[0] >= 3
    oix = oracle.index_load(idx)
    yield {"ctx": ctx, "oix": oix, "starts": starts, "gw": gw, "dev": dev, "bench": bench}
    oracle.index_free(oix)
    ctx.close()
    ix.close()


def host_reads(blob, L):
    return [bytes(r) for r in blob.cpu().numpy().reshape(-1, L)]


def test_scale_se_trich_100(oracle, big):
    from tests.test_gpu_se_parity import compare_se
    n, L = 200_000, 100
    reads = host_reads(big["bench"].sample_reads(big["gw"], big["starts"], n, L, 99, big["dev"])[0], L)
    res, cig, off = big["ctx"].map_se(reads, mode=0)
    o_res, o_cig, o_n, work = oracle.map_se(big["oix"], reads, mode=0, threads=os.cpu_count() or 8)
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, "scale SE T-rich 100 bp")
    assert (res["pos"] != 0).mean() > 0.9
    assert work["candidates"] / n > 200, "genome no longer repeat-rich enough to stress the filter"


def test_scale_se_random_pbat_150(oracle, big):
    from tests.test_gpu_se_parity import compare_se
    n, L = 100_000, 150
    t = host_reads(big["bench"].sample_reads(big["gw"], big["starts"], n, L, 4242, big["dev"])[0], L)
    reads = [r.translate(COMP)[::-1] if i & 1 else r for i, r in enumerate(t)]  # half the reads from the PBAT strand
    res, cig, off = big["ctx"].map_se(reads, mode=2)
    o_res, o_cig, o_n, _ = oracle.map_se(big["oix"], reads, mode=2, threads=os.cpu_count() or 8)
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, "scale SE -R 150 bp")
    flags = res["flags"][res["pos"] != 0]
    assert (res["pos"] != 0).mean() > 0.9 and 0.3 < ((flags & 0x1000) != 0).mean() < 0.7


def test_scale_pe_150(oracle, big):
    from tests.test_gpu_pe_parity import compare_pe
    n, L = 50_000, 150
    b1, b2 = big["bench"].sample_pairs(big["gw"], big["starts"], n, L, 777, big["dev"])
    r1, r2 = host_reads(b1, L), host_reads(b2, L)
    gpu = big["ctx"].map_pe(r1, r2, mode=0)
    orc = oracle.map_pe(big["oix"], r1, r2, mode=0, threads=os.cpu_count() or 8)
    compare_pe(gpu, orc, "scale PE 2x150")
    assert (gpu[0]["r1"]["pos"] != 0).mean() > 0.8
