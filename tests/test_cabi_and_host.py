"""CPU: the C-ABI library loads and exports every symbol the header declares;
host-side logic (index files, builder, argument checks) behaves like the
reference.  No compute call is made here -- there is no GPU."""
import ctypes
import hashlib
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "abismal_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(abm_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import abismal_amd as A
    lib = A.load_library()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/abismal_amd.h but not exported"
    assert set(A.EXPORTED_SYMBOLS) <= set(names)


def test_no_gpu_means_loud_failure_not_fallback(trex_index):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import abismal_amd as A
    ix = A.Index(trex_index)
    with pytest.raises(A.AbismalAmdError):
        A.Context(ix, 0)
    ix.close()


def test_index_open_rejects_like_the_reference(tmp_path, trex_index):
    import abismal_amd as A
    with pytest.raises(A.AbismalAmdError, match="cannot open input file"):
        A.Index(str(tmp_path / "missing.idx"))
    bad = tmp_path / "bad.idx"
    bad.write_bytes(b"NotAnIndexFile" + b"\0" * 64)
    with pytest.raises(A.AbismalAmdError, match="index file format problem"):
        A.Index(str(bad))
    # wrong seed constant (key_weight) -> the reference's message (src/AbismalIndex.cpp:1000-1003)
    head = open(trex_index, "rb").read(64)
    wrong = tmp_path / "wrong.idx"
    wrong.write_bytes(head[:12] + struct.pack("<I", 24) + head[16:])
    with pytest.raises(A.AbismalAmdError, match="inconsistent k-mer size. Expected: 25, got: 24"):
        A.Index(str(wrong))


def test_index_metadata(trex_index):
    import abismal_amd as A
    ix = A.Index(trex_index)
    assert ix.chrom_names == ["pad_start", "chr1", "chr2", "pad_end"]
    assert ix.chrom_starts.tolist() == [0, 32767, 532767, 1032767, 1065534]
    assert ix.max_candidates == 100
    ix.close()


def test_product_index_builder_matches_golden_and_oracle(tmp_path, oracle):
    import abismal_amd as A
    from tests import synth
    gold = dict(reversed(l.split()) for l in open(os.path.join(ROOT, "tests", "golden", "md5sum.txt")))
    out = tmp_path / "t.idx"
    for threads in (1, 5):
        A.index_build(os.path.join(ROOT, "tests", "golden", "tRex1.fa"), str(out), threads)
        assert hashlib.md5(out.read_bytes()).hexdigest() == gold["tests/tRex1.idx"]
    # a genome with short and long N runs, lower case, repeats: product builder == oracle builder
    fa = tmp_path / "rep.fa"
    synth.repeat_rich_genome(str(fa), seed=11, n_chroms=2, chrom_len=400_000)
    A.index_build(str(fa), str(tmp_path / "p.idx"), 4)
    oracle.index_build(str(fa), str(tmp_path / "o.idx"), threads=2)
    assert hashlib.md5((tmp_path / "p.idx").read_bytes()).hexdigest() == \
        hashlib.md5((tmp_path / "o.idx").read_bytes()).hexdigest()


def test_readloader_rules():
    from tests import synth
    reads = ["N" * 10 + "ACGT" * 12 + "NN", "ACGT" * 10, "NNNN" + "A" * 50, "ACGT" * 11 + "N" * 40]
    out = synth.trim_like_readloader(reads)
    assert out[0] == "ACGT" * 12
    assert out[1] == ""            # 40 informative bases < 44
    assert out[2] == "A" * 50
    assert out[3] == "ACGT" * 11


def test_product_sim_reproduces_the_fastq_goldens(tmp_path):
    """`abismal-amd sim` (host-only) against data/md5sum.txt:1-7, command lines of test_scripts/test_simreads*.test."""
    import subprocess
    cli = os.path.join(ROOT, "abismal_amd", "abismal-amd")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    os.makedirs(tmp_path / "tests")
    os.symlink(os.path.join(ROOT, "tests", "golden", "tRex1.fa"), tmp_path / "tests" / "tRex1.fa")
    common = ["-seed", "1", "-n", "10000", "-m", "0.01", "-b", "0.98", "tests/tRex1.fa"]
    for flags, prefix in ((["-single"], "tests/reads"), ([], "tests/reads_pe"), (["-a"], "tests/reads_pbat_pe"),
                          (["-R"], "tests/reads_rpbat_pe")):
        subprocess.run([cli, "sim"] + flags + ["-o", prefix] + common, cwd=tmp_path, check=True)
    gold = dict(reversed(l.split()) for l in open(os.path.join(ROOT, "tests", "golden", "md5sum.txt")))
    n = 0
    for rel, want in gold.items():
        if rel.endswith(".fq"):
            assert hashlib.md5((tmp_path / rel).read_bytes()).hexdigest() == want, rel
            n += 1
    assert n == 7
