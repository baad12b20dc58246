"""CPU: `idx -A targets` (src/AbismalIndex.cpp:83-123, :206-279).  The reference holds no golden for it, so the
product's builder is pinned to the oracle's (md5 of the whole index file), with region lists that exercise the
reference's rules: chromosome order taken from the genome, unknown names dropped, unsorted regions rejected,
and the base just past a region's end surviving the mask."""
import hashlib
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FA = os.path.join(ROOT, "tests", "golden", "tRex1.fa")


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def chrom_names():
    return [ln[1:].split()[0] for ln in open(FA) if ln.startswith(">")]


@pytest.mark.parametrize("case", ["two_chroms", "reordered_and_unknown", "touching"])
def test_targets_index_equals_oracle(oracle, tmp_path, case):
    import abismal_amd as A
    c1, c2 = chrom_names()[:2]
    regions = {
        "two_chroms": [(c1, 1000, 60000), (c1, 200000, 260000), (c2, 5000, 90000)],
        # the genome's chromosome order decides, not the file's; names the genome lacks are ignored
        "reordered_and_unknown": [(c2, 100, 50000), ("chrNope", 1, 1000), (c1, 40000, 41000), (c1, 41000, 120000)],
        "touching": [(c1, 0, 30000), (c1, 30000, 30001), (c1, 30002, 80000), (c2, 499000, 500000)],
    }[case]
    tf = tmp_path / "targets.bed"
    tf.write_text("".join(f"{c}\t{a}\t{b}\n" for c, a, b in regions))
    A.index_build(FA, str(tmp_path / "p.idx"), 5, targets=str(tf))
    oracle.index_build(FA, str(tmp_path / "o.idx"), threads=1, targets=str(tf))
    assert md5(tmp_path / "p.idx") == md5(tmp_path / "o.idx")
    # and it is a different (much smaller) index than the whole genome's
    A.index_build(FA, str(tmp_path / "whole.idx"), 5)
    assert md5(tmp_path / "p.idx") != md5(tmp_path / "whole.idx")


def test_unsorted_targets_are_rejected(oracle, tmp_path):
    import abismal_amd as A
    c1 = chrom_names()[0]
    tf = tmp_path / "bad.bed"
    tf.write_text(f"{c1}\t5000\t6000\n{c1}\t100\t200\n")
    with pytest.raises(A.AbismalAmdError, match="target regions not sorted"):
        A.index_build(FA, str(tmp_path / "p.idx"), 2, targets=str(tf))
    with pytest.raises(RuntimeError, match="target regions not sorted"):
        oracle.index_build(FA, str(tmp_path / "o.idx"), threads=1, targets=str(tf))
    tf.write_text(f"{c1}\t5000\n")
    with pytest.raises(A.AbismalAmdError, match="failed parsing target region"):
        A.index_build(FA, str(tmp_path / "p.idx"), 2, targets=str(tf))


def test_short_read_index_equals_oracle(oracle, tmp_path):
    """Window 12 (the reference's --enable-short build, src/AbismalIndex.hpp:73-77): product builder == oracle
    builder, the file says 12, and both loaders take it."""
    import struct
    import abismal_amd as A
    A.index_build(FA, str(tmp_path / "p12.idx"), 5, window=12)
    oracle.index_build(FA, str(tmp_path / "o12.idx"), threads=1, window=12)
    assert md5(tmp_path / "p12.idx") == md5(tmp_path / "o12.idx")
    with open(tmp_path / "p12.idx", "rb") as f:
        assert f.read(12) == b"AbismalIndex" and struct.unpack("<3I", f.read(12)) == (25, 12, 256)
    ix = A.Index(str(tmp_path / "p12.idx"))
    assert ix.window == 12
    ix.close()
    oix = oracle.index_load(str(tmp_path / "o12.idx"))
    oracle.index_free(oix)
    # more positions are kept than with window 20 (a denser selection)
    A.index_build(FA, str(tmp_path / "p20.idx"), 5)
    assert os.path.getsize(tmp_path / "p12.idx") > os.path.getsize(tmp_path / "p20.idx")
