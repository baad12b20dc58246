"""End-to-end through the product CLI on the GPU: the reference's own regression
chain (test_scripts/test_abismal*.test) must reproduce data/md5sum.txt byte for
byte -- SAM and statistics files -- with the index built by the product's
indexer and the reads produced by the (md5-pinned) simulator restatement."""
import hashlib
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "abismal_amd", "abismal-amd")


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def golden():
    g = {}
    for line in open(os.path.join(ROOT, "tests", "golden", "md5sum.txt")):
        h, p = line.split()
        g[p] = h
    return g


@pytest.fixture(scope="module")
def chain(oracle, tmp_path_factory):
    wd = tmp_path_factory.mktemp("chain")
    os.makedirs(wd / "tests")
    fa = os.path.join(ROOT, "tests", "golden", "tRex1.fa")
    os.symlink(fa, wd / "tests" / "tRex1.fa")
    subprocess.run([CLI, "idx", "tests/tRex1.fa", "tests/tRex1.idx"], cwd=wd, check=True)
    for prefix, kw in (("tests/reads", {"single_end": True}), ("tests/reads_pe", {}),
                       ("tests/reads_pbat_pe", {"pbat": True}), ("tests/reads_rpbat_pe", {"random_pbat": True})):
        oracle.simulate(fa, str(wd / prefix), 10000, **kw)
    return wd


def test_index_md5(chain):
    assert md5(chain / "tests" / "tRex1.idx") == golden()["tests/tRex1.idx"]


@pytest.mark.parametrize("args,outs", [
    (["-s", "tests/reads.mstats", "-o", "tests/reads.sam", "-i", "tests/tRex1.idx", "tests/reads_1.fq"],
     ["tests/reads.sam", "tests/reads.mstats"]),
    (["-s", "tests/reads_pe.mstats", "-o", "tests/reads_pe.sam", "-i", "tests/tRex1.idx", "tests/reads_pe_1.fq",
      "tests/reads_pe_2.fq"], ["tests/reads_pe.sam", "tests/reads_pe.mstats"]),
    (["-P", "-s", "tests/reads_pbat_pe.mstats", "-o", "tests/reads_pbat_pe.sam", "-i", "tests/tRex1.idx",
      "tests/reads_pbat_pe_1.fq", "tests/reads_pbat_pe_2.fq"], ["tests/reads_pbat_pe.sam", "tests/reads_pbat_pe.mstats"]),
    (["-P", "-s", "tests/reads_rpbat_pe.mstats", "-o", "tests/reads_rpbat_pe.sam", "-i", "tests/tRex1.idx",
      "tests/reads_rpbat_pe_1.fq", "tests/reads_rpbat_pe_2.fq"], ["tests/reads_rpbat_pe.sam", "tests/reads_rpbat_pe.mstats"]),
])
def test_map_goldens(chain, args, outs):
    r = subprocess.run([CLI, "map"] + args, cwd=chain, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    g = golden()
    for o in outs:
        assert md5(chain / o) == g[o], f"{o} differs from the reference golden"
