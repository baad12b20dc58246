"""CPU: the oracle (oracle/) must reproduce all 16 md5 goldens of the reference
(data/md5sum.txt, committed as tests/golden/md5sum.txt) by running the exact
command lines of test_scripts/*.test.  This is what pins the oracle."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def chain(oracle, tmp_path_factory):
    from tests import oracle_binding as ob
    wd = tmp_path_factory.mktemp("oracle_chain")
    os.makedirs(wd / "tests")
    os.symlink(os.path.join(ROOT, "tests", "golden", "tRex1.fa"), wd / "tests" / "tRex1.fa")
    cmds = [
        ["idx", "tests/tRex1.fa", "tests/tRex1.idx"],
        ["sim", "-single", "-seed", "1", "-o", "tests/reads", "-n", "10000", "-m", "0.01", "-b", "0.98", "tests/tRex1.fa"],
        ["sim", "-seed", "1", "-o", "tests/reads_pe", "-n", "10000", "-m", "0.01", "-b", "0.98", "tests/tRex1.fa"],
        ["sim", "-a", "-seed", "1", "-o", "tests/reads_pbat_pe", "-n", "10000", "-m", "0.01", "-b", "0.98", "tests/tRex1.fa"],
        ["sim", "-R", "-seed", "1", "-o", "tests/reads_rpbat_pe", "-n", "10000", "-m", "0.01", "-b", "0.98", "tests/tRex1.fa"],
        ["map", "-s", "tests/reads.mstats", "-o", "tests/reads.sam", "-i", "tests/tRex1.idx", "tests/reads_1.fq"],
        ["map", "-s", "tests/reads_pe.mstats", "-o", "tests/reads_pe.sam", "-i", "tests/tRex1.idx", "tests/reads_pe_1.fq",
         "tests/reads_pe_2.fq"],
        ["map", "-P", "-s", "tests/reads_pbat_pe.mstats", "-o", "tests/reads_pbat_pe.sam", "-i", "tests/tRex1.idx",
         "tests/reads_pbat_pe_1.fq", "tests/reads_pbat_pe_2.fq"],
        ["map", "-P", "-s", "tests/reads_rpbat_pe.mstats", "-o", "tests/reads_rpbat_pe.sam", "-i", "tests/tRex1.idx",
         "tests/reads_rpbat_pe_1.fq", "tests/reads_rpbat_pe_2.fq"],
    ]
    for c in cmds:
        subprocess.run([ob.CLI] + c, cwd=wd, check=True)
    return wd


def test_all_sixteen_goldens(chain):
    bad = []
    n = 0
    for line in open(os.path.join(ROOT, "tests", "golden", "md5sum.txt")):
        want, path = line.split()
        got = hashlib.md5(open(chain / path, "rb").read()).hexdigest()
        n += 1
        if got != want:
            bad.append(path)
    assert n == 16 and not bad, f"oracle output differs from the reference goldens: {bad}"
