"""CPU test of the product CLI's FASTQ front end (splitter + parser of abismal_amd/csrc/abm_cli.cpp,
compiled into a small harness): batches must be whole records in file order whatever the batch size,
and every record must come out as ReadLoader would hand it over (src/abismal.cpp:164-201) -- checked
against the independent Python restatement in tests/oracle_binding.py."""
import gzip
import os
import random
import shutil
import subprocess

import pytest

from tests import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    lib = os.path.join(ROOT, "abismal_amd", "libabismal_amd.so")
    if not os.path.exists(lib):
        import abismal_amd
        abismal_amd.build(verbose=False)
    exe = str(tmp_path_factory.mktemp("harness") / "cli_parse_harness")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "cli_parse_harness.cpp"),
           os.path.join(ROOT, "abismal_amd", "csrc", "abm_sim.cpp"), "-o", exe,
           "-L" + os.path.join(ROOT, "abismal_amd"), "-labismal_amd", "-Wl,-rpath," + os.path.join(ROOT, "abismal_amd"),
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def run(exe, path, batch):
    r = subprocess.run([exe, path, str(batch)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    batches, cur = [], None
    for line in r.stdout.split("\n"):
        if line.startswith("#batch"):
            fields = dict(kv.split("=") for kv in line.split()[1:])
            cur = {"first_line": int(fields["first_line"]), "records": int(fields["records"]), "rows": []}
            batches.append(cur)
        elif line and cur is not None:
            name, _, read = line.partition("\t")
            cur["rows"].append((name, read))
    return batches


def write_fastq(path, n, seed, trailing_newline=True):
    rnd = random.Random(seed)
    recs = []
    for i in range(n):
        L = rnd.choice([30, 44, 45, 50, 100, 100, 100, 151])
        seq = "".join(rnd.choice("ACGT") for _ in range(L))
        kind = rnd.random()
        if kind < 0.15:   # N at both ends
            seq = "N" * rnd.randint(1, 6) + seq + "N" * rnd.randint(1, 9)
        elif kind < 0.25:  # too few informative bases
            seq = seq[:20] + "N" * (len(seq) - 20)
        elif kind < 0.30:  # Ns inside
            seq = seq[:10] + "NN" + seq[12:]
        name = "@read%d" % i + rnd.choice(["", " desc 1:N:0", "\tx", "/1"])
        recs.append("%s\n%s\n+\n%s" % (name, seq, "I" * len(seq)))
    text = "\n".join(recs) + ("\n" if trailing_newline else "")
    with open(path, "w") as f:
        f.write(text)


@pytest.mark.parametrize("n,batch,newline", [(1, 1, True), (7, 3, True), (1000, 64, False), (5000, 5000, True),
                                             (5000, 100000, True), (20000, 4096, False)])
def test_batches_are_whole_records_in_order(harness, tmp_path, n, batch, newline):
    fq = str(tmp_path / "r.fq")
    write_fastq(fq, n, seed=n + batch, trailing_newline=newline)
    names, reads = ob.read_fastq_like_readloader(fq)
    got = run(harness, fq, batch)
    assert [b["records"] for b in got] == [min(batch, n - k) for k in range(0, n, batch)]
    line = 0
    for b in got:
        assert b["first_line"] == line and b["records"] == len(b["rows"])
        line += 4 * b["records"]
    flat = [row for b in got for row in b["rows"]]
    assert [r[0] for r in flat] == names
    assert [r[1] for r in flat] == reads


def test_gzip_and_long_lines(harness, tmp_path):
    fq = str(tmp_path / "long.fq")
    with open(fq, "w") as f:  # read names longer than the splitter's 8 KiB scan blocks
        for i in range(300):
            f.write("@%s_%d\n%s\n+\n%s\n" % ("n" * (9000 + i), i, "ACGT" * 30, "I" * 120))
    with open(fq, "rb") as fi, gzip.open(fq + ".gz", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    names, reads = ob.read_fastq_like_readloader(fq)
    for path in (fq, fq + ".gz"):
        flat = [row for b in run(harness, path, 37) for row in b["rows"]]
        assert [r[0] for r in flat] == names and [r[1] for r in flat] == reads
