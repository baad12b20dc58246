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
    # the reads come from the product's own `sim` (md5-pinned on the CPU side by test_cabi_and_host.py)
    common = ["-seed", "1", "-n", "10000", "-m", "0.01", "-b", "0.98", "tests/tRex1.fa"]
    for flags, prefix in ((["-single"], "tests/reads"), ([], "tests/reads_pe"), (["-a"], "tests/reads_pbat_pe"),
                          (["-R"], "tests/reads_rpbat_pe")):
        subprocess.run([CLI, "sim"] + flags + ["-o", prefix] + common, cwd=wd, check=True)
    return wd


def test_index_and_fastq_md5(chain):
    g = golden()
    for rel in ("tests/tRex1.idx", "tests/reads_1.fq", "tests/reads_pe_1.fq", "tests/reads_pe_2.fq",
                "tests/reads_pbat_pe_1.fq", "tests/reads_pbat_pe_2.fq", "tests/reads_rpbat_pe_1.fq",
                "tests/reads_rpbat_pe_2.fq"):
        assert md5(chain / rel) == g[rel], rel


def test_gzip_input_gives_the_same_sam(chain):
    import gzip
    import shutil
    with open(chain / "tests/reads_1.fq", "rb") as fi, gzip.open(chain / "tests/reads_1.fq.gz", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    subprocess.run([CLI, "map", "-o", "tests/gz.sam", "-i", "tests/tRex1.idx", "tests/reads_1.fq.gz"], cwd=chain, check=True)
    subprocess.run([CLI, "map", "-o", "tests/plain.sam", "-i", "tests/tRex1.idx", "tests/reads_1.fq"], cwd=chain, check=True)
    body = lambda p: [l for l in open(chain / p) if not l.startswith("@PG")]
    assert body("tests/gz.sam") == body("tests/plain.sam") and len(body("tests/gz.sam")) > 8000


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
@pytest.mark.parametrize("units", [None, ("997", "4096", "7"), ("64", "1000", "1")])
def test_map_goldens(chain, args, outs, units):
    env = dict(os.environ)
    if units:  # shrink slices / chunks / line marks so that the 10 k-read fixtures cross many of each
        env.update(ABM_CLI_SLICE_READS=units[0], ABM_CLI_CHUNK_BYTES=units[1], ABM_CLI_MARK_LINES=units[2],
                   ABM_CLI_BATCH_READS="3000")  # (not -batch: the SAM's @PG line carries the command line)
    r = subprocess.run([CLI, "map"] + args, cwd=chain, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    g = golden()
    for o in outs:
        assert md5(chain / o) == g[o], f"{o} differs from the reference golden"


@pytest.mark.parametrize("flags", [[], ["-a"], ["-R"], ["-A", "-a"]])
def test_sam_text_from_the_device_equals_the_hosts(chain, flags):
    """Single-end SAM text is written by the mapping kernel itself when a GPU has few host workers to itself
    (abm_ctx_set_sam_tails; ABM_CLI_DEVICE_SAM=1 forces it here): the file must be the one the host's formatter writes, and
    for the reference's own command line the reference's golden.  Reads with IUPAC letters (SEQ shows them
    as they are: htslib has 4-bit codes for them), reads of 44-46 bases, reads too short to map and reads whose
    CIGAR outgrows its 4-op slot (formatted by the host all the same) ride along in a second file."""
    g = golden()
    outs = {}
    for by in ("1", "0"):
        env = dict(os.environ, ABM_CLI_DEVICE_SAM=by, ABM_CLI_SLICE_READS="997")
        # (the reference's own command line, to the letter: the SAM's @PG line carries it)
        r = subprocess.run([CLI, "map"] + flags + ["-s", "tests/reads.mstats", "-o", "tests/reads.sam", "-i", "tests/tRex1.idx", "tests/reads_1.fq"],
                           cwd=chain, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        outs[by] = (open(chain / "tests/reads.sam").read(), open(chain / "tests/reads.mstats").read())
        if not flags:
            assert md5(chain / "tests/reads.sam") == g["tests/reads.sam"], f"SAM text by {'device' if by == '1' else 'host'} differs from the reference golden"
            assert md5(chain / "tests/reads.mstats") == g["tests/reads.mstats"]
    # (-A maps the simulator's T-rich reads as A-rich: few of them find a hit)
    assert outs["1"] == outs["0"] and outs["1"][0].count("\n") > (400 if "-A" in flags else 8000)
    # the odd reads
    lines = open(chain / "tests/reads_1.fq").read().split("\n")
    import random
    rng = random.Random(5)
    for k in range(0, len(lines) - 3, 4):
        seq = lines[k + 1]
        kind = (k // 4) % 7
        if kind == 1:
            seq = seq[:44 + (k // 28) % 3]
        elif kind == 2:
            j = rng.randrange(len(seq)); seq = seq[:j] + "RYKMSWN"[rng.randrange(7)] + seq[j + 1:]
        elif kind == 3:
            j = rng.randrange(5, len(seq) - 25)  # three small deletions and an insertion: five or more CIGAR ops
            seq = seq[:j] + seq[j + 2:j + 10] + "A" + seq[j + 10:j + 18] + seq[j + 19:j + 30] + seq[j + 32:]
        elif kind == 4:
            seq = seq[:30]
        lines[k + 1], lines[k + 3] = seq, lines[k + 3][:len(seq)].ljust(len(seq), "B")
    open(chain / "tests/odd.fq", "w").write("\n".join(lines))
    outs = {}
    for by in ("1", "0"):
        env = dict(os.environ, ABM_CLI_DEVICE_SAM=by, ABM_CLI_SLICE_READS="997")
        r = subprocess.run([CLI, "map"] + flags + ["-s", "tests/odd.mstats", "-o", "tests/odd.sam", "-i", "tests/tRex1.idx", "tests/odd.fq"],
                           cwd=chain, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        outs[by] = (open(chain / "tests/odd.sam").read(), open(chain / "tests/odd.mstats").read())
    assert outs["1"] == outs["0"] and outs["1"][0].count("\n") > (200 if "-A" in flags else 5000)
    if "-A" in flags:
        return
    assert sum(1 for ln in outs["1"][0].split("\n") if not ln.startswith("@") and ln and len(__import__("re").findall(r"[MIDS]", ln.split("\t")[5])) >= 5) > 100


@pytest.mark.parametrize("reads,extra", [
    (["tests/reads_1.fq"], []),
    (["tests/reads_pe_1.fq", "tests/reads_pe_2.fq"], []),
    (["tests/reads_rpbat_pe_1.fq", "tests/reads_rpbat_pe_2.fq"], ["-R"]),
])
def test_many_small_batches_keep_input_order(chain, reads, extra):
    """Batches of 700 records over three mapper contexts and several host threads: records must leave in
    input order and the statistics must add up exactly as in a single-batch run."""
    def run(tag, flags):
        r = subprocess.run([CLI, "map"] + extra + flags + ["-s", f"tests/{tag}.stats", "-o", f"tests/{tag}.sam", "-i",
                           "tests/tRex1.idx"] + reads, cwd=chain, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
        return ([l for l in open(chain / f"tests/{tag}.sam") if not l.startswith("@PG")], open(chain / f"tests/{tag}.stats").read())
    one = run("one_batch", [])
    many = run("many_batches", ["-batch", "700", "-mappers", "3", "-t", "5"])
    assert many[0] == one[0]
    assert many[1] == one[1]
    assert len(one[0]) > 8000


def test_bam_output_carries_the_same_records(chain):
    """-B: decode the BGZF/BAM stream with nothing but gzip + struct and compare every field with the SAM text."""
    import gzip
    import struct
    for args, tag in ((["tests/reads_1.fq"], "se"), (["tests/reads_pe_1.fq", "tests/reads_pe_2.fq"], "pe")):
        subprocess.run([CLI, "map", "-B", "-o", f"tests/{tag}.bam", "-i", "tests/tRex1.idx"] + args, cwd=chain, check=True)
        subprocess.run([CLI, "map", "-o", f"tests/{tag}.sam", "-i", "tests/tRex1.idx"] + args, cwd=chain, check=True)
        raw = open(chain / f"tests/{tag}.bam", "rb").read()
        assert raw.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
        data = gzip.decompress(raw)
        assert data[:4] == b"BAM\x01"
        (l_text,) = struct.unpack_from("<I", data, 4)
        text = data[8:8 + l_text].decode()
        at = 8 + l_text
        (n_ref,) = struct.unpack_from("<I", data, at); at += 4
        refs = []
        for _ in range(n_ref):
            (ln,) = struct.unpack_from("<I", data, at); at += 4
            refs.append(data[at:at + ln - 1].decode()); at += ln + 4
        assert refs == ["chr1", "chr2"] and "@SQ\tSN:chr1\tLN:500000" in text
        recs = []
        while at < len(data):
            (bs,) = struct.unpack_from("<I", data, at); at += 4
            tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHIiii", data, at)
            p = at + 32
            name = data[p:p + l_name - 1].decode(); p += l_name
            cig = "".join(f"{c >> 4}{'MIDNSHP=XB'[c & 15]}" for c in struct.unpack_from(f"<{n_cig}I", data, p)); p += 4 * n_cig
            sq = data[p:p + (l_seq + 1) // 2]; p += (l_seq + 1) // 2
            seq = "".join("=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
            assert data[p:p + l_seq] == b"\xff" * l_seq; p += l_seq
            aux = data[p:at + bs]
            assert aux[:3] == b"NMC" and aux[4:7] == b"CVA"
            recs.append((name, flag, refs[tid], pos + 1, mapq, cig, "*" if mtid < 0 else "=", mpos + 1, tlen, seq, aux[3], chr(aux[7])))
            at += bs
        sam = []
        for line in open(chain / f"tests/{tag}.sam"):
            if line.startswith("@"):
                continue
            f = line.rstrip("\n").split("\t")
            sam.append((f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[6], int(f[7]), int(f[8]), f[9], int(f[11][5:]), f[12][5:]))
        assert recs == sam and len(recs) > 8000
