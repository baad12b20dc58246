"""CPU: the host pipeline of `abismal-amd map` around N GPUs, run with virtual GPUs (-virtual-gpus N: no device, every
read gets a made-up hit that depends on the read alone).  What is checked is everything the multi-GPU CLI path does on
the host: counting / cutting / parsing, the dealing of batches to per-GPU mapper threads, the lead-in a batch carries
for 44-46-base reads, results handed over slice by slice, the ordered merge, per-region output files (-out-parts) whose
concatenation is the one-file output, and the statistics sum."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "abismal_amd", "abismal-amd")
SMALL = dict(ABM_CLI_SLICE_READS="1024", ABM_CLI_FIRST_BATCH="1024", ABM_CLI_CHUNK_BYTES="65536", ABM_CLI_MARK_LINES="64")


def run(args, env=None, **kw):
    r = subprocess.run([CLI, "map"] + [str(a) for a in args], env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600, **kw)
    assert r.returncode == 0, r.stderr
    return r


def body(paths):
    out = []
    for p in paths:
        out += [ln for ln in open(p) if not ln.startswith("@PG")]
    return out


@pytest.fixture(scope="module")
def reads(oracle, tmp_path_factory):
    d = tmp_path_factory.mktemp("vgpu")
    oracle.simulate(os.path.join(ROOT, "tests", "golden", "tRex1.fa"), str(d / "r"), 60000, single_end=True, seed=21)
    return str(d / "r_1.fq"), d


def test_virtual_gpus_write_what_one_gpu_writes(reads, trex_index):
    fq, d = reads
    run(["-virtual-gpus", 1, "-t", 3, "-i", trex_index, "-o", d / "one.sam", "-s", d / "one.st", fq])
    ref = body([d / "one.sam"])
    assert len(ref) > 50000
    for gpus, parts, extra in ((4, 1, []), (4, 4, []), (8, 8, ["-mappers", 1]), (2, 4, []), (3, 1, ["-batch", 5000])):
        out = d / f"g{gpus}p{parts}.sam"
        run(["-virtual-gpus", gpus, "-out-parts", parts, "-batch", 4096, "-t", 6, "-timing", d / "t.json", "-i", trex_index, "-o", out,
             "-s", d / "x.st", fq] + extra, env=SMALL)
        files = [out] if parts == 1 else [f"{out}.part{k:03d}" for k in range(parts)]
        assert body(files) == ref, (gpus, parts)
        assert open(d / "x.st").read() == open(d / "one.st").read()
        t = json.load(open(d / "t.json"))
        assert t["out_parts"] == parts and t["gpus"] == gpus and sum(t["reads_per_gpu"]) == 60000
        if parts >= gpus:  # (a region of its own: every "GPU" maps; GPUs that share a region race for its batches, and a
            assert min(t["batches_per_gpu"]) >= 1, t  # virtual GPU is done with one at once)
        for f in files:
            os.remove(f)


def test_whole_batches_equal_slices(reads, trex_index):
    fq, d = reads
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", d / "s.sam", fq], env=SMALL)
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", d / "w.sam", fq], env=dict(SMALL, ABM_CLI_NO_STREAM="1"))
    assert body([d / "s.sam"]) == body([d / "w.sam"])


def test_pread_path_equals_mapped_input(reads, trex_index):
    # plain files are parsed in place from a read-only mapping; ABM_CLI_NO_MMAP=1 (and any file that cannot be mapped)
    # takes the pread path with per-slice text buffers
    fq, d = reads
    run(["-virtual-gpus", 2, "-out-parts", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", d / "m.sam", fq], env=SMALL)
    run(["-virtual-gpus", 2, "-out-parts", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", d / "p.sam", fq], env=dict(SMALL, ABM_CLI_NO_MMAP="1"))
    assert body([f"{d}/m.sam.part000", f"{d}/m.sam.part001"]) == body([f"{d}/p.sam.part000", f"{d}/p.sam.part001"])


def _write_bgzf(src, dst, block=0xff00):
    import struct
    import zlib
    with open(src, "rb") as f, open(dst, "wb") as o:
        while True:
            d = f.read(block)
            if not d:
                break
            co = zlib.compressobj(1, zlib.DEFLATED, -15)
            z = co.compress(d) + co.flush()
            o.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(z) + 25) + z + struct.pack("<II", zlib.crc32(d), len(d)))
        o.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))


def test_bgzf_input_is_inflated_block_parallel(reads, trex_index):
    # bgzip-compressed FASTQ: the workers inflate its blocks side by side (single-member gzip: one inflating thread);
    # same output as the plain file, also with blocks of odd sizes, chunks of a few blocks, and the inflated text held
    # to a window of 300 kB ahead of the writer (the mappers then take what is there instead of waiting for a batch)
    import gzip
    fq, d = reads
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", d / "plain.sam", fq], env=SMALL)
    ref = body([d / "plain.sam"])
    _write_bgzf(fq, d / "b.fq.gz")
    _write_bgzf(fq, d / "odd.fq.gz", block=7919)
    with open(fq, "rb") as f, gzip.open(d / "single.fq.gz", "wb", compresslevel=1) as o:
        o.write(f.read())
    for name, env in (("b", SMALL), ("odd", dict(SMALL, ABM_CLI_CHUNK_BYTES="30000")), ("b", dict(SMALL, ABM_CLI_INFLATE_AHEAD="300000")),
                      ("b", dict(SMALL, ABM_CLI_NO_BGZF="1")), ("single", SMALL)):
        run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", d / "z.sam", d / f"{name}.fq.gz"], env=env)
        assert body([d / "z.sam"]) == ref, (name, env)
    # a damaged block is an error, not silence
    raw = bytearray(open(d / "b.fq.gz", "rb").read())
    raw[len(raw) // 2] ^= 0x55
    open(d / "bad.fq.gz", "wb").write(raw)
    r = subprocess.run([CLI, "map", "-virtual-gpus", "1", "-i", trex_index, "-o", str(d / "bad.sam"), str(d / "bad.fq.gz")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and ("BGZF" in r.stderr or "reads file" in r.stderr), r.stderr


def test_bgzf_blocks_of_the_fast_deflate_inflate_to_their_input(tmp_path):
    # -B output is compressed by the CLI's own deflate at -z 1 (one fixed-Huffman block per BGZF block, greedy matches from a
    # single-probe hash), stored at -z 0, by zlib above: `abismal-amd bgzf` runs the same code on any file.  Runs, text,
    # BAM-like records, incompressible bytes (a block the fast encoder cannot fit is stored), tiny and empty inputs.
    import gzip
    import random
    rnd = random.Random(5)
    cases = {"empty": b"", "one": b"x", "three": b"abc", "zeros": bytes(200000), "ff": b"\xff" * 70000,
             "text": b"".join(b"@read%d\nACGTTGCA%s\n+\nIIIIIIII\n" % (k, b"ACGT"[k % 4:k % 4 + 1] * (k % 50)) for k in range(20000)),
             "random": bytes(rnd.getrandbits(8) for _ in range(150000)), "high": bytes(144 + rnd.getrandbits(6) for _ in range(140000)),
             "period7": (b"abcdefg" * 30000)[:199999], "far": (bytes(rnd.getrandbits(8) for _ in range(40000)) * 4)}
    for name, data in cases.items():
        src, dst = tmp_path / f"{name}.bin", tmp_path / f"{name}.bgzf"
        open(src, "wb").write(data)
        for z in (1, 0, 6):
            r = subprocess.run([CLI, "bgzf", "-z", str(z), str(src), str(dst)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            assert r.returncode == 0, r.stderr
            assert gzip.open(dst).read() == data, (name, z)
            raw = open(dst, "rb").read()
            at = 0
            while at < len(raw):  # every block a valid BGZF block of at most 64 KB
                assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 14] == b"BC"
                bsize = raw[at + 16] | (raw[at + 17] << 8)
                at += bsize + 1
            assert at == len(raw)
        if name in ("zeros", "ff", "text", "period7"):
            assert len(open(dst, "rb").read()) < len(data) // 2, name  # (-z 6 ran last)


def test_bam_output_is_the_same_records_at_every_level(reads, trex_index):
    import gzip
    fq, d = reads
    got = {}
    for z in (1, 0, 6):
        run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-B", "-z", z, "-i", trex_index, "-o", d / f"z{z}.bam", fq], env=SMALL)
        x = gzip.open(d / f"z{z}.bam").read()
        l_text = int.from_bytes(x[4:8], "little")
        got[z] = x[8 + l_text:]  # (the header text holds the command line)
        assert x[:4] == b"BAM\x01"
    assert got[1] == got[0] == got[6] and len(got[1]) > 60000 * 150
    assert os.path.getsize(d / "z1.bam") < os.path.getsize(d / "z0.bam") // 3
    # part files of BAM output: `cat` of the parts is one valid BAM (header in the first, end-of-file block in the last)
    run(["-virtual-gpus", 2, "-out-parts", 2, "-t", 4, "-batch", 4096, "-B", "-i", trex_index, "-o", d / "p.bam", fq], env=SMALL)
    whole = open(f"{d}/p.bam.part000", "rb").read() + open(f"{d}/p.bam.part001", "rb").read()
    assert whole[-28:-16] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0]) and open(f"{d}/p.bam.part000", "rb").read()[-28:] != whole[-28:]
    x = gzip.decompress(whole)
    assert x[8 + int.from_bytes(x[4:8], "little"):] == got[1]


def test_lead_in_is_bounded_for_a_library_of_short_reads(reads, trex_index):
    # a library of uniformly short reads (45 bases: every read could be a ghost-bit source for the next
    # batch) made each batch carry every read seen so far.  The lead-in holds only records that are longer than
    # everything after them: for one read length, one record.
    fq, d = reads
    lines = open(fq).read().split("\n")
    for k in range(0, len(lines) - 3, 4):
        lines[k + 1], lines[k + 3] = lines[k + 1][:45], lines[k + 3][:45]
    open(d / "short.fq", "w").write("\n".join(lines))
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 2048, "-timing", d / "t.json", "-i", trex_index, "-o", d / "short.sam", d / "short.fq"], env=SMALL)
    t = json.load(open(d / "t.json"))
    assert t["batches"] >= 20 and t["max_lead_in_records"] <= 2, t
    # mixed lengths 44..120: at most one record per length step up to 110 bases
    for k in range(0, len(lines) - 3, 4):
        cut = 44 + (k // 4 * 7919) % 77
        lines[k + 1], lines[k + 3] = lines[k + 1][:cut], lines[k + 3][:cut]
    open(d / "mixed.fq", "w").write("\n".join(lines))
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 2048, "-timing", d / "t.json", "-i", trex_index, "-o", d / "mixed.sam", d / "mixed.fq"], env=SMALL)
    t = json.load(open(d / "t.json"))
    assert t["batches"] >= 20 and 1 <= t["max_lead_in_records"] <= 67, t


def test_region_lead_in_is_scanned_back_until_it_closes(reads, trex_index):
    # with -out-parts a later region's lead-in came from the 4096 records before it only.  It is now scanned
    # backwards, window by window, until the records found cover all 110 positions a 44-46-base read can look at, the
    # input begins, or the budget is spent.  One 140-base record at record 100 of a 100-base library (the simulator's
    # reads with that one record's sequence doubled): the second of two regions starts at record 30000, its lead-in is
    # that record plus the last one before the region, found after scanning some 29600 records back in windows of 64 (regions start at whole slices).
    fq, d = reads
    lines = open(fq).read().split("\n")
    lines[4 * 100 + 1] = (lines[4 * 100 + 1] * 2)[:140]
    lines[4 * 100 + 3] = (lines[4 * 100 + 3] * 2)[:140]
    open(d / "one_long.fq", "w").write("\n".join(lines))
    env = dict(SMALL, ABM_CLI_LEAD_RECORDS="64")
    run(["-virtual-gpus", 2, "-out-parts", 2, "-t", 4, "-batch", 2048, "-timing", d / "t.json", "-i", trex_index, "-o", d / "ol.sam", d / "one_long.fq"], env=env)
    t = json.load(open(d / "t.json"))
    assert t["region_lead_in_records"] == 2 and 29000 <= t["region_lead_in_scanned_records"] <= 30000, t
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 2048, "-i", trex_index, "-o", d / "ol1.sam", d / "one_long.fq"], env=env)
    assert body([f"{d}/ol.sam.part000", f"{d}/ol.sam.part001"]) == body([d / "ol1.sam"])
    # the budget bounds the scan for a library that never closes it (one read length below 110 bases: the last record is the lead-in)
    run(["-virtual-gpus", 2, "-out-parts", 2, "-t", 4, "-batch", 2048, "-timing", d / "t.json", "-i", trex_index, "-o", d / "ol.sam", fq],
        env=dict(env, ABM_CLI_LEAD_BUDGET="1000"))
    t = json.load(open(d / "t.json"))
    assert t["region_lead_in_records"] == 1 and 1000 <= t["region_lead_in_scanned_records"] < 1064, t


def test_tiny_and_ragged_inputs(reads, trex_index):
    # an empty file, one record, a last line without its newline, a record cut off after its sequence line -- with one
    # output file and with more parts than there are slices (the later regions are empty), plain and BGZF
    fq, d = reads
    head = open(fq).read().split("\n")
    cases = {"empty": "", "one": "\n".join(head[:4]) + "\n", "nonl": "\n".join(head[:8]), "partial": "\n".join(head[:6]) + "\n"}
    for name, text in cases.items():
        open(d / f"{name}.fq", "w").write(text)
        _write_bgzf(d / f"{name}.fq", d / f"{name}.fq.gz")
        want = {"empty": 0, "one": 1, "nonl": 2, "partial": 2}[name]
        for src, parts in ((f"{name}.fq", 1), (f"{name}.fq", 2), (f"{name}.fq.gz", 1)):
            run(["-virtual-gpus", 2, "-out-parts", parts, "-t", 3, "-i", trex_index, "-o", d / "e.sam", "-s", d / "e.st", d / src])
            files = [d / "e.sam"] if parts == 1 else [f"{d}/e.sam.part000", f"{d}/e.sam.part001"]
            assert f"total_reads: {want}\n" in open(d / "e.st").read(), (name, src, parts)
            assert len([ln for ln in body(files) if not ln.startswith("@")]) <= want
            for f in files:
                os.remove(f)


def test_host_report():
    # `abismal-amd host`: the NUMA nodes, cores and CPU quota the pipeline places its threads by, and its default worker counts
    r = subprocess.run([CLI, "host"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    rows = dict(ln.split(": ", 1) for ln in r.stdout.splitlines() if ": " in ln and not ln.startswith(" "))
    assert int(rows["numa_nodes"]) >= 1
    workers = [int(rows[f"default host workers with {g} GPU(s)"]) for g in (1, 2, 4, 8)]
    assert workers == sorted(workers) and 1 <= workers[0] <= (os.cpu_count() or 1)
    quota = float(rows["cpu_quota_cpus"].split()[0])
    if quota > 0:
        assert workers[-1] <= max(1, round(quota))


def test_more_parts_than_mappers_is_refused(reads, trex_index):
    fq, d = reads
    r = subprocess.run([CLI, "map", "-virtual-gpus", "1", "-out-parts", "3", "-i", trex_index, "-o", str(d / "x.sam"), fq],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and "out-parts" in r.stderr


def test_vectorised_seq_equals_the_tables(trex_index, tmp_path):
    """SEQ of a SAM line is written 32 bases at a time where the CPU has AVX2 (forward: a copy where nothing but upper-case
    A, C, G, T, N is met; reverse-complemented hits: compares and blends) and through the per-base tables otherwise
    (ABM_CLI_SCALAR_SEQ=1 forces them): the same bytes for reads of every length around the vector width, with
    lower-case letters, IUPAC codes, N runs and other bytes in them, on both strands (virtual GPUs hand half of the reads
    a reverse-strand hit)."""
    import random
    rng = random.Random(5)
    alphabet = "ACGT" * 12 + "N" * 3 + "acgtn" + "RYKMSWBDHV" + "ryk" + "=.*-"
    fq = tmp_path / "odd.fq"
    with open(fq, "w") as f:
        for k in range(6000):
            L = rng.choice([44, 45, 63, 64, 65, 95, 96, 97, 100, 127, 128, 129, 150, 151, 200, 257])
            core = "".join(rng.choice("ACGT") for _ in range(L))
            if k % 3:
                s = list(core)
                for _ in range(rng.randrange(1, 12)):
                    s[rng.randrange(1, L - 1)] = rng.choice(alphabet)
                core = "".join(s)
            f.write(f"@r{k}\n{core}\n+\n{'I' * L}\n")
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", tmp_path / "v.sam", "-s", tmp_path / "v.st", fq], env=SMALL)
    run(["-virtual-gpus", 2, "-t", 4, "-batch", 4096, "-i", trex_index, "-o", tmp_path / "t.sam", "-s", tmp_path / "t.st", fq],
        env=dict(SMALL, ABM_CLI_SCALAR_SEQ="1"))
    got, want = body([tmp_path / "v.sam"]), body([tmp_path / "t.sam"])
    assert len(want) > 5000 and any(ln.split("\t")[1] == "16" for ln in want) and any(ln.split("\t")[1] == "0" for ln in want)
    assert got == want
