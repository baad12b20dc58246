"""CPU: the oracle's multi-threaded modes must equal its single-threaded run (= the reference at -t 1)
even for reads of 44-46 bases, whose result depends on what earlier reads left in the reused buffers
(SURVEY A.11): every share replays the reads before it for that side effect (Mapper::touch_*)."""
import os
import subprocess

import numpy as np

from tests import oracle_binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ghosty_reads(oracle, workdir, n=6000, seed=3):
    prefix = os.path.join(workdir, "ghosty")
    oracle.simulate(os.path.join(ROOT, "tests", "golden", "tRex1.fa"), prefix, n, single_end=True, seed=seed)
    _, reads = ob.read_fastq_like_readloader(prefix + "_1.fq")
    rng = np.random.default_rng(seed)
    out = []
    for r in reads:
        u = rng.random()
        if u < 0.5 and len(r) >= 47:
            r = r[: int(rng.integers(44, 47))]
        elif u < 0.6 and len(r) >= 60:
            r = r[: int(rng.integers(47, len(r)))]
        out.append(r)
    return out


def test_sharded_library_equals_single_thread(oracle, trex_index, workdir):
    reads = ghosty_reads(oracle, workdir)
    oix = oracle.index_load(trex_index)
    try:
        for mode in (0, 2):
            a = oracle.map_se(oix, reads, mode=mode, max_candidates=3, threads=1)
            b = oracle.map_se(oix, reads, mode=mode, max_candidates=3, threads=7)
            assert a[0].tobytes() == b[0].tobytes() and a[2].tobytes() == b[2].tobytes() and (a[1] == b[1]).all()
        r1, r2 = reads[:2500], reads[2500:5000]
        a = oracle.map_pe(oix, r1, r2, mode=0, max_candidates=3, threads=1)
        b = oracle.map_pe(oix, r1, r2, mode=0, max_candidates=3, threads=5)
        for x, y in zip(a[:3], b[:3]):
            assert x.tobytes() == y.tobytes()
    finally:
        oracle.index_free(oix)


def test_cli_threads_equal_single_thread(oracle, trex_index, workdir):
    reads = ghosty_reads(oracle, workdir, seed=8)
    fq = os.path.join(workdir, "ghosty_cli.fq")
    with open(fq, "w") as f:
        for i, r in enumerate(reads):
            f.write(f"@g{i}\n{r or 'N' * 50}\n+\n{'I' * len(r or 'N' * 50)}\n")
    outs = []
    for t in ("1", "6"):
        sam = os.path.join(workdir, f"ghosty_t{t}.sam")
        subprocess.run([ob.CLI, "map", "-t", t, "-c", "3", "-i", trex_index, "-o", sam, "-s", sam + ".stats", fq], check=True)
        outs.append(([ln for ln in open(sam) if not ln.startswith("@PG")], open(sam + ".stats").read()))
    assert outs[0] == outs[1] and len(outs[0][0]) > 3000
