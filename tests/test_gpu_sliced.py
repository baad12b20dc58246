"""GPU: abm_map_se_batch_sliced / abm_ctx_slice_results -- a batch's results handed over slice by slice while the kernel
runs -- give exactly what abm_map_se_batch gives for the same reads (hits and CIGARs), whatever the slices look
like: uneven, empty, one, thousands, a lead-in that belongs to no slice, reads of the long-read launch in the batch
(then every slice arrives at the end), and a CIGAR arena that overflows (the batch is mapped again, no slice twice)."""
import os

import numpy as np
import pytest

from tests import oracle_binding as ob
from tests.test_gpu_edges_and_properties import _reads_from_genome, cigars

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(trex_index):
    import abismal_amd as A
    ix = A.Index(trex_index)
    c = A.Context(ix, 0)
    yield c
    c.close()
    ix.close()


@pytest.fixture(scope="module")
def reads(oracle, workdir):
    prefix = os.path.join(workdir, "sliced_reads")
    oracle.simulate(os.path.join(GOLD, "tRex1.fa"), prefix, 30000, single_end=True, seed=17)
    return ob.read_fastq_like_readloader(prefix + "_1.fq")[1]


def same_as_whole_batch(ctx, reads, first, mode=0):
    res, cig, off = ctx.map_se(reads, mode=mode)
    s_res, s_cig, s_off, arrived = ctx.map_se_sliced(reads, first, mode=mode)
    lo = int(first[0])
    assert sorted(arrived) == list(range(len(first) - 1))
    for f in ("pos", "diffs", "flags"):
        assert (res[f][lo:] == s_res[f][lo:]).all(), f
    assert cigars(cig, off)[lo:] == cigars(s_cig, s_off)[lo:]
    return arrived


@pytest.mark.parametrize("layout", ["even", "uneven", "one", "tiny", "empty_slices", "lead_in"])
def test_slices_equal_the_whole_batch(ctx, reads, layout):
    n = len(reads)
    rng = np.random.default_rng(5)
    if layout == "even":
        first = list(range(0, n, 4096)) + [n]
    elif layout == "uneven":
        cuts = sorted(set(int(x) for x in rng.integers(1, n, 23)))
        first = [0] + cuts + [n]
    elif layout == "one":
        first = [0, n]
    elif layout == "tiny":  # thousands of slices, a few reads each (blocks of the ordering kernels span many slices)
        first = list(range(0, n, 7)) + [n]
    elif layout == "empty_slices":
        first = [0, 0, 5000, 5000, 5000, 12000, n, n]
    else:  # the first 11 reads belong to no slice
        first = [11, 3000, 20000, n]
    arrived = same_as_whole_batch(ctx, reads, first, mode=2 if layout == "uneven" else 0)
    assert len(arrived) == len(first) - 1


def test_slices_with_reads_of_the_long_read_launch(ctx, trex_index):
    lengths = [100] * 300 + [1500, 2500] + [120] * 200 + [5000]
    rs = _reads_from_genome(trex_index, lengths, seed=4, indel_every=3000)
    same_as_whole_batch(ctx, rs, [0, 100, 301, 400, len(rs)])


def test_arena_overflow_maps_again_and_hands_no_slice_twice(trex_index):
    # 1000-base reads with an indel every ~100 bases: ~20 CIGAR ops each, beyond the 4-op slots -- 9000 of them need
    # more arena than the default 65536 ops.  A fresh context, and the sliced call first: the arena it starts with
    # is the default one (a context keeps the larger arena once a batch has needed it).
    import abismal_amd as A
    rs = _reads_from_genome(trex_index, [1000] * 9000, seed=8, indel_every=100)
    first = [0, 2300, 4600, 6900, 9000]
    ix = A.Index(trex_index)
    c = A.Context(ix, 0)
    try:
        s_res, s_cig, s_off, arrived = c.map_se_sliced(rs, first)
        assert sorted(arrived) == [0, 1, 2, 3]
        assert int(s_off[-1]) > 70000, "the batch must overflow the default arena"
        res, cig, off = c.map_se(rs)
        for f in ("pos", "diffs", "flags"):
            assert (res[f] == s_res[f]).all(), f
        assert cigars(cig, off) == cigars(s_cig, s_off)
    finally:
        c.close()
        ix.close()


_MANY_PER_WAVE = r"""
import sys
import numpy as np
import abismal_amd as A
from tests.test_gpu_edges_and_properties import _reads_from_genome, cigars
idx = sys.argv[1]
rs = _reads_from_genome(idx, [1000] * 9000, seed=8, indel_every=100)
first = list(range(0, 9000, 500)) + [9000]
ix = A.Index(idx)
c = A.Context(ix, 0)
s_res, s_cig, s_off, arrived = c.map_se_sliced(rs, first)
assert sorted(arrived) == list(range(len(first) - 1)), arrived
assert int(s_off[-1]) > 70000, "the batch must overflow the default arena"
res, cig, off = c.map_se(rs)
for f in ("pos", "diffs", "flags"):
    assert (res[f] == s_res[f]).all(), f
a, b = cigars(cig, off), cigars(s_cig, s_off)
bad = [i for i in range(len(a)) if a[i] != b[i]]
assert not bad, f"{len(bad)} reads handed over with another CIGAR than the whole-batch call's, first {bad[:5]}"
c.close(); ix.close()
print("ok")
"""


def test_arena_overflow_with_many_overflowing_reads_per_wave(trex_index):
    # `overflow` was sticky per wave, so only a wave's FIRST read without room in the arena kept its
    # slice open; every later one was handed over truncated.  64 waves for 9000 reads = 140 overflowing reads per wave
    # (the grid switch is an experiment variable read once per process: a child process).
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ABM_EXPERIMENTS="1", ABM_GRID_WAVES="64", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _MANY_PER_WAVE, trex_index], env=env, cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
