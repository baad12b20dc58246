"""The read sharding over GPUs and the path's one collective.  Every test runs twice: as TWO REPLICAS ON ONE DEVICE
(`-devices 0,0`: what a one-GPU box can execute -- batches dealt to both replicas' mapper threads, the lead-in carried
across the boundary, the ordered merge, per-GPU statistics summed) and on two physical GPUs (skipped where fewer are
visible).
 * abm_stats_allreduce over two contexts (RCCL, in-process communicators) == the host sum;
 * `abismal-amd map -gpus 2` writes the same SAM body and statistics as `-gpus 1`;
 * `python bench.py --gpus 2` starts two ranks itself and reports ranks_seen == 2."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "abismal_amd", "abismal-amd")


def _n_gpus():
    import torch
    return torch.cuda.device_count()


needs_two = pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")


def _variants():
    # what a one-GPU box can execute always runs; the two-physical-GPU variant joins it where two are visible
    return ["replicas_on_one_device"] + (["two_gpus"] if _n_gpus() >= 2 else [])


def test_stats_allreduce_two_contexts(trex_index):
    for devices in [(0, 0)] + ([(0, 1)] if _n_gpus() >= 2 else []):
        _stats_allreduce(trex_index, devices)


def _stats_allreduce(trex_index, devices):
    # (0, 0): two contexts that share a device are summed on the host inside the call (a communicator takes a device once)
    import abismal_amd as A
    lib = A.load_library()
    lib.abm_stats_allreduce.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(C.c_uint64))]
    ix = A.Index(trex_index)
    ctxs = [A.Context(ix, d) for d in devices]
    try:
        rng = np.random.default_rng(7)
        mine = [rng.integers(0, 1 << 40, 18, dtype=np.uint64) for _ in range(2)]
        want = mine[0] + mine[1]
        bufs = [m.copy() for m in mine]
        handles = (C.c_void_p * 2)(*[c.handle for c in ctxs])
        ptrs = (C.POINTER(C.c_uint64) * 2)(*[b.ctypes.data_as(C.POINTER(C.c_uint64)) for b in bufs])
        rc = lib.abm_stats_allreduce(handles, 2, ptrs)
        assert rc == 0, lib.abm_last_error().decode()
        assert (bufs[0] == want).all() and (bufs[1] == want).all()
    finally:
        for c in ctxs:
            c.close()
        ix.close()


def _short_every_fifth(path):
    # every fifth read cut to 44-46 bases: what such a read finds past its end comes from the reads before it, which
    # for the first reads of a batch were mapped by the OTHER GPU (the lead-in a batch carries along)
    lines = open(path).read().split("\n")
    for k in range(0, len(lines) - 3, 20):
        cut = 44 + (k // 20) % 3
        lines[k + 1], lines[k + 3] = lines[k + 1][:cut], lines[k + 3][:cut]
    open(path, "w").write("\n".join(lines))


def test_cli_two_gpus_same_output_as_one(oracle, trex_index, tmp_path):
    for how in _variants():
        d = tmp_path / how
        d.mkdir()
        _cli_two_gpus_same_output_as_one(oracle, trex_index, d, how)


def _cli_two_gpus_same_output_as_one(oracle, trex_index, tmp_path, how):
    import re
    fa = os.path.join(ROOT, "tests", "golden", "tRex1.fa")
    oracle.simulate(fa, str(tmp_path / "r"), 40000, single_end=True, seed=11)
    _short_every_fifth(tmp_path / "r_1.fq")
    body = {}
    # slices and batches of 1024 reads (the units are 32 k reads by default: one or two slices would be one batch on one GPU)
    env = dict(os.environ, ABM_CLI_SLICE_READS="1024", ABM_CLI_FIRST_BATCH="1024", ABM_CLI_CHUNK_BYTES="65536",
               ABM_CLI_MARK_LINES="64")
    two = ["-gpus", "2"] if how == "two_gpus" else ["-devices", "0,0"]
    for g, sel in ((1, ["-gpus", "1"]), (2, two), (3, two + ["-out-parts", "2"])):
        out, st = tmp_path / f"g{g}.sam", tmp_path / f"g{g}.mstats"
        r = subprocess.run([CLI, "map", "-v"] + sel + ["-batch", "1024", "-s", str(st), "-o", str(out), "-i", trex_index,
                            str(tmp_path / "r_1.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 0, r.stderr
        per_gpu = [int(m.group(1)) for m in re.finditer(r"GPU \d+: (\d+) batches", r.stderr)]
        assert len(per_gpu) == min(g, 2) and sum(per_gpu) >= 30, r.stderr
        if g >= 2:
            assert ("one RCCL all-reduce" if how == "two_gpus" else "replicas share a device") in r.stderr, r.stderr
            assert min(per_gpu) >= 5, f"both GPUs must have mapped batches: {per_gpu}"
        files = [out] if g < 3 else [f"{out}.part000", f"{out}.part001"]
        lines = []
        for f in files:
            lines += [ln for ln in open(f) if not ln.startswith("@PG")]
        body[g] = (lines, open(st).read())
    assert body[1] == body[2] and len(body[1][0]) > 30000
    # two parts: each GPU maps its own contiguous half of the input into its own file; `cat` of the parts is the file
    assert body[1] == body[3]


def test_cli_two_gpus_paired_end(oracle, trex_index, tmp_path):
    for how in _variants():
        d = tmp_path / how
        d.mkdir()
        _cli_two_gpus_paired_end(oracle, trex_index, d, how)


def _cli_two_gpus_paired_end(oracle, trex_index, tmp_path, how):
    fa = os.path.join(ROOT, "tests", "golden", "tRex1.fa")
    oracle.simulate(fa, str(tmp_path / "p"), 12000, seed=12)
    env = dict(os.environ, ABM_CLI_SLICE_READS="512", ABM_CLI_FIRST_BATCH="512", ABM_CLI_CHUNK_BYTES="65536", ABM_CLI_MARK_LINES="64")
    two = ["-gpus", "2"] if how == "two_gpus" else ["-devices", "0,0"]
    body = {}
    for g, sel in ((1, ["-gpus", "1"]), (2, two + ["-out-parts", "2"])):
        out, st = tmp_path / f"g{g}.sam", tmp_path / f"g{g}.mstats"
        r = subprocess.run([CLI, "map"] + sel + ["-batch", "512", "-s", str(st), "-o", str(out), "-i", trex_index,
                            str(tmp_path / "p_1.fq"), str(tmp_path / "p_2.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        assert r.returncode == 0, r.stderr
        files = [out] if g == 1 else [f"{out}.part000", f"{out}.part001"]
        lines = []
        for f in files:
            lines += [ln for ln in open(f) if not ln.startswith("@PG")]
        body[g] = (lines, open(st).read())
    assert body[1] == body[2] and len(body[1][0]) > 15000


@needs_two
def test_bench_self_launch_two_gpus(tmp_path):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mbp", "40",
                        "--reads", "200000", "--no-cpu-baseline", "--workdir", str(tmp_path)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_reads_per_s"]) == 2
    assert d["mapping"]["total"] == 2 * 200000
    assert d["value"] > max(d["per_rank_reads_per_s"])
