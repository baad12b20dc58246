"""Parity at scale for the modes bench.py does not exercise: GPU vs oracle on a large synthetic genome,
single-end A-rich / random-PBAT and paired-end PBAT / random-PBAT, hits and CIGARs."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, abismal_amd as A
from tests import oracle_binding as ob
from tests.test_gpu_se_parity import compare_se
from tests.test_gpu_pe_parity import compare_pe
mbp = float(os.environ.get("ABM_BENCH_GENOME_MBP", 1000)); n = int(os.environ.get("ABM_BENCH_READS", 200000))
dev = torch.device("cuda", 0)
wd = "/tmp/abismal_bench"; os.makedirs(wd, exist_ok=True)
idx = f"{wd}/g{int(mbp)}.idx"
if not os.path.exists(idx):
    bench.synth_genome_fasta(idx + ".fa", mbp, 1234, dev); A.index_build(idx + ".fa", idx, os.cpu_count())
# (window records for reads of up to 172 bases unless SCALE_PARITY_WINDOW_RECORDS says otherwise: the record-fed kernels)
index = A.Index(idx, window_records=int(os.environ.get("SCALE_PARITY_WINDOW_RECORDS", 172))); ctx = A.Context(index, 0)
print("window records serve reads up to", ctx.window_records(), "bases", flush=True)
_, starts, gw = bench.read_index_genome(idx)
o = ob.load(); oix = o.index_load(idx)
threads = os.cpu_count() or 8
comp = bytes.maketrans(b"ACGT", b"TGCA")
def host_reads(blob, L): return [bytes(r) for r in blob.cpu().numpy().reshape(-1, L)]
# single-end: reads sampled T-rich; A-rich input = reverse complements; random = a mix
L = 100
se = host_reads(bench.sample_reads(gw, starts, n, L, 4242, dev)[0], L)
arich = [r.translate(comp)[::-1] for r in se]
mixed = [a if i & 1 else t for i, (t, a) in enumerate(zip(se, arich))]
for mode, reads, name in ((1, arich, "SE A-rich"), (2, mixed, "SE random-PBAT")):
    t = time.time(); res, cig, off = ctx.map_se(reads, mode=mode); tg = time.time() - t
    t = time.time(); o_res, o_cig, o_n, _ = o.map_se(oix, reads, mode=mode, threads=threads); to = time.time() - t
    compare_se(res, cig, off, o_res, o_cig, o_n, reads, name)
    print(f"{name}: {len(reads)} reads identical (hits + CIGARs); mapped {(res['pos'] != 0).mean():.3f}; GPU {tg:.2f}s oracle {to:.2f}s", flush=True)
Lp, npairs = 150, n // 2
b1, b2 = bench.sample_pairs(gw, starts, npairs, Lp, 777, dev)
r1, r2 = host_reads(b1, Lp), host_reads(b2, Lp)
p1 = [r.translate(comp)[::-1] for r in r1]; p2 = [r.translate(comp)[::-1] for r in r2]
m1 = [a if i & 1 else t for i, (t, a) in enumerate(zip(r1, p1))]; m2 = [a if i & 1 else t for i, (t, a) in enumerate(zip(r2, p2))]
for mode, x1, x2, name in ((1, p1, p2, "PE PBAT"), (2, m1, m2, "PE random-PBAT")):
    t = time.time(); gpu = ctx.map_pe(x1, x2, mode=mode); tg = time.time() - t
    t = time.time(); orc = o.map_pe(oix, x1, x2, mode=mode, threads=threads); to = time.time() - t
    compare_pe(gpu, orc, name)
    print(f"{name}: {len(x1)} pairs identical; concordant {(gpu[0]['r1']['pos'] != 0).mean():.3f}; GPU {tg:.2f}s oracle {to:.2f}s", flush=True)
