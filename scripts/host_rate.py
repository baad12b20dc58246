"""PCIe-inclusive rate: reads handed over in HOST buffers through abm_map_se_batch
(H2D of reads, kernels, D2H of results, host-side CIGAR compaction).  Never the bench `value`."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, abismal_amd as A
mbp = float(os.environ.get("ABM_BENCH_GENOME_MBP", 400)); n = int(os.environ.get("ABM_BENCH_READS", 2000000)); L = 100
dev = torch.device("cuda", 0)
wd = "/tmp/abismal_bench"; os.makedirs(wd, exist_ok=True)
idx = f"{wd}/g{int(mbp)}.idx"
if not os.path.exists(idx):
    bench.synth_genome_fasta(idx + ".fa", mbp, 1234, dev); A.index_build(idx + ".fa", idx, os.cpu_count())
index = A.Index(idx); ctx = A.Context(index, 0)
_, starts, gw = bench.read_index_genome(idx)
blob, _ = bench.sample_reads(gw, starts, n, L, 1000, dev)
host = blob.cpu().numpy()
off = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
res = np.zeros(n, dtype=A.HIT_DTYPE); cig = np.zeros(n * 16, dtype=np.uint32); co = np.zeros(n + 1, dtype=np.uint64)
import ctypes as C
p = A.Params()
lib = A.load_library()
for it in range(3):
    t = time.perf_counter()
    rc = lib.abm_map_se_batch(ctx.handle, 0, C.byref(p), n, host.ctypes.data, off.ctypes.data, res.ctypes.data, cig.ctypes.data, len(cig), co.ctypes.data)
    dt = time.perf_counter() - t
    assert rc == 0, lib.abm_last_error()
    print(f"abm_map_se_batch (host buffers): {n} reads in {dt*1e3:.1f} ms = {n/dt/1e6:.2f} M reads/s; mapped {(res['pos']!=0).sum()}")
