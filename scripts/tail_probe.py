"""Diagnostic: per-read GPU time distribution and the oracle's work counts for the slowest reads."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, abismal_amd as A
from tests import oracle_binding as ob
mbp = float(os.environ.get("ABM_BENCH_GENOME_MBP", 400)); n = int(os.environ.get("ABM_BENCH_READS", 1000000)); L = int(os.environ.get("ABM_TAIL_L", 100))
MODE = int(os.environ.get("ABM_TAIL_MODE", 0))  # 0 T-rich, 2 random PBAT
dev = torch.device("cuda", 0)
wd = "/tmp/abismal_bench"; os.makedirs(wd, exist_ok=True)
idx = f"{wd}/g{int(mbp)}.idx"
if not os.path.exists(idx):
    bench.synth_genome_fasta(idx + ".fa", mbp, 1234, dev); A.index_build(idx + ".fa", idx, os.cpu_count())
index = A.Index(idx, window_records=L); ctx = A.Context(index, 0)
names, starts, gw = bench.read_index_genome(idx)
blob, _ = bench.sample_reads(gw, starts, n, L, int(os.environ.get("ABM_TAIL_SEED", 1000)), dev, pbat_frac=0.5 if MODE == 2 else 0.0)
off = torch.arange(0, (n + 1) * L, L, dtype=torch.int64, device=dev)
res = torch.zeros((n, 2), dtype=torch.int32, device=dev); cig = torch.zeros((n, 8), dtype=torch.int32, device=dev)
cn = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
rc = torch.zeros(n, dtype=torch.int32, device=dev)
ctx.set_phase_stamps(True); ctx.set_read_cycles(rc.data_ptr())
p = A.Params()
for _ in range(2):
    ctx.map_se_device(MODE, p, n, blob.data_ptr(), off.data_ptr(), L, res.data_ptr(), cig.data_ptr(), 8, cn.data_ptr(), st.data_ptr(), 0)
    torch.cuda.synchronize()
c = rc.cpu().numpy().astype(np.float64) * 1024 / 2.1e3  # microseconds at 2.1 GHz
print("per-read us: mean %.1f p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f ; sum %.2f s; top-100 sum %.3f s" % (
    c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), np.percentile(c, 99.9), c.max(), c.sum() / 1e6, np.sort(c)[-100:].sum() / 1e6))
edges = [0, 10, 30, 100, 300, 1000, 3000, 10000, 30000, 100000, 1e9]
for lo_, hi_ in zip(edges[:-1], edges[1:]):
    sel = (c >= lo_) & (c < hi_)
    print("  reads with %7.0f <= us < %7.0f : %9d reads, %8.3f wave-seconds (%.1f %%)" % (lo_, hi_, int(sel.sum()), c[sel].sum() / 1e6, 100 * c[sel].sum() / c.sum()))
worst = np.argsort(c)[-8:][::-1]
o = ob.load(); oix = o.index_load(idx)
host = blob.cpu().numpy().reshape(n, L)
for w in worst:
    r = bytes(host[w]); _, _, _, wk = o.map_se(oix, [r], mode=MODE)
    print("read %d gpu %.0f us: %s  %s" % (w, c[w], r.decode(), {k: wk[k] for k in ("candidates", "set_updates", "search_probes", "aligns")}))
print(ctx.take_work().get("phase_cycles"))
