"""Diagnostic: the costliest kinds of read, each mapped ALONE on the device (launch time = that read's time),
with the phase split and the counts of filter steps / ordered set updates (how many through the O(1) run path)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, abismal_amd as A
mbp = float(os.environ.get("ABM_BENCH_GENOME_MBP", 3100)); L = 100
dev = torch.device("cuda", 0)
wd = "/tmp/abismal_bench"; os.makedirs(wd, exist_ok=True)
idx = f"{wd}/g{int(mbp)}.idx"
if not os.path.exists(idx):
    bench.synth_genome_fasta(idx + ".fa", mbp, 1234, dev); A.index_build(idx + ".fa", idx, os.cpu_count())
index = A.Index(idx); ctx = A.Context(index, 0)
reads = [b"T" * 100, b"A" * 100, b"T" * 29 + b"A" + b"T" * 7 + b"A" + b"T" * 62,
         b"GTATTAGAAGTGG" + b"T" * 87, b"TTGAGGTGTTTTA" + b"T" * 87, b"ATGGATGAGTTG" + b"T" * 88,
         b"CACACACACA" * 10, b"TTAGGG" * 16 + b"TTAG"]
p = A.Params()
for timed in (False, True):
    ctx.set_phase_stamps(timed)
    for r in reads:
        n = 1
        blob = torch.frombuffer(bytearray(r), dtype=torch.uint8).to(dev)
        off = torch.tensor([0, L], dtype=torch.int64, device=dev)
        res = torch.zeros((n, 2), dtype=torch.int32, device=dev); cig = torch.zeros((n, 8), dtype=torch.int32, device=dev)
        cn = torch.zeros(n, dtype=torch.int32, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.map_se_device(0, p, n, blob.data_ptr(), off.data_ptr(), L, res.data_ptr(), cig.data_ptr(), 8, cn.data_ptr(), st.data_ptr(), 0)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            w = ctx.take_work()
        line = "%s %-40s %8.2f ms  cands %8d updates %7d" % ("timed" if timed else "plain", r[:40].decode(), best * 1e3, w["candidates"], w["set_updates"])
        if timed:
            pc = w["phase_cycles"]; tot = max(1, pc["total"])
            line += "  fifo %7d steps %6d | probe %.0f%% filter %.0f%% replay %.0f%% align %.0f%%" % (
                w.get("fifo_updates", -1), w.get("filter_steps", -1), 100 * pc["probe_narrow"] / tot, 100 * pc["gather_hamming"] / tot, 100 * pc["replay"] / tot, 100 * pc["align"] / tot)
        print(line, flush=True)
