"""End-to-end rate of the abismal-amd CLI (FASTQ in, SAM out, host I/O included)."""
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, abismal_amd as A
mbp = float(os.environ.get("ABM_BENCH_GENOME_MBP", 400)); n = int(os.environ.get("ABM_BENCH_READS", 4000000)); L = 100
dev = torch.device("cuda", 0)
wd = "/tmp/abismal_bench"; os.makedirs(wd, exist_ok=True)
idx = f"{wd}/g{int(mbp)}.idx"
if not os.path.exists(idx):
    bench.synth_genome_fasta(idx + ".fa", mbp, 1234, dev); A.index_build(idx + ".fa", idx, os.cpu_count())
_, starts, gw = bench.read_index_genome(idx)
blob, _ = bench.sample_reads(gw, starts, n, L, 1000, dev)
host = blob.cpu().numpy().reshape(n, L)
fq = f"{wd}/reads_{n}.fq"
t = time.time()
with open(fq, "wb") as f:
    for a in range(0, n, 200000):
        b = min(n, a + 200000)
        names = [b"@r%d\n" % i for i in range(a, b)]
        rec = np.empty((b - a, 2 * L + 4), dtype=np.uint8)
        rec[:, :L] = host[a:b]; rec[:, L] = 10; rec[:, L + 1] = ord("+"); rec[:, L + 2] = 10; rec[:, L + 3:2 * L + 3] = ord("B"); rec[:, 2 * L + 3] = 10
        f.write(b"".join(nm + bytes(r) for nm, r in zip(names, rec)))
print("fastq written", time.time() - t, "s", os.path.getsize(fq) / 1e6, "MB")
del blob
torch.cuda.empty_cache()
for extra, outp in (([], f"{wd}/out.sam"), (["-mappers", "3"], f"{wd}/out.sam"), (["-batch", "524288", "-mappers", "3"], f"{wd}/out.sam"), (["-mappers", "1"], f"{wd}/out.sam"), (["-B"], f"{wd}/out.bam")):
    t = time.time()
    r = subprocess.run(["abismal_amd/abismal-amd", "map", "-v", "-i", idx, "-o", outp, "-s", f"{wd}/out.stats"] + extra + [fq],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.time() - t
    print(extra, outp, "rc", r.returncode, f"wall {dt:.2f}s -> {n/dt/1e6:.2f} M reads/s incl. index load;", " | ".join(r.stdout.strip().split("\n")[-2:])[:400])
print(open(f"{wd}/out.stats").read()[:300])
