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
fqs = [fq]
if os.environ.get("ABM_CLI_PE"):
    # paired-end input: n/2 pairs of 2 x 150 bp from 150-500 bp fragments
    npairs, Lp = n // 2, 150
    b1, b2 = bench.sample_pairs(gw, starts, npairs, Lp, 2000, dev)
    fqs = []
    for tag, bl in (("1", b1), ("2", b2)):
        hostp = bl.cpu().numpy().reshape(npairs, Lp)
        path = f"{wd}/pairs_{npairs}_{tag}.fq"
        with open(path, "wb") as f:
            for a in range(0, npairs, 200000):
                b = min(npairs, a + 200000)
                rec = np.empty((b - a, 2 * Lp + 4), dtype=np.uint8)
                rec[:, :Lp] = hostp[a:b]; rec[:, Lp] = 10; rec[:, Lp + 1] = ord("+"); rec[:, Lp + 2] = 10; rec[:, Lp + 3:2 * Lp + 3] = ord("B"); rec[:, 2 * Lp + 3] = 10
                f.write(b"".join(b"@p%d\n" % i + bytes(r) for i, r in zip(range(a, b), rec)))
        fqs.append(path)
    del b1, b2
    torch.cuda.empty_cache()
    n = 2 * npairs
variants = (([], f"{wd}/out.sam"), (["-mappers", "3"], f"{wd}/out.sam"), (["-B"], f"{wd}/out.bam"))
if os.environ.get("ABM_CLI_BATCH"):
    variants = ((["-batch", os.environ["ABM_CLI_BATCH"]], f"{wd}/out.sam"),)
for extra, outp in variants:
    t = time.time()
    r = subprocess.run(["abismal_amd/abismal-amd", "map", "-v", "-i", idx, "-o", outp, "-s", f"{wd}/out.stats"] + extra + fqs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    dt = time.time() - t
    if os.environ.get("ABM_TRACE_HOST"):
        print("\n".join(r.stdout.strip().split("\n")[-70:]))
        break
    print(extra, outp, "rc", r.returncode, f"wall {dt:.2f}s -> {n/dt/1e6:.2f} M reads/s incl. index load;", " | ".join(r.stdout.strip().split("\n")[-2:])[:400])
print(open(f"{wd}/out.stats").read()[:300])
