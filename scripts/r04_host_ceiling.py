#!/usr/bin/env python3
"""What the host pipeline of `abismal-amd map` carries around N (virtual) GPUs on this box: -virtual-gpus runs (no
device, made-up hits) of 40 M reads x 100 bp over sinks, part counts, host-thread counts and pinning, median of --reps
runs each with the spread.  Needs no GPU.  python3 scripts/r04_host_ceiling.py [--reads-m 40] [--reps 5] [--quick]"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "abismal_amd", "abismal-amd")
ap = argparse.ArgumentParser()
ap.add_argument("--reads-m", type=int, default=40)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--wd", default="/dev/shm/abm_ceiling")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_host_ceiling.json"))
ap.add_argument("--quick", action="store_true")
ap.add_argument("--only", default="")
ap.add_argument("--final", action="store_true", help="the short list: the rows DESIGN quotes")
a = ap.parse_args()
os.makedirs(a.wd, exist_ok=True)
os.makedirs(os.path.dirname(a.out), exist_ok=True)
fa = os.path.join(ROOT, "tests", "golden", "tRex1.fa")
idx = os.path.join(a.wd, "tRex1.idx")
fq = os.path.join(a.wd, "reads_1.fq")
if not os.path.exists(idx):
    subprocess.run([CLI, "idx", "-t", "32", fa, idx], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
if not os.path.exists(fq):
    base = os.path.join(a.wd, "base")
    subprocess.run([CLI, "sim", "-single", "-seed", "1", "-n", "10000000", "-l", "100", "-m", "0.01", "-b", "0.98", "-o", base, fa],
                   check=True, stdout=subprocess.DEVNULL)
    with open(fq, "wb") as out:
        for _ in range(max(1, a.reads_m // 10)):
            with open(base + "_1.fq", "rb") as f:
                while True:
                    blk = f.read(64 << 20)
                    if not blk:
                        break
                    out.write(blk)
    os.remove(base + "_1.fq")
n_reads = max(1, a.reads_m // 10) * 10_000_000
rows = []


FINAL = ("8 parts on tmpfs, -t 8", "8 parts on tmpfs, -t 16", "8 parts on tmpfs, -t 64", "8 parts on tmpfs, -t 256", "one tmpfs file, -t 16", "one tmpfs file, -t 64",
         "/dev/null, -t 16", "/dev/null, -t 64", "/dev/null, -t 256", "BAM", "1 vGPU", "default -t, not pinned")


def one(label, args, env=None, sink="tmpfs"):
    if a.only and a.only not in label:
        return
    if a.final and not any(k in label for k in FINAL):
        return
    rates, busy, cpu = [], None, None
    out = "/dev/null" if sink == "null" else os.path.join(a.wd, "out.sam")
    for _ in range(a.reps):
        for f in os.listdir(a.wd):
            if f.startswith("out.sam"):
                os.remove(os.path.join(a.wd, f))
        r = subprocess.run([CLI, "map"] + args + ["-i", idx, "-o", out, "-timing", os.path.join(a.wd, "t.json"), fq],
                           env=dict(os.environ, **(env or {})), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print(label, "FAILED", r.stderr[-400:], flush=True)
            return
        t = json.load(open(os.path.join(a.wd, "t.json")))
        assert t["reads"] == n_reads, t
        rates.append(t["reads"] / t["seconds"] / 1e6)
        busy, cpu = t["busy_s"], t["cpu_s"]
    row = {"label": label, "args": args, "env": env or {}, "sink": sink, "reads": n_reads, "M_reads_per_s_median": round(statistics.median(rates), 2),
           "runs": [round(x, 2) for x in rates], "host_threads": t["host_threads"], "cpu_quota_cpus": t.get("cpu_quota_cpus"),
           "throttled_periods_last": t.get("throttled_periods"), "throttled_s_last": t.get("throttled_s"), "gpus": t["gpus"], "out_parts": t["out_parts"],
           "out_GB": round(t["out_bytes"] / 1e9, 2), "busy_s_last": {k: round(v, 2) for k, v in busy.items()},
           "cpu_s_last": {k: round(v, 2) for k, v in cpu.items()}}
    rows.append(row)
    print(f"{label:58s} {row['M_reads_per_s_median']:7.2f} M reads/s  runs {row['runs']}  workers {row['host_threads']}  throttled {row['throttled_periods_last']} periods {row['throttled_s_last']:.2f} s  busy {row['busy_s_last']} cpu {row['cpu_s_last']}", flush=True)
    json.dump(rows, open(a.out, "w"), indent=1)


for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip(), flush=True)
threads = [8, 16, 32, 64, 128, 256]
for th in threads:
    one(f"8 vGPUs, 8 parts on tmpfs, -t {th}", ["-virtual-gpus", "8", "-out-parts", "8", "-t", str(th)])
for th in threads:
    one(f"8 vGPUs, one tmpfs file, -t {th}", ["-virtual-gpus", "8", "-t", str(th)])
for th in threads:
    one(f"8 vGPUs, /dev/null, -t {th}", ["-virtual-gpus", "8", "-t", str(th)], sink="null")
one("8 vGPUs, 8 parts on tmpfs, default -t", ["-virtual-gpus", "8", "-out-parts", "8"])
one("8 vGPUs, 2 parts on tmpfs, default -t", ["-virtual-gpus", "8", "-out-parts", "2"])
one("8 vGPUs, 4 parts on tmpfs, default -t", ["-virtual-gpus", "8", "-out-parts", "4"])
one("8 vGPUs, one tmpfs file, default -t", ["-virtual-gpus", "8"])
one("1 vGPU, one tmpfs file, default -t", ["-virtual-gpus", "1"])
one("1 vGPU, /dev/null, default -t", ["-virtual-gpus", "1"], sink="null")
one("2 vGPUs, 2 parts on tmpfs, default -t", ["-virtual-gpus", "2", "-out-parts", "2"])
one("4 vGPUs, 4 parts on tmpfs, default -t", ["-virtual-gpus", "4", "-out-parts", "4"])
# what the container's CPU quota does to a process with more runnable threads than it pays for (the clamp off)
for th in (32, 64, 128):
    one(f"8 vGPUs, 8 parts, -t {th}, quota clamp OFF", ["-virtual-gpus", "8", "-out-parts", "8", "-t", str(th)], env={"ABM_CLI_NO_QUOTA_CLAMP": "1"})
one("8 vGPUs, 8 parts, -t 64, clamp OFF, not pinned", ["-virtual-gpus", "8", "-out-parts", "8", "-t", "64"], env={"ABM_CLI_NO_QUOTA_CLAMP": "1", "ABM_CLI_PIN": "0"})
one("8 vGPUs, 8 parts, default -t, not pinned", ["-virtual-gpus", "8", "-out-parts", "8"], env={"ABM_CLI_PIN": "0"})
one("8 vGPUs, 8 parts, default -t, input through pread (no mapping)", ["-virtual-gpus", "8", "-out-parts", "8"], env={"ABM_CLI_NO_MMAP": "1"})
one("8 vGPUs, 8 parts BAM (-B) on tmpfs, default -t", ["-virtual-gpus", "8", "-out-parts", "8", "-B"])
one("8 vGPUs, 8 parts BAM (-B -z 2: zlib) on tmpfs, default -t", ["-virtual-gpus", "8", "-out-parts", "8", "-B", "-z", "2"])
one("8 vGPUs, 8 parts BAM (-B -z 0: stored) on tmpfs, default -t", ["-virtual-gpus", "8", "-out-parts", "8", "-B", "-z", "0"])
one("8 vGPUs, one BAM (-B) on tmpfs, default -t", ["-virtual-gpus", "8", "-B"])
for f in os.listdir(a.wd):
    if f.startswith("out.sam"):
        os.remove(os.path.join(a.wd, f))
print("wrote", a.out)
