"""Copy the summaries scripts/profile_round.sh left in gpurun_out/prof/ into profiles/ (tracked)."""
import csv, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, rnd = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles"), int(sys.argv[1]) if len(sys.argv) > 1 else 1
tag = f"r{rnd:02d}"
def json_line(path):
    for ln in open(path):
        if ln.startswith('{"metric"'):
            return json.loads(ln)
    raise SystemExit("no bench line in " + path)
line = json_line(os.path.join(src, "bench_line_under_rocprof.log"))
json.dump(line, open(os.path.join(dst, f"{tag}_bench_line_under_rocprof.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_header.csv"), os.path.join(dst, f"{tag}_pmc_header.csv"))
shutil.copy(os.path.join(src, "pmc_rdreq_map_se.csv"), os.path.join(dst, f"{tag}_pmc_rdreq_map_se.csv"))
hdr = next(csv.reader(open(os.path.join(src, "pmc_header.csv"))))
cnt, secs = {}, None
for row in csv.reader(open(os.path.join(src, "pmc_rdreq_map_se.csv"))):
    r = dict(zip(hdr, row))
    cnt[r["Counter_Name"]] = int(float(r["Counter_Value"]))
    secs = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
pmc_line = json_line(os.path.join(src, "bench_line_under_pmc.log"))
wl = line["config"]
nbytes = cnt["TCC_EA0_RDREQ_128B_sum"] * 128 + cnt["TCC_EA0_RDREQ_64B_sum"] * 64 + cnt["TCC_EA0_RDREQ_32B_sum"] * 32
traffic = {"round": rnd, "kernel": "map_se_kernel",
           "workload": {"genome_mbp": 3100, "reads": wl["reads_per_step_per_gpu"], "read_len": wl["read_len"]},
           "counters": cnt, "hbm_read_bytes_per_launch": nbytes, "kernel_seconds_under_pmc": secs,
           "note": "rocprofv3 --pmc TCC_EA0_RDREQ_{,32B,64B,128B}_sum in a pass of its own over `python3 bench.py --steps 1 "
                   "--warmup 0 --no-cpu-baseline` (scripts/profile_round.sh; rows in %s_pmc_rdreq_map_se.csv). Essentially every "
                   "L2->HBM read request of this kernel is a 128-B line; an earlier FETCH_SIZE pass reported requests x 64 B, "
                   "i.e. exactly half, as MI355X_MICROARCH.md says for gfx950. Write traffic is negligible (8 B + CIGAR slot "
                   "per read)." % tag}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
rows = list(csv.DictReader(open(os.path.join(src, "bench_kernel_stats.csv"))))[:6]
calls_note = ""
cpath = os.path.join(src, "map_se_calls.csv")
if os.path.exists(cpath):
    shutil.copy(cpath, os.path.join(dst, f"{tag}_map_se_calls.csv"))
    cr = list(csv.DictReader(open(cpath)))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in cr]
    if len(durs) > line["steps"]:
        timed = durs[-line["steps"]:]
        calls_note = (f"  Per launch (`{tag}_map_se_calls.csv`): " + ", ".join(f"{d:.1f}" for d in durs) +
                      f" ms -- every step maps a different batch; the first launch is the warm-up, and the {line['steps']} timed "
                      f"launches average {sum(timed) / len(timed):.1f} ms in the rocprofv3 trace.")
rf, cb = line["roofline"], line["cpu_baseline"]
with open(os.path.join(dst, f"{tag}_README.md"), "w") as f:
    f.write(f"# Round {rnd} profiles\n\n`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py` (default flags: 3100 Mbp "
            "synthetic hg38-shaped genome, 10 M reads x 100 bp per step, a different batch every step, 3 steps + 1 warm-up) on one "
            "MI355X; commands in `scripts/profile_round.sh`.\n\n| kernel | calls | avg ms | % |\n|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {r['Percentage']} |\n")
    f.write(f"\nbench.py's own line from that run (`{tag}_bench_line_under_rocprof.json`): value {line['value']:.0f} reads/s, HIP-event "
            f"average of map_se_kernel {rf['avg_kernel_ms']} ms (compare the rocprofv3 average above), algorithmic {rf['achieved']} GB/s = "
            f"{rf['frac']} of 8 TB/s, cpu_baseline {cb['value']} reads/s on {cb['cores']} threads ({cb['positions_identical_to_gpu']} "
            "positions identical)." + calls_note + "\n\n"
            f"HBM traffic (`{tag}_traffic.json`, separate `--pmc` pass, `{tag}_pmc_rdreq_map_se.csv`): {nbytes / 1e12:.2f} TB of 128-byte "
            f"line reads per launch = {nbytes / secs / 1e12:.2f} TB/s during the kernel ({secs * 1e3:.0f} ms under the counters).\n\n"
            "Work per read in that run: " + ", ".join(f"{k} {v}" for k, v in line["work_per_read"].items()) + ".\n\n"
            "Hardware probes used for the roofline discussion (`tests/hip/*.hip`, run by hand): random 8-byte gathers 56 G/s, random "
            "72-byte windows 37.6 G/s over 1.5 GB; a flag stored by a running kernel to pinned host memory is seen by the host "
            "within 0.03-0.4 ms (`flag_probe.hip`).\n")
print(open(os.path.join(dst, f"{tag}_README.md")).read())
