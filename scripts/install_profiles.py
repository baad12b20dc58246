"""Copy the summaries scripts/profile_round.sh left in gpurun_out/prof/ into profiles/ (tracked)."""
import csv, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst, rnd = os.path.join(root, "gpurun_out", "prof"), os.path.join(root, "profiles"), int(sys.argv[1]) if len(sys.argv) > 1 else 1
tag = f"r{rnd:02d}"
# the build the profile was taken on (scripts/r04_profile.sh copies BUILD_ID, written when the call was launched) must be
# HEAD's as far as the kernels and the bench go: a README generated from an older build misleads whoever reads it
build = "n/a"
bpath = os.path.join(src, "build.txt")
if os.path.exists(bpath):
    build = open(bpath).read().strip()
    if os.system(f"git -C {root} diff --quiet {build} HEAD -- abismal_amd/csrc bench.py") != 0 and "--force" not in sys.argv:
        raise SystemExit(f"profile was taken on build {build}, HEAD differs in abismal_amd/csrc or bench.py: profile again (or --force)")
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
    # the production kernel only (the diagnostic build runs once after it, for the work tallies), its first dispatch
    if "map_se_kernel<false" not in r["Kernel_Name"] or r["Counter_Name"] in cnt:
        continue
    cnt[r["Counter_Name"]] = int(float(r["Counter_Value"]))
    secs = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
pmc_line = json_line(os.path.join(src, "bench_line_under_pmc.log"))
wl = line["config"]
nbytes = cnt["TCC_EA0_RDREQ_128B_sum"] * 128 + cnt["TCC_EA0_RDREQ_64B_sum"] * 64 + cnt["TCC_EA0_RDREQ_32B_sum"] * 32
traffic = {"round": rnd, "kernel": "map_se_kernel",
           "workload": {"kind": "se_trich", "genome_mbp": 3100, "reads": wl["reads_per_step_per_gpu"], "read_len": wl["read_len"]},
           "counters": cnt, "hbm_read_bytes_per_launch": nbytes, "kernel_seconds_under_pmc": secs,
           "build": build,
           "note": "rocprofv3 --pmc TCC_EA0_RDREQ_{,32B,64B,128B}_sum in a pass of its own over `python3 bench.py --steps 1 "
                   "--warmup 0 --no-cpu-baseline` (scripts/profile_round.sh; rows in %s_pmc_rdreq_map_se.csv). Essentially every "
                   "L2->HBM read request of this kernel is a 128-B line; an earlier FETCH_SIZE pass reported requests x 64 B, "
                   "i.e. exactly half, as MI355X_MICROARCH.md says for gfx950. Write traffic is negligible (8 B + CIGAR slot "
                   "per read).  TCC_EA0_RDREQ counts Infinity-Cache hits as well: fabric-side requests, an upper bound on HBM bytes." % tag}
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
# the other configurations' passes (scripts/r04_profile.sh): paired-end (both tiers of one step) and 150 bp random PBAT
def sum_rows(path, want):
    """counters of the LAST dispatch that matches: `bench.py --pe --steps 1 --warmup 0` launches each tier twice -- one step that
    sizes the slot's workspaces, then the timed step -- and rounds 3's figures (and this round's until the last day) added the
    two up: 22.8 k / 18.8 k lines per pair were 11.4 k / 9.4 k"""
    per = {}
    for row in csv.reader(open(path)):
        r = dict(zip(hdr, row))
        if not want(r["Kernel_Name"]):
            continue
        d = per.setdefault(int(r["Dispatch_Id"]), {"acc": {}, "secs": 0.0})
        d["acc"][r["Counter_Name"]] = int(float(r["Counter_Value"]))
        d["secs"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    if not per:
        return {}, 0.0
    last = per[max(per)]
    return last["acc"], last["secs"]
extra = {}
pe_csv = os.path.join(src, "pmc_rdreq_map_pe.csv")
if os.path.exists(pe_csv):
    shutil.copy(pe_csv, os.path.join(dst, f"{tag}_pmc_rdreq_map_pe.csv"))
    # the pair kernels by template arguments <BIG, TIMED, COOP, WPS, LONG, PHASE>: production builds only (TIMED = false; the
    # bench's last step runs the diagnostic ones), PHASE 1 = seed, 2 = mate (BIG: lists in device memory), 0 = whole pairs
    import re
    def which(kname):
        m = re.search(r"map_pe_kernel<([^>]*)>", kname)
        if not m:
            return None
        t = [x.strip() for x in m.group(1).split(",")]
        big, timed, phase = t[0] in ("true", "1"), t[1] in ("true", "1"), int(t[5]) if len(t) > 5 else 0
        if timed:
            return None
        return {(False, 1): "seed", (False, 2): "mate_lds_lists", (True, 2): "mate_device_lists", (True, 0): "whole_pairs", (False, 0): "tier1_unsplit"}.get((big, phase))
    names = ("seed", "mate_lds_lists", "mate_device_lists", "whole_pairs", "tier1_unsplit")
    kernels = {}
    for name in names:
        acc, secs_t = sum_rows(pe_csv, lambda k, name=name: which(k) == name)
        if acc:
            kernels[name] = {"counters": acc, "kernel_seconds_under_pmc": round(secs_t, 4),
                             "bytes": acc.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + acc.get("TCC_EA0_RDREQ_64B_sum", 0) * 64 + acc.get("TCC_EA0_RDREQ_32B_sum", 0) * 32}
    wr_csv = os.path.join(src, "pmc_wrreq_map_pe.csv")
    if os.path.exists(wr_csv):
        shutil.copy(wr_csv, os.path.join(dst, f"{tag}_pmc_wrreq_map_pe.csv"))
        for name in list(kernels):
            kernels[name]["write_side_counters"] = sum_rows(wr_csv, lambda k, name=name: which(k) == name)[0]
    pe_bytes = sum(k["bytes"] for k in kernels.values())
    extra["pe"] = {"round": rnd, "kernel": "map_pe_kernel (seed + mate + whole-pair launches of one step)",
                   "workload": {"kind": "pe", "genome_mbp": 3100, "reads": 1000000, "read_len": 150}, "kernels": kernels,
                   "hbm_read_bytes_per_launch": pe_bytes, "lines_per_pair": pe_bytes / 128 / 1e6, "build": build,
                   "requests_per_second_by_kernel": {n: round(k["counters"].get("TCC_EA0_RDREQ_sum", 0) / k["kernel_seconds_under_pmc"] / 1e9, 2) for n, k in kernels.items() if k["kernel_seconds_under_pmc"]},
                   "note": "rocprofv3 --pmc TCC_EA0_RDREQ_* in a pass of its own over `python3 bench.py --pe --reads 1000000 --read-len 150 --steps 1 "
                           "--warmup 0 --streams 1` (scripts/r05_profile.sh); per step = one launch of each production kernel (the LAST dispatch of "
                           "each: the run's first step only sizes the workspaces, its last one runs the diagnostic builds for the work tallies); "
                           "requests_per_second_by_kernel in G/s, each kernel alone on the device"}
    json.dump(extra["pe"], open(os.path.join(dst, f"{tag}_traffic_pe.json"), "w"), indent=1)
r_csv = os.path.join(src, "pmc_rdreq_map_se_r150.csv")
if os.path.exists(r_csv):
    shutil.copy(r_csv, os.path.join(dst, f"{tag}_pmc_rdreq_map_se_r150.csv"))
    acc, secs_r = {}, None
    for row in csv.reader(open(r_csv)):
        r = dict(zip(hdr, row))
        if "map_se_kernel<false" not in r["Kernel_Name"] or r["Counter_Name"] in acc:
            continue
        acc[r["Counter_Name"]] = int(float(r["Counter_Value"]))
        secs_r = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    rb = acc["TCC_EA0_RDREQ_128B_sum"] * 128 + acc["TCC_EA0_RDREQ_64B_sum"] * 64 + acc["TCC_EA0_RDREQ_32B_sum"] * 32
    extra["r150"] = {"round": rnd, "kernel": "map_se_kernel", "workload": {"kind": "se_random", "genome_mbp": 3100, "reads": 4000000, "read_len": 150},
                     "counters": acc, "hbm_read_bytes_per_launch": rb, "kernel_seconds_under_pmc": secs_r, "build": build,
                     "note": "as the single-end pass, over `python3 bench.py --mode random --read-len 150 --reads 4000000 --steps 1 --warmup 0`"}
    json.dump(extra["r150"], open(os.path.join(dst, f"{tag}_traffic_rpbat150.json"), "w"), indent=1)
who_csv = os.path.join(src, "pmc_who_map_se.csv")
if os.path.exists(who_csv):
    shutil.copy(who_csv, os.path.join(dst, f"{tag}_pmc_who_map_se.csv"))
    acc = {}
    for row in csv.reader(open(who_csv)):
        r = dict(zip(hdr, row))
        if "map_se_kernel<false" in r["Kernel_Name"] and r["Counter_Name"] not in acc:
            acc[r["Counter_Name"]] = int(float(r["Counter_Value"]))
    extra["who"] = acc
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
    f.write(f"# Round {rnd} profiles\n\n`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-e2e` (3100 Mbp "
            "synthetic hg38-shaped genome, 10 M reads x 100 bp per step, a different batch every step, 3 steps + 1 warm-up; the "
            "end-to-end leg runs the CLI as a child process and is kept out of the profiled run) on one MI355X; commands in "
            "`scripts/profile_round.sh`, this file written by `scripts/install_profiles.py`.\n\n| kernel | calls | avg ms | % |\n|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | {r['Percentage']} |\n")
    f.write(f"\nbench.py's own line from that run (`{tag}_bench_line_under_rocprof.json`): value {line['value']:.0f} reads/s, HIP-event "
            f"average of map_se_kernel {rf['avg_kernel_ms']} ms (compare the rocprofv3 average above)." + calls_note + "\n\n")
    f.write("## Recomputing the roofline line\n\n"
            "`roofline.achieved = alg_bytes_per_read x reads per launch / avg_kernel_ms`, `frac = achieved / 8000 GB/s`, with\n"
            "`alg_bytes_per_read = L + S*16 + P*4.5 + C*4 + (W + C)*8 + A*(L + bw)/2 + 8*(1 + ops/2)` (SURVEY.md 8d; L = 100, bw = 21).\n\n"
            "| basis | S | P | C | W | A | ops | bytes/read | GB/s | frac |\n|---|---|---|---|---|---|---|---|---|---|\n")
    sc = rf.get("strict_counts_per_read") or {}
    n_reads = line["config"]["reads_per_step_per_gpu"]
    if sc:
        f.write(f"| strict (oracle counters on the cpu_baseline sample: reference algorithm, early-exit full_compare) | {sc['S']} | {sc['P']} | "
                f"{sc['C']} | {sc['W']} | {sc['A']} | {sc['ops']} | {rf['alg_bytes_per_read_strict']} | "
                f"{rf['alg_bytes_per_read_strict'] * n_reads / rf['avg_kernel_ms'] / 1e6:.1f} | {rf['frac_strict']} |\n")
    wp = line["work_per_read"]
    f.write(f"| kernel tally (what the kernel did: full windows, position-cache hits not fetched) | {wp['seed_offsets']} | {wp['search_probes']} | "
            f"{wp['candidates']} | {wp['read_words']} (of which {wp['window_cache_hits']} candidates cached) | {wp['alignments']} | - | "
            f"{rf['alg_bytes_per_read_kernel_tally']} | {rf['alg_bytes_per_read_kernel_tally'] * n_reads / rf['avg_kernel_ms'] / 1e6:.1f} | "
            f"{rf['frac_kernel_tally']} |\n\n")
    if rf.get("stages"):
        f.write("Per stage (time shares from one extra step of the s_memtime-stamped diagnostic kernel, applied to the real kernel's "
                "average duration; bytes = that stage's terms of the formula):\n\n| stage | share | ms | bytes/read | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|\n")
        for st in rf["stages"]:
            f.write(f"| {st['stage']} | {st['time_share']} | {st['ms']} | {st['alg_bytes_per_read']} | {st['achieved_GBps']} | {st['frac']} |\n")
        f.write("\n")
    f.write(f"cpu_baseline: {cb['value']} reads/s on {cb['cores']} threads ({cb['sample']}); identical to the GPU on the sample: positions "
            f"{cb['positions_identical_to_gpu']}, + diffs and flags {cb['pos_diffs_flags_identical_to_gpu']}, + CIGARs "
            f"{cb['pos_diffs_flags_cigar_identical_to_gpu']} (long CIGARs through the arena: {cb['long_cigars_checked_through_the_arena']}).\n\n")
    f.write(f"## Memory-side traffic\n\n`{tag}_traffic.json` (separate `--pmc` pass, rows in `{tag}_pmc_rdreq_map_se.csv`): "
            f"{cnt['TCC_EA0_RDREQ_sum'] / 1e9:.2f} G read requests from the L2s to the fabric per launch, all of them 128-byte lines = "
            f"{nbytes / 1e12:.2f} TB = {nbytes / secs / 1e12:.2f} TB/s during the kernel ({secs * 1e3:.0f} ms under the counters).  "
            "`TCC_EA0_RDREQ` counts Infinity-Cache hits too, so this is fabric request traffic and an UPPER bound on HBM bytes "
            f"(MI355X_MICROARCH.md, HBM).  It is {nbytes / (rf['alg_bytes_per_read_kernel_tally'] * n_reads):.1f}x the kernel-tally bytes and "
            f"{nbytes / (rf['alg_bytes_per_read_strict'] * n_reads):.1f}x the strict bytes: every 8-64-byte gather costs a whole 128-byte line.\n\n")
    f.write("Work per read in that run: " + ", ".join(f"{k} {v}" for k, v in line["work_per_read"].items()) + ".\n")
    if "pe" in extra:
        ks = extra["pe"]["kernels"]
        f.write(f"\nPaired-end (config 3, `{tag}_traffic_pe.json`), read requests per step of 1 M pairs by kernel, each alone on the device: " +
                "; ".join(f"{n} {k['counters'].get('TCC_EA0_RDREQ_sum', 0) / 1e9:.2f} G in {k['kernel_seconds_under_pmc'] * 1e3:.0f} ms = {extra['pe']['requests_per_second_by_kernel'].get(n, 0)} G/s" for n, k in ks.items()) +
                f" = {extra['pe']['lines_per_pair'] / 1e3:.1f} k lines per pair.\n")
    if "r150" in extra:
        f.write(f"\n150 bp random PBAT (config 5, `{tag}_traffic_rpbat150.json`): {extra['r150']['counters']['TCC_EA0_RDREQ_sum'] / 1e9:.2f} G read requests per launch of 4 M reads "
                f"= {extra['r150']['counters']['TCC_EA0_RDREQ_sum'] / 4e6:.0f} lines per read.\n")
    if "who" in extra:
        w = extra["who"]
        f.write(f"\nWho asks the L2s (single-end kernel, one launch of 10 M reads, `{tag}_pmc_who_map_se.csv`): " + ", ".join(f"{k} {v / 1e9:.2f} G" for k, v in w.items()) + ".\n")
    f.write(f"\nBuild: {build}.\n")
print(open(os.path.join(dst, f"{tag}_README.md")).read())
