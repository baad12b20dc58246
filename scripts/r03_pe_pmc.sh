#!/bin/bash
# memory-side read requests of the paired-end kernels (both tiers) for one step of 1 M pairs 2x150: a rocprofv3 --pmc
# pass of its own over `bench.py --pe --steps 1 --warmup 0` (counters serialise the launches; the rate is bench.py's)
set -u
export TMPDIR=/tmp ABM_BENCH_GENOME_MBP=3100
REPO=$(pwd)
rm -rf /tmp/prof_pe
(cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/prof_pe -- python3 $REPO/bench.py --pe --reads 1000000 --read-len 150 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline > /tmp/pe_pmc.log 2>&1)
CC=$(find /tmp/prof_pe -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then
  python3 - "$CC" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); secs = collections.defaultdict(float); seen = set()
for row in csv.DictReader(open(sys.argv[1])):
    name = row.get("Kernel_Name", "")
    if "map_pe_kernel" not in name: continue
    tier = "tier2" if "map_pe_kernel<true" in name else "tier1"
    acc[tier][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"], tier)
    if key not in seen:
        seen.add(key); secs[tier] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
out = {}
for tier in acc:
    c = acc[tier]
    nbytes = c["TCC_EA0_RDREQ_128B_sum"] * 128 + c["TCC_EA0_RDREQ_64B_sum"] * 64 + c["TCC_EA0_RDREQ_32B_sum"] * 32
    out[tier] = {"requests": c["TCC_EA0_RDREQ_sum"], "bytes": nbytes, "kernel_seconds_under_pmc": round(secs[tier], 4), "bytes_per_pair": nbytes / 1e6}
print(json.dumps({"workload": "1 M pairs 2x150, hg38-shaped 3.1 Gbp index, one step", "kernels": out}))
PY
else echo "no counters"; tail -5 /tmp/pe_pmc.log; fi
tail -2 /tmp/pe_pmc.log | cut -c1-400
