#!/bin/bash
# kernel time and L2 requests/misses of map_se_kernel for library variants in abismal_amd/_ab/
set -u
export TMPDIR=/tmp ABM_BENCH_GENOME_MBP=3100
REPO=$(pwd)
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1   # builds the index once
for v in ${VARIANTS:-new fn nn fnnn}; do
  cp abismal_amd/_ab/libabismal_amd_$v.so abismal_amd/libabismal_amd.so
  rm -rf /tmp/prof_ab
  (cd /tmp && rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/prof_ab -- python3 $REPO/bench.py --no-cpu-baseline --no-e2e --no-stage-split --steps 2 --warmup 0 > /tmp/ab.log 2>&1)
  CC=$(find /tmp/prof_ab -name '*counter_collection.csv' | head -1)
  python3 - "$CC" "$v" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); dur = []
for row in csv.DictReader(open(sys.argv[1])):
    if "map_se_kernel" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "TCC_REQ_sum": dur.append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6)
n = max(1, len(dur))
print(sys.argv[2], "kernel ms", [round(d) for d in dur], {k: f"{v / n:.4g}" for k, v in acc.items()})
PY
done
cp abismal_amd/_ab/libabismal_amd_new.so abismal_amd/libabismal_amd.so
