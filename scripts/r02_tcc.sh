#!/bin/bash
export ABM_EXPERIMENTS=1  # (the experiment variables below are honoured only with this set: abm_api.hip, experiment_env)
# L2 / fabric read requests of map_se_kernel by size, current build, for a few environment variants
set -u
export TMPDIR=/tmp ABM_BENCH_GENOME_MBP=3100
REPO=$(pwd)
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1   # builds the index once
run() {
  for pass in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum"; do
    rm -rf /tmp/prof_ab
    (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_ab -- python3 $REPO/bench.py --no-cpu-baseline --no-e2e --no-stage-split --steps 1 --warmup 0 > /tmp/ab.log 2>&1)
    CC=$(find /tmp/prof_ab -name '*counter_collection.csv' | head -1)
    if [ -n "$CC" ]; then
      python3 - "$CC" "$1" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); dur = 0
for row in csv.DictReader(open(sys.argv[1])):
    if "map_se_kernel" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); dur = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6
print(sys.argv[2], f"{dur:.0f} ms", {k: f"{v:.4g}" for k, v in acc.items()})
PY
    else echo "$1: no counters for: $pass"; tail -3 /tmp/ab.log; fi
  done
}
run "two copies"
ABM_PLANES_COPIES=1 run "one copy"
