#!/bin/bash
# round 4, call 13: who asks the L2s for the lines the pair kernels' tallies do not explain -- instruction fetch? scalar loads?
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
rocprofv3 --list-counters 2>/dev/null | grep -oE "(SQC_[A-Z0-9_]+|TCP_TCC_[A-Z0-9_]+|TCC_EA0_RDREQ[A-Z0-9_]*|TCC_REQ[A-Z0-9_]*|TCC_READ[A-Z0-9_]*|TCC_HIT[A-Z0-9_]*|TCC_MISS[A-Z0-9_]*|SQ_INSTS_VMEM_RD|SQ_INSTS_SMEM|SQ_IFETCH[A-Z0-9_]*)" | sort -u | tr '\n' ' ' > gpurun_out/r04_counter_names.log
cat gpurun_out/r04_counter_names.log
cd /tmp
for pass in "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_REQ TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/prof_pe
  (cd "$REPO" && timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_pe -- python3 bench.py --pe --reads 1000000 --read-len 150 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e > /tmp/pe_pmc.log 2>&1)
  CC=$(find /tmp/prof_pe -name '*counter_collection.csv' | head -1)
  if [ -n "$CC" ]; then
    python3 - "$CC" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); secs = collections.defaultdict(float); seen = set()
for row in csv.DictReader(open(sys.argv[1])):
    name = row.get("Kernel_Name", "")
    if "map_pe_kernel" not in name: continue
    tier = "tier2" if "map_pe_kernel<true" in name else "tier1"
    acc[tier][row["Counter_Name"]] += float(row["Counter_Value"])
    key = (row["Dispatch_Id"], tier)
    if key not in seen:
        seen.add(key); secs[tier] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
for tier in sorted(acc):
    print(tier, "seconds under pmc %.3f" % secs[tier], {k: "%.4g" % v for k, v in acc[tier].items()})
PY
  else echo "no counters for $pass"; tail -5 /tmp/pe_pmc.log; fi
done > "$REPO/gpurun_out/r04_pe_pmc_who.log" 2>&1
cat "$REPO/gpurun_out/r04_pe_pmc_who.log"
