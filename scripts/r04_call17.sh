#!/bin/bash
# round 4, call 17: the pair kernels' memory-side requests by phase (phases switched off in the kernels: results are garbage, counters are not)
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call17_index.err
cd /tmp
for skip in 0 1 3 2 4; do
  rm -rf /tmp/prof_pe
  (cd "$REPO" && ABM_EXPERIMENTS=1 ABM_PE_DIAG_SKIP=$skip timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQC_TC_INST_REQ --kernel-trace --output-format csv -d /tmp/prof_pe -- python3 bench.py --pe --reads 1000000 --read-len 150 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e > /tmp/pe_pmc.log 2>&1)
  CC=$(find /tmp/prof_pe -name '*counter_collection.csv' | head -1)
  if [ -n "$CC" ]; then
    python3 - "$CC" $skip <<'PY'
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
    print("skip", sys.argv[2], tier, "seconds under pmc %.3f" % secs[tier], {k: "%.4g" % v for k, v in acc[tier].items()})
PY
  else echo "no counters for skip $skip"; tail -5 /tmp/pe_pmc.log; fi
done > "$REPO/gpurun_out/r04_pe_pmc_phases.log" 2>&1
cat "$REPO/gpurun_out/r04_pe_pmc_phases.log"
