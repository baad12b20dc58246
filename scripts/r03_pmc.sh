#!/bin/bash
# SQ and TCC counters of map_se_kernel for the installed library (separate rocprofv3 --pmc passes, one launch each)
set -u
export TMPDIR=/tmp ABM_BENCH_GENOME_MBP=3100
REPO=$(pwd)
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
            "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/prof_ab
  (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_ab -- python3 $REPO/bench.py --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split --steps 1 --warmup 0 > /tmp/ab.log 2>&1)
  CC=$(find /tmp/prof_ab -name '*counter_collection.csv' | head -1)
  if [ -n "$CC" ]; then
    python3 - "$CC" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); info = {}
for row in csv.DictReader(open(sys.argv[1])):
    if "map_se_kernel<false" in row.get("Kernel_Name", ""):  # (the production kernel; the diagnostic build runs once after it)
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"):
            if k in row: info[k] = row[k]
print({k: f"{v:.4g}" for k, v in acc.items()}, info)
PY
  else echo "no counters for: $pass"; tail -3 /tmp/ab.log; fi
done
