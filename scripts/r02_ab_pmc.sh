#!/bin/bash
# A/B of two library builds (abismal_amd/_ab/libabismal_amd_{old,new}.so) on the default bench workload: kernel time
# un-profiled, then SQ and TCC counters of map_se_kernel in separate rocprofv3 --pmc passes (one launch each)
set -u
export TMPDIR=/tmp ABM_BENCH_GENOME_MBP=3100
REPO=$(pwd)
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1   # builds the index once
for v in ${VARIANTS:-old new}; do
  cp abismal_amd/_ab/libabismal_amd_$v.so abismal_amd/libabismal_amd.so
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"
  for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
              "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    rm -rf /tmp/prof_ab
    (cd /tmp && rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/prof_ab -- python3 $REPO/bench.py --no-cpu-baseline --no-e2e --no-stage-split --steps 1 --warmup 0 > /tmp/ab.log 2>&1)
    CC=$(find /tmp/prof_ab -name '*counter_collection.csv' | head -1)
    if [ -n "$CC" ]; then
      python3 - "$CC" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float)
for row in csv.DictReader(open(sys.argv[1])):
    if "map_se_kernel" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
print(sys.argv[2], {k: f"{v:.4g}" for k, v in acc.items()})
PY
    else echo "$v: no counters for: $pass"; tail -3 /tmp/ab.log; fi
  done
done
cp abismal_amd/_ab/libabismal_amd_new.so abismal_amd/libabismal_amd.so
