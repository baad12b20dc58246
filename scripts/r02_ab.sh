#!/bin/bash
# same-box A/B of library variants in abismal_amd/_ab/ on the default bench workload (un-profiled kernel time)
set -u
export ABM_BENCH_GENOME_MBP=3100
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1   # builds the index once
for rep in 1 2; do
  for v in ${VARIANTS:-new nc}; do
    cp abismal_amd/_ab/libabismal_amd_$v.so abismal_amd/libabismal_amd.so
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-stage-split --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'rep $rep', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"
  done
done
cp abismal_amd/_ab/libabismal_amd_new.so abismal_amd/libabismal_amd.so
