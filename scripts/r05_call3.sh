#!/bin/bash
# round 5, call 3: the three-phase packed scoring loop (wavefront_quad) -- unit check, parity, then old loop vs new loop on one box,
# single-end (10 M x 100 bp) and paired-end (1 M pairs 2x150)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_se_set.py tests/test_gpu_pe_split.py tests/test_gpu_pe_parity.py tests/test_gpu_se_parity.py tests/test_gpu_scale_parity.py -x -q 2>&1 | tail -15 > gpurun_out/r05_call3_tests.log
cat gpurun_out/r05_call3_tests.log
export ABM_BENCH_GENOME_MBP=3100
{
for rep in 1 2; do
  for v in old new; do
    unset ABISMAL_AMD_LIB
    [ $v = old ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_mate3.so
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline 2> gpurun_out/r05_call3_se_$v.err | tail -1 > gpurun_out/r05_call3_se_${v}_$rep.json
    python3 -c "
import json; d=json.load(open('gpurun_out/r05_call3_se_${v}_$rep.json')); r=d['roofline']
print('SE scoring loop $v rep $rep: %.3f M reads/s  kernel %.1f ms  stages %s' % (d['value']/1e6, r['avg_kernel_ms'], [(s['stage'], s['ms']) for s in (r.get('stages') or [])]))"
  done
done
} 2>&1 | tee gpurun_out/r05_exp_scoring_loop.log
unset ABISMAL_AMD_LIB
OUT=gpurun_out/r05_exp_scoring_loop.log FORMS="split@mate3 split" REPS=2 scripts/r05_pe_forms.sh
