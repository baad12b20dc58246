#!/bin/bash
# round 5, call 17: tier 2's lists sorted by radix passes -- the pair tests, then this build against the one before it
# (pairs 2x150, 1 M pairs per step, 16 slots)
set -u
mkdir -p gpurun_out
[ -n "${SKIP_TESTS:-}" ] || timeout 1500 python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_pe_split.py tests/test_gpu_scale_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_edges_and_properties.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call17_tests.log
cat gpurun_out/r05_call17_tests.log
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_radix_sort.log
[ -n "${SKIP_TESTS:-}" ] || : > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${ORDER:-prev tree}; do
    unset ABISMAL_AMD_LIB
    [ "$v" != tree ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so
    python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --cpu-sample 40000 --no-e2e 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("pairs 2x150, build %-5s rep %s  %.3f M reads/s  %.1f ms/step  alone %s  parity %s" % (f, rep, d["value"] / 1e6, d["ms_per_step"], (d.get("phase_stamps") or {}).get("kernel_ms"), {k: v for k, v in (d.get("cpu_baseline") or {}).items() if "identical" in k or "vs_oracle" in k}))
except Exception as e:
    print("pairs, build", f, "rep", rep, "FAILED", e)
PY
  done
done
