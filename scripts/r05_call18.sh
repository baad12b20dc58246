#!/bin/bash
# round 5, call 18: window records as two tables (low bits / high bits), the specific passes staged on the low bits --
# parity tests, then this build against the one before it: single-end 100 bp, pairs 2x150, 150 bp random PBAT
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_window_records.py tests/test_gpu_se_parity.py tests/test_gpu_pe_parity.py tests/test_gpu_pe_split.py tests/test_gpu_scale_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_params.py tests/test_gpu_seed_extension.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call18_tests.log
cat gpurun_out/r05_call18_tests.log
OUT=gpurun_out/r05_exp_staged_records.log VARIANTS="prev tree" REPS=2 scripts/r05_lib_ab.sh
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_staged_records.log
for rep in 1 2; do
  for v in prev tree; do
    unset ABISMAL_AMD_LIB
    [ "$v" != tree ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so
    python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("pairs 2x150, build %-5s rep %s  %.3f M reads/s  %.1f ms/step  alone %s" % (f, rep, d["value"] / 1e6, d["ms_per_step"], (d.get("phase_stamps") or {}).get("kernel_ms")))
except Exception as e:
    print("pairs, build", f, "rep", rep, "FAILED", e)
PY
    python bench.py --mode random --read-len 150 --reads 4000000 --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("150 bp -R,   build %-5s rep %s  %.3f M reads/s  kernel %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch")))
except Exception as e:
    print("150 bp, build", f, "rep", rep, "FAILED", e)
PY
  done
done
