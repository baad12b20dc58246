#!/bin/bash
# round 5, call 15: one lane per candidate for reads of up to 172 bases as well (the pair kernels' seed passes, 150-base
# single-end reads) -- the pair and single-end parity tests, then this build against the one before it on one box: pairs
# 2x150 (1 M pairs per step, 16 slots) and 150-base random-PBAT reads (4 M per step)
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_window_records.py tests/test_gpu_se_parity.py tests/test_gpu_pe_parity.py tests/test_gpu_pe_split.py tests/test_gpu_scale_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_params.py tests/test_gpu_seed_extension.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call15_tests.log
cat gpurun_out/r05_call15_tests.log
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_one_lane_filter_150.log
: > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
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
