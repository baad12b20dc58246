#!/bin/bash
# round 5: the single-end kernel with and without window records, REPS processes each, per-launch times (HIP events)
set -u
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/r05_exp_window_records_reps.log}
: > $OUT
export ABM_EXPERIMENTS=1
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1   # builds the index once
for rep in $(seq 1 ${REPS:-4}); do
  for v in ${VARIANTS:-100 0}; do
    ABM_WINDOW_RECORDS=$v python bench.py --steps ${STEPS:-6} --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split ${BENCH_EXTRA:-} 2>gpurun_out/r05_wrec_$v.err | tail -1 > gpurun_out/r05_wrec_reps.json
    python - "$v" "$rep" gpurun_out/r05_wrec_reps.json <<'PY' | tee -a $OUT
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
    r = d["roofline"]
    print("records", v, "rep", rep, "ms/step", d["ms_per_step"], "kernel avg", r["avg_kernel_ms"], "per launch", r.get("kernel_ms_per_launch"))
except Exception as e:
    print("records", v, "rep", rep, "FAILED", e)
PY
  done
done
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -8 >> $OUT
