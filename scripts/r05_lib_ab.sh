#!/bin/bash
# round 5: same-box A/B of library builds (abismal_amd/_ab/libabismal_amd_<name>.so; "tree" = the tree's own) on the default
# single-end workload with window records: per-launch kernel times.  VARIANTS="tree w4 ...", REPS=n, STEPS=n
set -u
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/r05_exp_lib_ab.log}
[ -n "${APPEND:-}" ] || : > $OUT
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1   # builds the index once
for rep in $(seq 1 ${REPS:-2}); do
  for v in ${VARIANTS:-tree}; do
    unset ABISMAL_AMD_LIB
    [ "$v" != tree ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so
    python bench.py --steps ${STEPS:-4} --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split ${BENCH_EXTRA:-} 2>gpurun_out/r05_lib_ab_$v.err | tail -1 > gpurun_out/r05_lib_ab.json
    python - "$v" "$rep" gpurun_out/r05_lib_ab.json <<'PY' | tee -a $OUT
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
    r = d["roofline"]
    print("build", v, "rep", rep, "ms/step", d["ms_per_step"], "kernel avg", r["avg_kernel_ms"], "per launch", r.get("kernel_ms_per_launch"))
except Exception as e:
    print("build", v, "rep", rep, "FAILED", e)
PY
  done
done
