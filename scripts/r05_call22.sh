#!/bin/bash
# round 5, call 22: the final single-end kernel built for six waves per SIMD against the tree's five; the pair kernels'
# direct narrowing from 64 entries on against the default 128
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r05_exp_final_knobs.log VARIANTS="tree w6" REPS=2 scripts/r05_lib_ab.sh
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_final_knobs.log
for rep in 1 2; do
  for v in 128 64; do
    ABM_BENCH_DIRECT_MIN=$v python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
    print("pairs 2x150, direct from %-4s entries rep %s  %.3f M reads/s  %.1f ms/step  alone %s" % (f, rep, d["value"] / 1e6, d["ms_per_step"], (d.get("phase_stamps") or {}).get("kernel_ms")))
except Exception as e:
    print("pairs", f, "rep", rep, "FAILED", e)
PY
  done
done
