#!/bin/bash
# round 5, call 11: the whole GPU suite on the build with window records (record-fed kernels by default in the suite),
# then the pair kernels with and without records on one box (1 M pairs 2x150 per step, 16 slots)
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_call11_tests.log
cat gpurun_out/r05_call11_tests.log
export ABM_BENCH_GENOME_MBP=3100 ABM_EXPERIMENTS=1
OUT=gpurun_out/r05_exp_window_records_pe.log
: > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in 1 2 3; do
  for v in 150 0; do
    ABM_WINDOW_RECORDS=$v python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline 2> gpurun_out/r05_wrec_pe_$v.err | tail -1 > gpurun_out/r05_wrec_pe_${v}_$rep.json
    python3 - "$v" "$rep" gpurun_out/r05_wrec_pe_${v}_$rep.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print("records", f, "rep", rep, "FAILED", e); sys.exit(0)
r = d["roofline"]
print("records %-4s rep %s  %.3f M reads/s  %.1f ms/step  per launch %s  alone %s" % (
    f, rep, d["value"] / 1e6, d["ms_per_step"], r.get("ms_per_launch"), (d.get("phase_stamps") or {}).get("kernel_ms")))
PY
  done
done
tail -2 gpurun_out/r05_wrec_pe_150.err >> $OUT
