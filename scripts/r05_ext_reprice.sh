#!/bin/bash
# round 5: the seed-extension tables re-priced on the record-fed kernels (7 + 4 letters = 90 GB, 6 + 3 = 36 GB, 5 + 2, 4 + 2,
# none): single-end 10 M x 100 bp per step, per-launch kernel times; pairs 1 M x 2x150 per step
set -u
mkdir -p gpurun_out
OUT=gpurun_out/r05_exp_tables_repriced_records.log
: > $OUT
export ABM_BENCH_GENOME_MBP=3100
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in 1 2; do
  for v in 7,4 6,3 5,2 4,2 0,0; do
    python bench.py --seed-ext $v --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > gpurun_out/r05_ext.json
    python - "$v" "$rep" gpurun_out/r05_ext.json <<'PY' | tee -a $OUT
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("single-end, tables", v, "rep", rep, "kernel avg", r["avg_kernel_ms"], "per launch", r.get("kernel_ms_per_launch"), d["seed_extension_tables"])
except Exception as e:
    print("tables", v, "rep", rep, "FAILED", e)
PY
  done
done
for rep in 1 2; do
  for v in 7,4 6,3 4,2; do
    python bench.py --seed-ext $v --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r05_ext.json
    python3 - "$v" "$rep" gpurun_out/r05_ext.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
    print("pairs 2x150, tables %-4s rep %s  %.3f M reads/s  %.1f ms/step" % (f, rep, d["value"] / 1e6, d["ms_per_step"]))
except Exception as e:
    print("pairs, tables", f, "rep", rep, "FAILED", e)
PY
  done
done
