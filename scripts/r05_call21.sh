#!/bin/bash
# round 5, call 21: the smallest range the single-end kernel narrows directly (64 / 128 / 256 / 512 entries)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_se_direct_threshold.log
: > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in 1 2; do
  for v in 128 64 256 512; do
    ABM_BENCH_DIRECT_MIN=$v python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("100 bp,    direct from %-4s entries rep %s  %.3f M reads/s  kernel %s  probes/read %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch"), d["work_per_read"].get("search_probes")))
except Exception as e:
    print("100 bp", f, "rep", rep, "FAILED", e)
PY
    ABM_BENCH_DIRECT_MIN=$v python bench.py --mode random --read-len 150 --reads 4000000 --steps 6 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("150 bp -R, direct from %-4s entries rep %s  %.3f M reads/s  kernel %s  probes/read %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch"), d["work_per_read"].get("search_probes")))
except Exception as e:
    print("150 bp", f, "rep", rep, "FAILED", e)
PY
  done
done
