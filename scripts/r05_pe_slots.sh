#!/bin/bash
# round 5: pairs 2x150, the bench loop with 16 / 24 / 32 slots (contexts) in flight, now that paired runs build no tables
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_pe_slots.log
: > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for v in 16 24 32 16 24; do
  python bench.py --pe --reads 1000000 --read-len 150 --streams $v --steps $((2 * v)) --warmup $v --no-e2e --no-cpu-baseline 2> gpurun_out/r05_pe_slots.err | tail -1 > gpurun_out/r05_ab.json
  python3 - "$v" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, path = sys.argv[1:3]
try:
    d = json.load(open(path))
    print("pairs 2x150, %s slots  %.3f M reads/s  %.1f ms/step" % (f, d["value"] / 1e6, d["ms_per_step"]))
except Exception as e:
    print("pairs,", f, "slots FAILED", e); print(open("gpurun_out/r05_pe_slots.err").read()[-400:])
PY
done
