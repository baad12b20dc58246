#!/bin/bash
# round 5: window records (DevIndex::wrec) -- parity of the record-fed filter on the GPU suite's single-end tests, then the
# same library with and without records on one box (10 M reads x 100 bp per step at 3.1 Gbp, oracle sample in rep 1)
set -u
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/r05_exp_window_records.log}
: > $OUT
export ABM_EXPERIMENTS=1
ABM_WINDOW_RECORDS=108 timeout 900 python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_scale_parity.py tests/test_gpu_seed_extension.py -x -q 2>&1 | tail -5 | tee -a $OUT
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
SAMPLE=${SAMPLE:-200000}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1   # builds the index once
for rep in 1 2; do
  for v in ${VARIANTS:-0 100}; do
    if [ $rep = 1 ]; then EXTRA="--cpu-sample $SAMPLE"; else EXTRA="--no-cpu-baseline --no-stage-split"; fi
    ABM_WINDOW_RECORDS=$v python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs $EXTRA 2>gpurun_out/r05_wrec_$v.err | tail -1 > gpurun_out/r05_wrec_${v}_$rep.json
    python - "$v" "$rep" gpurun_out/r05_wrec_${v}_$rep.json <<'PY' | tee -a $OUT
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print("records", v, "rep", rep, "FAILED", e); sys.exit(0)
r = d["roofline"]
print("records", v, "rep", rep, "ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], "value", d["value"])
if d.get("phase_shares_diagnostic"): print("   shares", d["phase_shares_diagnostic"])
if rep == "1":
    print("   work", d["work_per_read"])
    c = d.get("cpu_baseline") or {}
    print("   parity", {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
PY
  done
done
tail -3 gpurun_out/r05_wrec_100.err >> $OUT
