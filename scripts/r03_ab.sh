#!/bin/bash
# same-box A/B of library builds in abismal_amd/_ab/libabismal_amd_<name>.so on the default bench workload:
# un-profiled kernel time (two repetitions, interleaved), work per read, phase shares of the stamped kernel,
# and the field-by-field comparison with the oracle on a sample.  VARIANTS="old new ..." (the last one stays installed)
set -u
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
SAMPLE=${SAMPLE:-200000}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1   # builds the index once
for rep in 1 2; do
  for v in ${VARIANTS:-old new}; do
    cp abismal_amd/_ab/libabismal_amd_$v.so abismal_amd/libabismal_amd.so
    if [ $rep = 1 ]; then EXTRA="--cpu-sample $SAMPLE"; else EXTRA="--no-cpu-baseline --no-stage-split"; fi
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs $EXTRA 2>gpurun_out/r03_ab_$v.err | tail -1 > gpurun_out/r03_ab_${v}_$rep.json
    python - "$v" "$rep" gpurun_out/r03_ab_${v}_$rep.json <<'PY'
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print(v, "rep", rep, "FAILED", e); sys.exit(0)
r = d["roofline"]
print(v, "rep", rep, "ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"])
if d.get("phase_shares_diagnostic"): print("   shares", d["phase_shares_diagnostic"])
if rep == "1":
    print("   work", d["work_per_read"])
    c = d.get("cpu_baseline") or {}
    print("   parity", {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
PY
  done
done
