#!/bin/bash
# seed-extension tables at hg38 scale: kernel time, probes per read, phase shares and oracle parity for several
# table depths (EXTS="0,0 4,2 7,4"), same box, same library
set -u
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
SAMPLE=${SAMPLE:-200000}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs --seed-ext 0,0 > /dev/null 2>&1   # builds the index once
for rep in 1 2; do
  for x in ${EXTS:-0,0 4,2 7,4}; do
    if [ $rep = 1 ]; then EXTRA="--cpu-sample $SAMPLE"; else EXTRA="--no-cpu-baseline --no-stage-split"; fi
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --seed-ext $x $EXTRA 2>gpurun_out/r03_ext_$x.err | tail -1 > gpurun_out/r03_ext_${x}_$rep.json
    grep "index loaded" gpurun_out/r03_ext_$x.err
    python - "$x" "$rep" gpurun_out/r03_ext_${x}_$rep.json <<'PY'
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print(v, "rep", rep, "FAILED", e); sys.exit(0)
r = d["roofline"]
print("ext", v, "rep", rep, "ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], d.get("seed_extension_tables"))
if d.get("phase_shares_diagnostic"): print("   shares", d["phase_shares_diagnostic"])
if rep == "1":
    print("   work", d["work_per_read"])
    c = d.get("cpu_baseline") or {}
    print("   parity", {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
PY
  done
done
