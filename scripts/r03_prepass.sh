#!/bin/bash
# the seed pre-pass kernel at hg38 scale: on (default) vs off (ABM_NO_PREPASS), and the size from which ranges are
# narrowed directly; kernel times (pre-pass + mapping), probes per read, oracle parity
set -u
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
SAMPLE=${SAMPLE:-200000}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1   # builds the index once
show() {
python - "$1" "$2" <<'PY'
import json, sys
label, path = sys.argv[1:3]
try:
    d = json.load(open(path))
except Exception as e:
    print(label, "FAILED", e); sys.exit(0)
r = d["roofline"]
print(label, "ms/step", d["ms_per_step"], "kernels", r["avg_kernel_ms"], r.get("avg_kernel_ms_parts"), "frac", r["frac"], "probes", d["work_per_read"]["search_probes"])
if d.get("phase_shares_diagnostic"): print("   shares", d["phase_shares_diagnostic"]); print("   stages", [(s["stage"], s["time_share"], s["ms"]) for s in (r.get("stages") or [])])
c = d.get("cpu_baseline") or {}
if c: print("   parity", {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
PY
}
for rep in 1 2; do
  if [ $rep = 1 ]; then EXTRA="--cpu-sample $SAMPLE"; else EXTRA="--no-cpu-baseline --no-stage-split"; fi
  python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs $EXTRA 2>gpurun_out/r03_prepass_on.err | tail -1 > gpurun_out/r03_prepass_on_$rep.json
  show "pre-pass on  rep $rep" gpurun_out/r03_prepass_on_$rep.json
  ABM_EXPERIMENTS=1 ABM_NO_PREPASS=1 python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs $EXTRA 2>gpurun_out/r03_prepass_off.err | tail -1 > gpurun_out/r03_prepass_off_$rep.json
  show "pre-pass off rep $rep" gpurun_out/r03_prepass_off_$rep.json
done
for m in 0 256 4096; do
  ABM_EXPERIMENTS=1 ABM_DIRECT_MIN=$m python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > gpurun_out/r03_prepass_dm$m.json
  show "pre-pass on, direct from $m" gpurun_out/r03_prepass_dm$m.json
done
