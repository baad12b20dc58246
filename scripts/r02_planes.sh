#!/bin/bash
export ABM_EXPERIMENTS=1  # (the experiment variables below are honoured only with this set: abm_api.hip, experiment_env)
# the bit-plane genome for filter + narrowing: one copy vs two half-line-shifted copies; 10 M x 100 bp at hg38 scale
set -u
export ABM_BENCH_GENOME_MBP=3100
for c in 2 1; do
  ABM_PLANES_COPIES=$c python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline 2> gpurun_out/r02_planes_c$c.err | tail -1 > gpurun_out/r02_planes_c$c.json
  python - $c <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r02_planes_c{sys.argv[1]}.json"))
print("copies", sys.argv[1], "reads/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "shares", d.get("phase_shares_diagnostic"))
PY
done
