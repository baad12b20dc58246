#!/bin/bash
set -u
export ABM_BENCH_GENOME_MBP=3100
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 400000 2> gpurun_out/r02_real2.err | tail -1 > gpurun_out/r02_real2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_real2.json"))
r = d["roofline"]
print("10M:", d["value"], "reads/s; ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], d["filter_genome"])
print("shares", d.get("phase_shares_diagnostic")); print("cpu", {k: v for k, v in d["cpu_baseline"].items() if "identical" in k or k == "value"})
PY
python bench.py --pe --reads 1000000 --read-len 150 --cpu-sample 300000 --no-e2e 2> gpurun_out/r02_pe3.err | tail -1 > gpurun_out/r02_pe3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_pe3.json"))
r = d["roofline"]
print("PE reads/s", d["value"], "ms/step", d["ms_per_step"], "tier ms", r.get("tier1_ms_per_launch"), r.get("tier2_ms_per_launch")); print("cpu", d["cpu_baseline"])
PY
VARIANTS="new" bash scripts/r02_tcc2.sh
