#!/bin/bash
# paired-end at hg38 scale (BASELINE config 3 shape: 2 x 150 bp), with tier timings and phase shares
set -u
export ABM_BENCH_GENOME_MBP=3100
python bench.py --pe --reads 1000000 --read-len 150 --steps 6 --warmup 3 --phase-stamps --cpu-sample 400000 --no-e2e 2> gpurun_out/r02_pe.err | tail -1 > gpurun_out/r02_pe.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_pe.json'))
print("PE reads/s", d["value"], "ms/step", d["ms_per_step"]); print("roofline", d["roofline"]); print("cpu", d["cpu_baseline"]); print("diag", json.dumps(d["phase_stamps"], indent=1)[:3000])
PY
