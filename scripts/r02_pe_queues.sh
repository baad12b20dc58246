#!/bin/bash
# paired-end at hg38 scale: how many batches in flight (streams = contexts, each with its own HIP stream) and how
# many hardware queues the runtime may use (GPU_MAX_HW_QUEUES, default 4) it takes to hide tier 2's single-wave tail
set -u
export ABM_BENCH_GENOME_MBP=3100
for q in ${QUEUES:-16 32}; do
  for s in ${STREAMS:-8 12 16}; do
    GPU_MAX_HW_QUEUES=$q python bench.py --pe --reads 1000000 --read-len 150 --steps $((s + 4)) --warmup $s --streams $s --no-cpu-baseline --no-e2e 2> gpurun_out/r02_pe_q${q}_s${s}.err | tail -1 > gpurun_out/r02_pe_q${q}_s${s}.json
    python - $q $s <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r02_pe_q{sys.argv[1]}_s{sys.argv[2]}.json"))
r = d["roofline"]
print("hwq", sys.argv[1], "streams", sys.argv[2], "reads/s", d["value"], "ms/step", d["ms_per_step"], "tier ms", r.get("tier1_ms_per_launch"), r.get("tier2_ms_per_launch"))
PY
  done
done
