#!/bin/bash
# experiment: mapping-kernel time vs reads per launch (fixed genome)
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
for n in "$@"; do
  ABM_BENCH_READS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("reads/launch", sys.argv[1], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "ms/step", d["ms_per_step"])
PY
done
