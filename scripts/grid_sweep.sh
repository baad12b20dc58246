#!/bin/bash
# experiment: kernel time vs waves in flight (ABM_GRID_WAVES overrides the launch grid)
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-400} ABM_BENCH_READS=${ABM_BENCH_READS:-1000000}
for g in "$@"; do
  ABM_EXPERIMENTS=1 ABM_GRID_WAVES=$g python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$g" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("grid", sys.argv[1], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"])
PY
done
