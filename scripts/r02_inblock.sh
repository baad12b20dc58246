#!/bin/bash
# in-block help: launch time vs batch size: one-wave workgroups (ABM_SE_HELP=0) vs help (4 waves per workgroup)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
for h in 0 1; do
  for n in 500000 1000000 2000000 4000000 10000000; do
    ABM_SE_HELP=$h ABM_BENCH_READS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split --no-e2e 2>/dev/null | tail -1 > /tmp/line.json
    python - "$h" "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("  help", sys.argv[1], "reads/launch", sys.argv[2], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "status", d["kernel_status"], d.get("tail_help_per_launch"))
PY
  done
done
