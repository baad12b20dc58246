#!/bin/bash
# in-block help: parity tests, then launch time vs batch size with one-wave workgroups (ABM_SE_HELP=0) and with help
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_scale_parity.py -x -q 2>&1 | tail -3
export ABM_BENCH_GENOME_MBP=3100
for h in 0 1; do
  for n in 1000000 4000000 10000000; do
    ABM_SE_HELP=$h ABM_BENCH_READS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split --no-e2e 2>/dev/null | tail -1 > /tmp/line.json
    python - "$h" "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("help", sys.argv[1], "reads/launch", sys.argv[2], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "status", d["kernel_status"])
PY
  done
done
ABM_BENCH_READS=1000000 python bench.py --steps 2 --warmup 1 --cpu-sample 1000000 --no-stage-split --no-e2e 2>/dev/null | tail -1 > gpurun_out/inblock_parity_1m.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/inblock_parity_1m.json')); print("1M parity", d["cpu_baseline"], "status", d["kernel_status"], "ms", d["roofline"]["avg_kernel_ms"])
PY
