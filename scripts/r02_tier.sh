#!/bin/bash
# two-kernel single-end launch: parity tests, then launch time vs batch size for variants
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_edges_and_properties.py -x -q 2>&1 | tail -3
export ABM_BENCH_GENOME_MBP=3100
run() {  # label, env...
  local label=$1; shift
  for n in 1000000 4000000 10000000; do
    env "$@" ABM_BENCH_READS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > /tmp/line.json
    python - "$label" "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print(sys.argv[1], "reads/launch", sys.argv[2], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "ms/step", d["ms_per_step"], "status", d["kernel_status"])
PY
  done
}
run "single-tier        " ABM_SE_TWO_TIER=0
run "two-tier c16 b128k " ABM_SE_TWO_TIER=1
run "two-tier c14 b128k " ABM_SE_HEAVY_CLASS=14
run "two-tier c18 b128k " ABM_SE_HEAVY_CLASS=18
run "two-tier c16 b32k  " ABM_SE_BUDGET=32768
run "two-tier c16 g2048 " ABM_SE_HEAVY_GRID=2048
# full comparison against the oracle on a 1 M batch
ABM_BENCH_READS=1000000 python bench.py --steps 2 --warmup 1 --cpu-sample 1000000 --no-stage-split 2>/dev/null | tail -1 > gpurun_out/tier_parity_1m.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/tier_parity_1m.json')); print("1M parity", d["cpu_baseline"], "status", d["kernel_status"], "ms", d["roofline"]["avg_kernel_ms"])
PY
