#!/bin/bash
# tail help: parity tests, then launch time vs batch size with help off/on, then a self-check build
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_se_parity.py -x -q 2>&1 | tail -3
export ABM_BENCH_GENOME_MBP=3100
for h in 0 1; do
  for n in 1000000 4000000 10000000; do
    ABM_SE_HELP=$h ABM_BENCH_READS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > /tmp/line.json
    python - "$h" "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("help", sys.argv[1], "reads/launch", sys.argv[2], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "status", d["kernel_status"], "cands/read", d["work_per_read"]["candidates"], "help", d.get("tail_help_per_launch"))
PY
  done
done
# full comparison against the oracle on a 1 M batch (help active through most of the launch)
ABM_BENCH_READS=1000000 python bench.py --steps 2 --warmup 1 --cpu-sample 1000000 --no-stage-split 2>/dev/null | tail -1 > gpurun_out/help_parity_1m.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/help_parity_1m.json')); print("1M parity", d["cpu_baseline"], "status", d["kernel_status"], "ms", d["roofline"]["avg_kernel_ms"])
PY
# self-check build: every handed-off result is recomputed by its owner
touch abismal_amd/csrc/abm_kernels.hip
make -C abismal_amd/csrc -j8 EXTRA="-DABM_HELP_SELFCHECK" 2>&1 | grep -E "error"
for n in 200000 1000000; do
  ABM_BENCH_READS=$n python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > /tmp/line.json
  python - "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("selfcheck reads/launch", sys.argv[1], "kernel_ms", d["roofline"]["avg_kernel_ms"], "status (16 = mismatch, 8 = timeout)", d["kernel_status"])
PY
done
