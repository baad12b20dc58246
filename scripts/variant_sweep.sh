#!/bin/bash
# experiment: rebuild the library with different compile-time knobs and time the bench
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-400} ABM_BENCH_READS=${ABM_BENCH_READS:-4000000}
for v in "$@"; do
  touch abismal_amd/csrc/abm_kernels.hip abismal_amd/csrc/abm_kernels_pe.hip
  make -C abismal_amd/csrc -j4 EXTRA="$v" 2>&1 | grep -E "error" 
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("variant", sys.argv[1], "| reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "step_ms", d["ms_per_step"])
PY
done
