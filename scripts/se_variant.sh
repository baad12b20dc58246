#!/bin/bash
# experiment: single-end kernel with a different waves-per-SIMD bound (registers per lane) and grid
for v in "$@"; do
  IFS=: read w g r <<< "$v"; r=${r:-8}
  touch abismal_amd/csrc/abm_kernels.hip
  make -C abismal_amd/csrc -j8 EXTRA="-DABM_SE_WAVES_PER_SIMD=$w -DABM_COOP_ROUNDS=$r" 2>&1 | grep -E "error"
  ABM_EXPERIMENTS=1 ABM_GRID_WAVES=$g ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_READS=10000000 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("waves/SIMD:grid:rounds", sys.argv[1], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"])
PY
done
