#!/bin/bash
# round-2 baseline diagnostics: phase shares at 10 M reads, launch time vs batch size, per-read time distribution
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
python bench.py --steps 3 --warmup 1 --phase-stamps --no-cpu-baseline 2> gpurun_out/diag_bench.err | tail -1 > gpurun_out/diag_bench10m.json
bash scripts/batch_sweep.sh 1000000 4000000 > gpurun_out/diag_sweep.txt 2>&1
ABM_BENCH_READS=4000000 python scripts/tail_probe.py > gpurun_out/diag_tail4m.txt 2>&1
cat gpurun_out/diag_sweep.txt gpurun_out/diag_tail4m.txt
