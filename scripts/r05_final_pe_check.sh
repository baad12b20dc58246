#!/bin/bash
# round 5: the paired-end defaults (24 contexts / slots, no tables) end to end: CLI goldens, the replicas test, config 3's bench line
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_cli_goldens.py tests/test_gpu_multi.py tests/test_gpu_params.py -x -q 2>&1 | tail -4 > gpurun_out/r05_final_pe_tests.log
cat gpurun_out/r05_final_pe_tests.log
export ABM_BENCH_GENOME_MBP=3100
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
python bench.py --pe --reads 1000000 --read-len 150 --cpu-sample 200000 > gpurun_out/r05_config3_final.json 2> gpurun_out/r05_config3_final.err
tail -c 1500 gpurun_out/r05_config3_final.json
