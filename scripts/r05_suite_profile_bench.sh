#!/bin/bash
# round 5, call 16: the whole GPU suite, the round profile (scripts/r05_profile.sh) and the default bench run on one box
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_gpu_tests_final.log
cat gpurun_out/r05_gpu_tests_final.log
scripts/r05_profile.sh > gpurun_out/r05_profile_run.log 2>&1
python bench.py > gpurun_out/r05_bench_default_final.json 2> gpurun_out/r05_bench_default_final.err
tail -c 600 gpurun_out/r05_bench_default_final.json
