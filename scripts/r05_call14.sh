#!/bin/bash
# round 5, call 14: the record-fed filter with one lane per candidate (reads up to 128 bases) -- single-end parity tests, then
# this build against the one before it on one box (10 M reads x 100 bp per step)
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_window_records.py tests/test_gpu_se_parity.py tests/test_gpu_scale_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_params.py tests/test_gpu_cli_goldens.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call14_tests.log
cat gpurun_out/r05_call14_tests.log
OUT=gpurun_out/r05_exp_one_lane_filter.log VARIANTS="prev tree" REPS=2 scripts/r05_lib_ab.sh
