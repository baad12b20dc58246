#!/bin/bash
# round 4, call 24: flakiness check -- the multi-GPU-path tests five times over, then the whole GPU suite once more
mkdir -p gpurun_out
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_sliced.py -x -q -m gpu 2>&1 | tail -1; done > gpurun_out/r04_flaky_check.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r04_flaky_check.log
cat gpurun_out/r04_flaky_check.log
