#!/bin/bash
# round 5, call 10: the device/host SAM-text test, the pair tests on the build whose tier-2 searches go through LDS samples,
# then that build against the one before it on one box (pair kernels, 1 M pairs 2x150 per step, 16 slots)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cli_goldens.py tests/test_gpu_pe_split.py tests/test_gpu_pe_parity.py tests/test_gpu_scale_parity.py tests/test_gpu_edges_and_properties.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call10_tests.log
cat gpurun_out/r05_call10_tests.log
OUT=gpurun_out/r05_exp_sampled_searches.log FORMS="split@unsampled split" REPS=3 scripts/r05_pe_forms.sh
