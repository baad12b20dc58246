#!/bin/bash
# round 5, call 12: pop_max with ballot-decided subtree walks -- the set checkers and the pair tests, then the pair
# kernels (alone times of the whole-pair launch = its costliest pair) with the seed kernel's lists capped at 128 / 1024 / 4096
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_se_set.py tests/test_gpu_pe_parity.py tests/test_gpu_pe_split.py tests/test_gpu_scale_parity.py tests/test_gpu_window_records.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call12_tests.log
cat gpurun_out/r05_call12_tests.log
OUT=gpurun_out/r05_exp_pe_forms_records.log FORMS="split split:1024 split:4096" REPS=2 scripts/r05_pe_forms.sh
