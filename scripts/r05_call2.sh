#!/bin/bash
# round 5, call 2: the split's parity tests (all forms), then register budgets of the seed / mate kernels on one box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pe_split.py tests/test_gpu_pe_parity.py tests/test_gpu_se_set.py tests/test_gpu_scale_parity.py -x -q 2>&1 | tail -15 > gpurun_out/r05_call2_tests.log
cat gpurun_out/r05_call2_tests.log
OUT=gpurun_out/r05_pe_forms_budgets.log FORMS="split split@seed5 split@mate3 split:1024@mate3 split:4096@mate3" REPS=2 scripts/r05_pe_forms.sh
