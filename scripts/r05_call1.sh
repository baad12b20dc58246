#!/bin/bash
# round 5, call 1: parity of the paired-end split (all launch forms), then the forms' rates at hg38 scale on one box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pe_split.py tests/test_gpu_pe_parity.py tests/test_gpu_se_set.py -x -q 2>&1 | tail -15 > gpurun_out/r05_call1_tests.log
cat gpurun_out/r05_call1_tests.log
FORMS="unsplit split split:1024 split:4096" REPS=2 scripts/r05_pe_forms.sh
