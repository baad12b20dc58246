#!/bin/bash
# round 4, call 2: GPU tests of the multi-GPU code path on one device + the host-ceiling sweep (no GPU work)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_cli_goldens.py tests/test_gpu_sliced.py -x -q -m gpu > gpurun_out/r04_multi_tests.log 2>&1
tail -5 gpurun_out/r04_multi_tests.log
timeout 1500 python3 scripts/r04_host_ceiling.py --reps 3 > gpurun_out/r04_host_ceiling.log 2>&1
cat gpurun_out/r04_host_ceiling.log
rm -rf /dev/shm/abm_ceiling
