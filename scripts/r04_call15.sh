#!/bin/bash
# round 4, call 15: the GPU suite again (tier-2 workspaces by batch size, BAM writer, fast deflate), then the round's profile
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1
tail -4 gpurun_out/r04_gpu_tests.log
bash scripts/r04_profile.sh > gpurun_out/r04_profile.log 2>&1
tail -25 gpurun_out/r04_profile.log
