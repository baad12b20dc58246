#!/bin/bash
# round 4, call 5: the whole GPU suite (new: pairs with long ends, explicit table rebuild, multi-GPU path on one device)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1
tail -15 gpurun_out/r04_gpu_tests.log
