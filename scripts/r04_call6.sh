#!/bin/bash
# round 4, call 6: the default bench run as the driver starts it (validates every leg), then multi-GPU tests
mkdir -p gpurun_out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time
tail -c 3000 gpurun_out/r04_bench_default.err
cat gpurun_out/r04_bench_default.time
head -c 1500 gpurun_out/r04_bench_default.json
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3
