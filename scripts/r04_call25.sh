#!/bin/bash
# round 4, call 25: the bench line and the round's profile on the final tree
mkdir -p gpurun_out
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time
cat gpurun_out/r04_bench_default.time
grep "\[bench\]" gpurun_out/r04_bench_default.err | cut -c1-160 | tail -5
bash scripts/r04_profile.sh > gpurun_out/r04_profile.log 2>&1
tail -3 gpurun_out/r04_profile.log
