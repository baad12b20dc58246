#!/bin/bash
# round 4, call 18: final validation of the final tree -- the GPU suite, the smoke entry, the default bench run, the round's profile
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1
tail -4 gpurun_out/r04_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; tail -1 gpurun_out/r04_smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time
cat gpurun_out/r04_bench_default.time
grep "\[bench\]" gpurun_out/r04_bench_default.err | cut -c1-200 | tail -7
bash scripts/r04_profile.sh > gpurun_out/r04_profile.log 2>&1
tail -3 gpurun_out/r04_profile.log
