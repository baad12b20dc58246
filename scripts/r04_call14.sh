#!/bin/bash
# round 4, call 14: final validation -- the whole GPU suite, the smoke entry, the default bench run
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_tests.log 2>&1
tail -4 gpurun_out/r04_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; tail -2 gpurun_out/r04_smoke.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time
cat gpurun_out/r04_bench_default.time
grep "\[bench\]" gpurun_out/r04_bench_default.err | cut -c1-300 | tail -6
head -c 600 gpurun_out/r04_bench_default.json
