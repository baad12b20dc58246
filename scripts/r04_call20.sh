#!/bin/bash
# round 4, call 20: the round's profile on the frozen tree (kernels of HEAD), the PE parity tests once more, the bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_se_parity.py tests/test_gpu_cli_goldens.py -x -q -m gpu > gpurun_out/r04_final_parity.log 2>&1
tail -3 gpurun_out/r04_final_parity.log
bash scripts/r04_profile.sh > gpurun_out/r04_profile.log 2>&1
tail -3 gpurun_out/r04_profile.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err ) 2> gpurun_out/r04_bench_default.time
cat gpurun_out/r04_bench_default.time
grep "\[bench\]" gpurun_out/r04_bench_default.err | cut -c1-160 | tail -5
