# packed four-jobs-per-slot scoring: primitive check, parity suites, same-box A/B (never / from three jobs / always), PE rate
set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_se_set.py tests/test_gpu_se_parity.py tests/test_gpu_pe_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_cli_goldens.py tests/test_gpu_scale_parity.py tests/test_gpu_seed_extension.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call17_tests.log 2>&1
tail -3 gpurun_out/r03_call17_tests.log
VARIANTS="noquad quadall quad3" bash scripts/r03_ab.sh 2>&1 | tee gpurun_out/r03_exp_quad_scoring.log
( timeout 900 python bench.py --pe --reads 1000000 --read-len 150 --steps 12 --warmup 12 --cpu-sample 100000 --phase-stamps > gpurun_out/r03_call17_pe.json 2> gpurun_out/r03_call17_pe.err )
tail -c 1800 gpurun_out/r03_call17_pe.json
