set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03_call10_tests.log 2>&1
tail -4 gpurun_out/r03_call10_tests.log
( ABM_EXPERIMENTS=1 ABM_DIRECT_MIN=64 timeout 1800 python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_scale_parity.py tests/test_gpu_params.py tests/test_gpu_edges_and_properties.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call10_tests_direct64.log 2>&1
tail -3 gpurun_out/r03_call10_tests_direct64.log
bash scripts/r03_prepass.sh 2>&1 | tee gpurun_out/r03_prepass.log
