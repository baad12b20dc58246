set -u
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_scale_parity.py tests/test_gpu_params.py tests/test_gpu_edges_and_properties.py tests/test_gpu_cli_goldens.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call11_tests.log 2>&1
tail -3 gpurun_out/r03_call11_tests.log
( ABM_EXPERIMENTS=1 ABM_DIRECT_MIN=64 timeout 1800 python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_scale_parity.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call11_tests_direct64.log 2>&1
tail -3 gpurun_out/r03_call11_tests_direct64.log
bash scripts/r03_prepass.sh 2>&1 | tee gpurun_out/r03_prologue.log
