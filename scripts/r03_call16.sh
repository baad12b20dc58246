# row-by-row traceback run: primitive check, parity subset, same-box A/B against the anti-diagonal run
set -u
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_se_set.py tests/test_gpu_se_parity.py tests/test_gpu_pe_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_cli_goldens.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call16_tests.log 2>&1
tail -3 gpurun_out/r03_call16_tests.log
VARIANTS="tbdiag tbrows" bash scripts/r03_ab.sh 2>&1 | tee gpurun_out/r03_exp_traceback_rows.log
