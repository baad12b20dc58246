set -u
mkdir -p gpurun_out
( timeout 2700 python -m pytest tests/test_gpu_edges_and_properties.py tests/test_gpu_targets_and_genome_option.py tests/test_gpu_seed_extension.py tests/test_gpu_cli_goldens.py tests/test_gpu_scale_parity.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r03_call7_tests.log 2>&1
tail -6 gpurun_out/r03_call7_tests.log
bash scripts/r03_sector_probe.sh 2>&1 | tee gpurun_out/r03_sector_probe.log
bash scripts/r03_ceiling_small.sh 2>&1 | tee gpurun_out/r03_ceiling_small.log
