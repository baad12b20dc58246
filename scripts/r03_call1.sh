set -u
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03_call1_tests.log 2>&1
tail -5 gpurun_out/r03_call1_tests.log
VARIANTS="old new" bash scripts/r03_ab.sh 2>&1 | tee gpurun_out/r03_call1_ab.log
