set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03_call2_tests.log 2>&1
tail -5 gpurun_out/r03_call2_tests.log
EXTS="0,0 4,2 6,3 7,4" bash scripts/r03_ext.sh 2>&1 | tee gpurun_out/r03_call2_ext.log
