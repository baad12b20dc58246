set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03_call5_tests.log 2>&1
tail -5 gpurun_out/r03_call5_tests.log
export ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_KEEP_FASTA=1
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
bash scripts/r03_host_ceiling.sh 2>&1 | tee gpurun_out/r03_host_ceiling2.log
