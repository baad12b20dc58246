# PCIe-inclusive rate of abm_map_se_batch by itself (host buffers in, results out; no parsing or formatting)
set -u
mkdir -p gpurun_out
( timeout 900 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> gpurun_out/r03_call40_prep.err )
ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_READS=10000000 python scripts/host_rate.py 2>&1 | tail -4 | tee gpurun_out/r03_host_rate.log
