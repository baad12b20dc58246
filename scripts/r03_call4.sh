set -u
mkdir -p gpurun_out
g++ -O2 -pthread scripts/r03_write_probe.cpp -o /tmp/write_probe && /tmp/write_probe /dev/shm/wp.bin 2>&1 | tee gpurun_out/r03_write_probe.log
( timeout 900 python scripts/r03_debug_pe.py 2>&1 | tail -30 ) | tee gpurun_out/r03_debug_pe_new.log
( ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_old.so timeout 900 python scripts/r03_debug_pe.py 2>&1 | tail -10 ) | tee gpurun_out/r03_debug_pe_old.log
export ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_KEEP_FASTA=1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs 2> gpurun_out/r03_call4_bench.err | tail -1 > gpurun_out/r03_call4_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r03_call4_bench.json')); print('bench (auto tables):', d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['seed_extension_tables'], d['index_upload_s'])"
bash scripts/r03_host_ceiling.sh 2>&1 | tee gpurun_out/r03_host_ceiling.log
bash scripts/r03_pmc.sh 2>&1 | tee gpurun_out/r03_pmc.log
