# paired-end: tier-1 list capacity 64 against 128 once more, five repetitions (the difference was inside the noise)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
for rep in 1 2 3 4 5; do
  for v in cap128 cap64; do
    ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/pe.json
    python3 -c "
import json
d = json.load(open('/tmp/pe.json'))
print('$v rep $rep reads/s', d['value'], 'ms/step', d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r03_exp_pe_tier1_cap_64_vs_128.log
