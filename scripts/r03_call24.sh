# paired-end 2x100 (shorter reads: less LDS per wave, so the register budget decides the occupancy): old code at 4 waves per SIMD against the looped code at 4 and 3
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
for rep in 1 2; do
  for v in peold4 loop4 loop3; do
    ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so python bench.py --pe --reads 1000000 --read-len 100 --steps 12 --warmup 12 --no-cpu-baseline 2> gpurun_out/r03_pe100_$v.err | tail -1 > gpurun_out/r03_pe100_${v}_$rep.json
    python3 -c "
import json
d = json.load(open('gpurun_out/r03_pe100_${v}_$rep.json'))
print('$v rep $rep 2x100 reads/s', d['value'], 'ms/step', d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r03_exp_pe_2x100_variants.log
