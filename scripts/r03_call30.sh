# direct narrowing threshold in the pair kernels: 256 / 128 / 64 entries (three repetitions: the rate is noisy)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
for rep in 1 2 3; do
  for v in direct256 direct128 direct64 nodirect; do
    ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so python bench.py --pe --reads 1000000 --read-len 150 --steps 12 --warmup 12 --no-cpu-baseline 2> gpurun_out/r03_pe_thr_$v.err | tail -1 > gpurun_out/r03_pe_thr_${v}_$rep.json
    python3 -c "
import json
d = json.load(open('gpurun_out/r03_pe_thr_${v}_$rep.json'))
print('$v rep $rep reads/s', d['value'], 'ms/step', d['ms_per_step'], 'probes/pair', d['roofline']['work_per_pair']['search_probes'])"
  done
done 2>&1 | tee gpurun_out/r03_exp_pe_direct_threshold.log
