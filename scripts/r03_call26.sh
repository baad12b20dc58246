# e2e against the slice size (what the run's last 20-45 ms are: the last slices' formatting)
set -u
mkdir -p gpurun_out
( ABM_BENCH_KEEP_FASTA=1 timeout 900 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> gpurun_out/r03_call26_prep.err )
WD=/dev/shm/abm_trace2
mkdir -p $WD
CLI=abismal_amd/abismal-amd
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads /tmp/abismal_bench/g3100.fa > /dev/null
for rep in 1 2 3; do
  for sl in 65536 32768 16384; do
    ABM_CLI_SLICE_READS=$sl $CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2>&1
    python3 -c "import json; t=json.load(open('$WD/t.json')); print('slice $sl rep $rep seconds', round(t['seconds'],4), 'M reads/s', round(t['reads']/t['seconds']/1e6,2))"
  done
done 2>&1 | tee gpurun_out/r03_exp_e2e_slice_size.log
rm -rf $WD
