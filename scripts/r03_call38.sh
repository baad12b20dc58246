# e2e against the first batch's size (the growth is x4 per batch) and the mappers per GPU
set -u
mkdir -p gpurun_out
( ABM_BENCH_KEEP_FASTA=1 timeout 900 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> gpurun_out/r03_call38_prep.err )
WD=/dev/shm/abm_trace2
mkdir -p $WD
CLI=abismal_amd/abismal-amd
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads /tmp/abismal_bench/g3100.fa > /dev/null
one() {
  local label="$1"; shift
  env "$@" > /dev/null 2>&1
  python3 -c "import json; t=json.load(open('$WD/t.json')); print('$label seconds', round(t['seconds'],4), 'M reads/s', round(t['reads']/t['seconds']/1e6,2), 'batches', t.get('batches_per_gpu'))"
}
for rep in 1 2 3; do
  one "first 1M (default) rep $rep" X=1 $CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
  one "first 512k rep $rep" ABM_CLI_FIRST_BATCH=524288 $CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
  one "first 2M rep $rep" ABM_CLI_FIRST_BATCH=2097152 $CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
  one "3 mappers rep $rep" X=1 $CLI map -mappers 3 -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
done 2>&1 | tee gpurun_out/r03_exp_e2e_first_batch.log
rm -rf $WD
