# where the end of a CLI run goes: joins and close traced
set -u
mkdir -p gpurun_out
( ABM_BENCH_KEEP_FASTA=1 timeout 900 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> gpurun_out/r03_call20_prep.err )
WD=/dev/shm/abm_trace2
mkdir -p $WD
CLI=abismal_amd/abismal-amd
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads /tmp/abismal_bench/g3100.fa > /dev/null
for rep in 1 2 3; do
  ABM_CLI_TRACE=1 $CLI map -v -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2> gpurun_out/r03_call20_cli_trace_$rep.err
  python3 -c "import json; t=json.load(open('$WD/t.json')); print('run $rep seconds', t['seconds'], 'reads/s', t['reads']/t['seconds'])"
  grep "abm cli" gpurun_out/r03_call20_cli_trace_$rep.err | grep -v "cut \|parsed\|formatted\|chunk" | grep "batch\|join\|closed" | cut -c1-90
  grep "abm cli" gpurun_out/r03_call20_cli_trace_$rep.err | grep "written" | tail -1
done
$CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2>&1
python3 -c "import json; t=json.load(open('$WD/t.json')); print('untraced seconds', t['seconds'], 'reads/s', t['reads']/t['seconds'])"
rm -rf $WD
