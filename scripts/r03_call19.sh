# results slice by slice, second round: write-through result stores, long last batch
set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_sliced.py tests/test_gpu_cli_goldens.py tests/test_gpu_se_set.py tests/test_gpu_se_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_targets_and_genome_option.py tests/test_gpu_pe_parity.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r03_call19_tests.log 2>&1
tail -6 gpurun_out/r03_call19_tests.log
( ABM_BENCH_KEEP_FASTA=1 timeout 1500 python bench.py --no-other-configs --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03_call19_bench.json 2> gpurun_out/r03_call19_bench.err )
python3 - <<'PY'
import json
s = open("gpurun_out/r03_call19_bench.json").read()
d = json.loads(s[s.find('{"metric"'):].splitlines()[0])
e = d["e2e"]
print("kernel ms", d["ms_per_step"], d["roofline"]["avg_kernel_ms"], "e2e", e["value"], e["seconds_of_each_run"], "sustained", e["sustained"]["value"], "parity", e["parity"], "cli", e["cli"])
PY
WD=/dev/shm/abm_trace2
mkdir -p $WD
CLI=abismal_amd/abismal-amd
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads /tmp/abismal_bench/g3100.fa > /dev/null
for rep in 1 2; do
  ABM_CLI_TRACE=1 ABM_TRACE_HOST=1 $CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2> gpurun_out/r03_call19_cli_trace_$rep.err
  python3 -c "import json; t=json.load(open('$WD/t.json')); print('run $rep seconds', t['seconds'], 'reads/s', t['reads']/t['seconds'])"
  grep "abm cli" gpurun_out/r03_call19_cli_trace_$rep.err | grep "batch" | cut -c1-80
  grep "abm cli" gpurun_out/r03_call19_cli_trace_$rep.err | grep "written" | awk 'NR%16==1' | cut -c1-60 | tr '\n' ';'; echo
  grep "abm cli" gpurun_out/r03_call19_cli_trace_$rep.err | grep "written" | tail -1
done
ABM_CLI_NO_STREAM=1 $CLI map -i /tmp/abismal_bench/g3100.idx -o $WD/out2.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2>&1
python3 -c "import json; t=json.load(open('$WD/t.json')); print('no-stream seconds', t['seconds'], 'reads/s', t['reads']/t['seconds'])"
cmp <(grep -v '^@PG' $WD/out.sam) <(grep -v '^@PG' $WD/out2.sam) && echo "streamed == whole-batch SAM"
rm -rf $WD
