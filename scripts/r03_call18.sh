# results slice by slice: the new tests, the CLI goldens, the e2e leg, and the device + host timeline of one run
set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_sliced.py tests/test_gpu_cli_goldens.py tests/test_gpu_se_set.py tests/test_gpu_se_parity.py tests/test_gpu_edges_and_properties.py tests/test_gpu_targets_and_genome_option.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r03_call18_tests.log 2>&1
tail -12 gpurun_out/r03_call18_tests.log
( ABM_BENCH_KEEP_FASTA=1 timeout 1500 python bench.py --no-other-configs --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r03_call18_bench.json 2> gpurun_out/r03_call18_bench.err )
grep "e2e" gpurun_out/r03_call18_bench.err | cut -c1-700
python3 - <<'PY'
import json
s = open("gpurun_out/r03_call18_bench.json").read()
d = json.loads(s[s.find('{"metric"'):].splitlines()[0])
print("kernel ms", d["ms_per_step"], d["roofline"]["avg_kernel_ms"], "e2e", d["e2e"]["value"], d["e2e"]["seconds_of_each_run"], "sustained", d["e2e"]["sustained"]["value"], "parity", d["e2e"]["parity"]["identical"])
PY
bash scripts/r03_cli_gputrace.sh 2>&1 | tail -12
