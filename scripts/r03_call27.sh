# final build: CLI-facing tests, e2e with 32 k slices, launch floors at 1 / 2 / 4 M reads
set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests/test_gpu_cli_goldens.py tests/test_gpu_sliced.py tests/test_gpu_params.py tests/test_gpu_targets_and_genome_option.py tests/test_gpu_edges_and_properties.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r03_call27_tests.log 2>&1
tail -3 gpurun_out/r03_call27_tests.log
( timeout 1500 python bench.py --no-other-configs > gpurun_out/r03_call27_bench.json 2> gpurun_out/r03_call27_bench.err )
python3 - <<'PY'
import json
s = open("gpurun_out/r03_call27_bench.json").read()
d = json.loads(s[s.find('{"metric"'):].splitlines()[0])
e = d["e2e"]
print("kernel ms", d["ms_per_step"], d["roofline"]["avg_kernel_ms"], "value", d["value"], "e2e", e["value"], e["seconds_of_each_run"], "sustained", e["sustained"]["value"], "parity", e["parity"]["identical"], "ceiling", e["host_ceiling_reads_per_s"], e["host_ceiling_reads_per_s_dev_null"])
PY
for n in 1000000 2000000 4000000; do
  ABM_BENCH_READS=$n python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split 2>/dev/null | tail -1 > /tmp/line.json
  python3 - "$n" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json')); print("reads/launch", sys.argv[1], "reads/s", d["value"], "kernel_ms", d["roofline"]["avg_kernel_ms"], "ms/step", d["ms_per_step"])
PY
done 2>&1 | tee gpurun_out/r03_launch_floors.log
