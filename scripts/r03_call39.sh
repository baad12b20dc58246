# final tree: CLI-facing tests and the default bench line
set -u
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_cli_goldens.py tests/test_gpu_sliced.py tests/test_gpu_params.py -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r03_call39_tests.log 2>&1
tail -2 gpurun_out/r03_call39_tests.log
( timeout 1500 python bench.py > gpurun_out/r03_call39_bench.json 2> gpurun_out/r03_call39_bench.err )
python3 - <<'PY'
import json
s = open("gpurun_out/r03_call39_bench.json").read()
d = json.loads(s[s.find('{"metric"'):].splitlines()[0])
e = d["e2e"]
print("value", d["value"], "kernel", d["roofline"]["avg_kernel_ms"], "e2e", e["value"], e["seconds_of_each_run"], "sustained", e["sustained"]["value"], e["sustained"]["seconds"], "parity", e["parity"]["identical"], "cli", e["cli"])
for k, v in d["other_configs"].items():
    print(k, v.get("value"), v.get("error", "")[:300])
PY
