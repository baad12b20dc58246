# the default bench line with 16 paired-end slots
set -u
mkdir -p gpurun_out
( timeout 1500 python bench.py > gpurun_out/r03_call34_bench.json 2> gpurun_out/r03_call34_bench.err )
tail -4 gpurun_out/r03_call34_bench.err | cut -c1-300
python3 - <<'PY'
import json
s = open("gpurun_out/r03_call34_bench.json").read()
d = json.loads(s[s.find('{"metric"'):].splitlines()[0])
e = d["e2e"]
print("value", d["value"], "kernel", d["roofline"]["avg_kernel_ms"], "e2e", e["value"], e["seconds_of_each_run"], "sustained", e["sustained"]["value"], "parity", e["parity"]["identical"])
for k, v in d["other_configs"].items():
    print(k, v.get("value"), v.get("error", "")[:300], (v.get("cpu_baseline") or {}).get("pairs_hits_fallbacks_cigars_vs_oracle"), (v.get("config") or {}).get("streams"))
PY
