set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03_call3_tests.log 2>&1
tail -5 gpurun_out/r03_call3_tests.log
export ABM_BENCH_GENOME_MBP=3100
# the default line (auto table depth) incl. e2e, host ceiling
python bench.py --steps 5 --warmup 2 2> gpurun_out/r03_call3_default.err | tail -1 > gpurun_out/r03_call3_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_call3_default.json"))
r = d["roofline"]
print("default: value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], d.get("seed_extension_tables"), "upload_s", d.get("index_upload_s"))
print("  shares", d.get("phase_shares_diagnostic")); print("  work", d["work_per_read"])
c = d.get("cpu_baseline") or {}
print("  parity", {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
e = d.get("e2e") or {}
print("  e2e", {k: e.get(k) for k in ("value", "seconds_of_each_run", "sustained", "busy_s", "cli", "host_ceiling", "parity", "error")})
PY
EXTS="6,3 7,4" SAMPLE=100000 bash scripts/r03_ext.sh 2>&1 | tee gpurun_out/r03_call3_ext.log
# config 5: 150 bp random PBAT
python bench.py --mode random --read-len 150 --reads 5000000 --steps 3 --warmup 1 --cpu-sample 300000 --no-e2e --no-other-configs 2> gpurun_out/r03_call3_cfg5.err | tail -1 > gpurun_out/r03_call3_cfg5.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_call3_cfg5.json"))
r = d["roofline"]
print("cfg5: value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"])
print("  shares", d.get("phase_shares_diagnostic")); print("  work", d["work_per_read"])
c = d.get("cpu_baseline") or {}
print("  cpu", c.get("value"), {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
PY
