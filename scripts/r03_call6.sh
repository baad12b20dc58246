set -u
mkdir -p gpurun_out
( timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r03_call6_tests.log 2>&1
tail -6 gpurun_out/r03_call6_tests.log
export ABM_BENCH_GENOME_MBP=3100
t0=$(date +%s)
python bench.py 2> gpurun_out/r03_call6_default.err | tail -1 > gpurun_out/r03_call6_default.json
echo "default bench run took $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03_call6_default.json"))
r = d["roofline"]
print("default: value", d["value"], "ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], d.get("seed_extension_tables"))
print("  shares", d.get("phase_shares_diagnostic"))
c = d.get("cpu_baseline") or {}
print("  cpu", c.get("value"), {k: c[k] for k in c if "identical" in k or "long_cigars" in k})
e = d.get("e2e") or {}
print("  e2e", {k: e.get(k) for k in ("value", "seconds_of_each_run", "sustained", "busy_s", "cli", "index_load_s", "host_prepare_s", "parity", "error")})
for c in e.get("host_ceiling") or []: print("    ceiling", c)
for k, v in (d.get("other_configs") or {}).items():
    print(" ", k, {x: v.get(x) for x in ("value", "ms_per_step", "wall_s", "error")}, (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
    cb = v.get("cpu_baseline") or {}
    print("     parity", {x: cb[x] for x in cb if "identical" in x or "oracle" in x})
PY
tail -5 gpurun_out/r03_call6_default.err
