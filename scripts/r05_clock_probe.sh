#!/bin/bash
# round 5: does the device's clock explain the spread of the record-fed kernel's launch times?  rocm-smi sampled every
# 0.25 s beside twelve launches with and without window records
set -u
mkdir -p gpurun_out
OUT=${OUT:-gpurun_out/r05_exp_clock_probe.log}
: > $OUT
export ABM_EXPERIMENTS=1
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for v in ${VARIANTS:-100 0}; do
  ( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Graphics Package Power|junction" | tr '\n' ' ' ; echo; sleep 0.25; done ) > gpurun_out/r05_smi_$v.log &
  SMI=$!
  ABM_WINDOW_RECORDS=$v python bench.py --steps 12 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > gpurun_out/r05_clock.json
  kill $SMI
  python - "$v" gpurun_out/r05_clock.json gpurun_out/r05_smi_$v.log <<'PY' | tee -a $OUT
import json, sys, re
v, path, smi = sys.argv[1:4]
d = json.load(open(path)); r = d["roofline"]
print("records", v, "kernel avg", r["avg_kernel_ms"], "per launch", r.get("kernel_ms_per_launch"))
rows = [l for l in open(smi) if "sclk" in l]
clk = [int(m.group(1)) for l in rows for m in [re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)] if m]
pw = [float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\): ([0-9.]+)", l)] if m]
print("   sclk samples (MHz):", clk)
print("   power samples (W):", pw)
PY
done
