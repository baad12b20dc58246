#!/bin/bash
# round 5: (1) the seed-extension tables on the final kernels (direct narrowing in both): none / 4 + 2 / 6 + 3 / 7 + 4 letters;
# (2) single-end end to end with the SAM text written by the host (the default with 16 workers) and by the device
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_e2e_and_tables.log
: > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in 1 2; do
  for v in 7,4 6,3 4,2 0,0; do
    python bench.py --seed-ext $v --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > gpurun_out/r05_ext.json
    python - "$v" "$rep" gpurun_out/r05_ext.json <<'PY' | tee -a $OUT
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("single-end, tables", v, "rep", rep, "kernel avg", r["avg_kernel_ms"], "per launch", r.get("kernel_ms_per_launch"), "probes/read", d["work_per_read"]["search_probes"], d["seed_extension_tables"]["gb"], "GB")
except Exception as e:
    print("tables", v, "rep", rep, "FAILED", e)
PY
  done
done
for v in 6,3 0,0; do
  python bench.py --seed-ext $v --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline 2> /dev/null | tail -1 > gpurun_out/r05_ext.json
  python3 - "$v" gpurun_out/r05_ext.json <<'PY' | tee -a $OUT
import json, sys
f, path = sys.argv[1:3]
try:
    d = json.load(open(path))
    print("pairs 2x150, tables %-4s  %.3f M reads/s  %.1f ms/step" % (f, d["value"] / 1e6, d["ms_per_step"]))
except Exception as e:
    print("pairs, tables", f, "FAILED", e)
PY
done
for v in host device host device; do
  if [ $v = device ]; then export ABM_CLI_DEVICE_SAM=1; else unset ABM_CLI_DEVICE_SAM; fi
  python bench.py --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 > gpurun_out/r05_e2e.json
  python3 - "$v" gpurun_out/r05_e2e.json <<'PY' | tee -a $OUT
import json, sys
f, path = sys.argv[1:3]
try:
    d = json.load(open(path)); e = d["e2e"]
    print("SAM text by %-6s: kernel %.2f M reads/s; end to end 10 M reads %.2f M reads/s %s, 40 M reads %.2f M; cpu_s %s busy_s %s" % (f, d["value"] / 1e6, e["value"] / 1e6, e.get("seconds_of_each_run"), e["sustained"]["value"] / 1e6, e.get("cpu_s"), e.get("busy_s")))
except Exception as ex:
    print("e2e", f, "FAILED", ex)
PY
done
