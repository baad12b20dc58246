#!/bin/bash
# paired-end, same box: library builds in abismal_amd/_ab/libabismal_amd_<name>.so (ABISMAL_AMD_LIB selects the one loaded);
# VARIANTS="a b ..."; the first run builds the index
set -u
export ABM_BENCH_GENOME_MBP=3100
for rep in 1 2; do
  for v in ${VARIANTS}; do
    EXTRA="--no-cpu-baseline"; [ $rep = 1 ] && EXTRA="--cpu-sample 100000"
    ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so python bench.py --pe --reads 1000000 --read-len 150 --steps 12 --warmup 12 $EXTRA 2> gpurun_out/r03_pe_ab_$v.err | tail -1 > gpurun_out/r03_pe_ab_${v}_$rep.json
    python3 - "$v" "$rep" gpurun_out/r03_pe_ab_${v}_$rep.json <<'PY'
import json, sys
v, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print(v, "rep", rep, "FAILED", e); sys.exit(0)
r = d["roofline"]
print(v, "rep", rep, "reads/s", d["value"], "ms/step", d["ms_per_step"], "tier ms per launch", r.get("tier1_ms_per_launch"), r.get("tier2_ms_per_launch"))
c = d.get("cpu_baseline")
if c: print("   parity", {k: c[k] for k in c if "identical" in k or "vs_oracle" in k})
PY
  done
done
