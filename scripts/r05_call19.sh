#!/bin/bash
# round 5, call 19: the threshold from which a read's wave takes priority (4096 candidates in one block of seed offsets;
# builds with 1024 and 256), on 150-base random-PBAT reads (4 M per step) and 100-base reads (10 M per step); then the seed
# kernel's list capacity (128 / 1024) for pairs on this build
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_heavy_threshold.log
: > $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in 1 2; do
  for v in tree hb1024 hb256; do
    unset ABISMAL_AMD_LIB
    [ "$v" != tree ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so
    python bench.py --mode random --read-len 150 --reads 4000000 --steps 6 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("150 bp -R, build %-7s rep %s  %.3f M reads/s  kernel %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch")))
except Exception as e:
    print("150 bp, build", f, "rep", rep, "FAILED", e)
PY
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("100 bp,    build %-7s rep %s  %.3f M reads/s  kernel %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch")))
except Exception as e:
    print("100 bp, build", f, "rep", rep, "FAILED", e)
PY
  done
done
unset ABISMAL_AMD_LIB
OUT=gpurun_out/r05_exp_pe_forms_final.log FORMS="split split:1024" REPS=2 scripts/r05_pe_forms.sh
