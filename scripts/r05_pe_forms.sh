#!/bin/bash
# paired-end launch forms on one box (round 5): tier 1 unsplit (rounds 1-4) against the seed / mate split at several
# seed-kernel list capacities.  FORMS="unsplit split split:1024 split@seed5 ..." (split:<seed_cap>, form@<library build>), REPS=n; the first run builds the index.
# Every line: reads/s of 16 overlapping slots, per-launch durations, pairs by route.
set -u
export ABM_BENCH_GENOME_MBP=${ABM_BENCH_GENOME_MBP:-3100}
export ABM_EXPERIMENTS=1
OUT=${OUT:-gpurun_out/r05_pe_forms.log}
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for f in ${FORMS:-unsplit split}; do
    unset ABM_PE_SPLIT ABM_PE_SCAP ABISMAL_AMD_LIB
    form=${f%%@*}
    # (form@build: the library build abismal_amd/_ab/libabismal_amd_<build>.so instead of the tree's)
    [ "$form" != "$f" ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_${f#*@}.so
    case $form in
      unsplit) export ABM_PE_SPLIT=0 ;;
      split) ;;
      split:*) export ABM_PE_SCAP=${form#split:} ;;
    esac
    python bench.py --pe --reads 1000000 --read-len 150 --steps ${STEPS:-16} --warmup 16 --no-e2e --no-cpu-baseline 2> gpurun_out/r05_pe_forms_$f.err | tail -1 > gpurun_out/r05_pe_forms_${f}_$rep.json
    python3 - "$f" "$rep" gpurun_out/r05_pe_forms_${f}_$rep.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path))
except Exception as e:
    print(f, "rep", rep, "FAILED", e); sys.exit(0)
r = d["roofline"]
print("%-12s rep %s  %.3f M reads/s  %.1f ms/step  per launch %s  routes %s  alone %s" % (
    f, rep, d["value"] / 1e6, d["ms_per_step"], r.get("ms_per_launch"), r.get("pairs_by_route"), (d.get("phase_stamps") or {}).get("kernel_ms")))
PY
  done
done
