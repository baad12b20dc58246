#!/bin/bash
# round 5, call 20: direct narrowing of big ranges (narrow_direct, the pair kernels' form) in the single-end kernel, now that
# the probes are the largest phase of a 150-base read (39 % of the wave time): 150-base random PBAT and 100-base reads,
# build "direct" (-DABM_SE_DIRECT_NARROWING=true) against the tree's
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
OUT=gpurun_out/r05_exp_se_direct_narrowing.log
: > $OUT
ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_direct.so timeout 900 python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_window_records.py tests/test_gpu_scale_parity.py -x -q 2>&1 | tail -3 | tee -a $OUT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e --no-other-configs > /dev/null 2>&1
for rep in 1 2; do
  for v in tree direct; do
    unset ABISMAL_AMD_LIB
    [ "$v" != tree ] && export ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so
    python bench.py --mode random --read-len 150 --reads 4000000 --steps 6 --warmup 1 --no-e2e --no-other-configs --cpu-sample 100000 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    c = d.get("cpu_baseline") or {}
    print("150 bp -R, build %-7s rep %s  %.3f M reads/s  kernel %s  probes/read %s  shares %s  parity %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch"), d["work_per_read"].get("search_probes"), d.get("phase_shares_diagnostic"), {k: c[k] for k in c if "identical" in k}))
except Exception as e:
    print("150 bp, build", f, "rep", rep, "FAILED", e)
PY
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> /dev/null | tail -1 > gpurun_out/r05_ab.json
    python3 - "$v" "$rep" gpurun_out/r05_ab.json <<'PY' | tee -a $OUT
import json, sys
f, rep, path = sys.argv[1:4]
try:
    d = json.load(open(path)); r = d["roofline"]
    print("100 bp,    build %-7s rep %s  %.3f M reads/s  kernel %s  probes/read %s" % (f, rep, d["value"] / 1e6, r.get("kernel_ms_per_launch"), d["work_per_read"].get("search_probes")))
except Exception as e:
    print("100 bp, build", f, "rep", rep, "FAILED", e)
PY
  done
done
