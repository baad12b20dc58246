#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats of the bench runs of BASELINE configs 3 (pairs 2x150) and 5 (150 bp random PBAT)
set -u
export TMPDIR=/tmp ABM_BENCH_GENOME_MBP=3100
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_other
mkdir -p "$OUT"
cd /tmp
rm -rf /tmp/prof_c3 /tmp/prof_c5
(cd "$REPO" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline > "$OUT/config3_line_under_rocprof.log" 2>&1)
find /tmp/prof_c3 -name '*kernel_stats.csv' -exec cp {} "$OUT/config3_kernel_stats.csv" \;
(cd "$REPO" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 bench.py --mode random --read-len 150 --reads 4000000 --steps 3 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > "$OUT/config5_line_under_rocprof.log" 2>&1)
find /tmp/prof_c5 -name '*kernel_stats.csv' -exec cp {} "$OUT/config5_kernel_stats.csv" \;
ls -la "$OUT"
