#!/bin/bash
# Round profile: (1) rocprofv3 kernel trace + stats of the default bench.py run, (2) a separate PMC
# pass for the HBM read requests of map_se_kernel.  Big trace files stay in /tmp on the GPU box;
# only the summaries are copied to gpurun_out/prof/ (then, by hand, into profiles/).
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_pmc
(cd "$REPO" && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 bench.py --cpu-sample 1000000 --no-e2e --no-other-configs > "$OUT/bench_line_under_rocprof.log" 2>&1)
grep '^{"metric"' "$OUT/bench_line_under_rocprof.log" > "$OUT/bench_line_under_rocprof.json"
find /tmp/prof_kt -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
KT=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
if [ -n "$KT" ]; then head -1 "$KT" > "$OUT/map_se_calls.csv"; grep map_se_kernel "$KT" >> "$OUT/map_se_calls.csv"; fi
(cd "$REPO" && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 bench.py --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split --steps 1 --warmup 0 > "$OUT/bench_line_under_pmc.log" 2>&1)
grep '^{"metric"' "$OUT/bench_line_under_pmc.log" > "$OUT/bench_line_under_pmc.json"
CC=$(find /tmp/prof_pmc -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then
  head -1 "$CC" > "$OUT/pmc_header.csv"
  grep map_se_kernel "$CC" > "$OUT/pmc_rdreq_map_se.csv"
fi
ls -la "$OUT"
