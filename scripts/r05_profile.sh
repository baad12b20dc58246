#!/bin/bash
# Round 5 profile: (1) rocprofv3 kernel trace + stats of the bench run (single-end, config 2), (2) separate --pmc passes
# for the memory-side requests of the mapping kernels of configs 2 (SE 100 bp), 3 (PE 2x150) and 5 (SE 150 bp random
# PBAT).  Big trace files stay in /tmp on the GPU box; summaries go to gpurun_out/prof/, from where
# scripts/install_profiles.py copies them into profiles/ (and refuses if the build is not HEAD's).
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cp BUILD_ID "$OUT/build.txt" 2>/dev/null || echo unknown > "$OUT/build.txt"
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_pmc /tmp/prof_pe /tmp/prof_r150
(cd "$REPO" && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 bench.py --cpu-sample 1000000 --no-e2e --no-other-configs > "$OUT/bench_line_under_rocprof.log" 2>&1)
grep '^{"metric"' "$OUT/bench_line_under_rocprof.log" > "$OUT/bench_line_under_rocprof.json"
find /tmp/prof_kt -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
KT=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
if [ -n "$KT" ]; then head -1 "$KT" > "$OUT/map_se_calls.csv"; grep map_se_kernel "$KT" >> "$OUT/map_se_calls.csv"; fi
PMC="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
(cd "$REPO" && timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 bench.py --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split --steps 1 --warmup 0 > "$OUT/bench_line_under_pmc.log" 2>&1)
grep '^{"metric"' "$OUT/bench_line_under_pmc.log" > "$OUT/bench_line_under_pmc.json"
CC=$(find /tmp/prof_pmc -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then head -1 "$CC" > "$OUT/pmc_header.csv"; grep map_se_kernel "$CC" > "$OUT/pmc_rdreq_map_se.csv"; fi
# config 3: one step of 1 M pairs 2x150, both tiers (counters serialise the launches; the rate is bench.py's)
(cd "$REPO" && timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/prof_pe -- python3 bench.py --pe --reads 1000000 --read-len 150 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e > "$OUT/pe_line_under_pmc.log" 2>&1)
CC=$(find /tmp/prof_pe -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then grep map_pe_kernel "$CC" > "$OUT/pmc_rdreq_map_pe.csv"; fi
rm -rf /tmp/prof_pe
(cd "$REPO" && timeout 900 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/prof_pe -- python3 bench.py --pe --reads 1000000 --read-len 150 --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e > /tmp/pe_wr.log 2>&1)
CC=$(find /tmp/prof_pe -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then grep map_pe_kernel "$CC" > "$OUT/pmc_wrreq_map_pe.csv"; fi
# who asks the L2s: instruction fetch and vector requests of the single-end kernel (cf. profiles/r05_pe_pmc_who.log)
rm -rf /tmp/prof_who
(cd "$REPO" && timeout 900 rocprofv3 --pmc SQC_TC_INST_REQ TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/prof_who -- python3 bench.py --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split --steps 1 --warmup 0 > /tmp/who.log 2>&1)
CC=$(find /tmp/prof_who -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then grep map_se_kernel "$CC" > "$OUT/pmc_who_map_se.csv"; fi
# config 5: one step of 4 M reads x 150 bp, random PBAT
(cd "$REPO" && timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/prof_r150 -- python3 bench.py --mode random --read-len 150 --reads 4000000 --no-cpu-baseline --no-e2e --no-other-configs --no-stage-split --steps 1 --warmup 0 > "$OUT/r150_line_under_pmc.log" 2>&1)
CC=$(find /tmp/prof_r150 -name '*counter_collection.csv' | head -1)
if [ -n "$CC" ]; then grep map_se_kernel "$CC" > "$OUT/pmc_rdreq_map_se_r150.csv"; fi
ls -la "$OUT"
