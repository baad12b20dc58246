#!/bin/bash
# GPU-side timeline of one `abismal-amd map` run (10 M reads): every kernel dispatch and memory copy with its start/end,
# to see what the device does between a batch's kernel handing out its last read and the next batch's kernel starting.
# Needs /tmp/abismal_bench/g3100.{idx,fa} (left by bench.py earlier in the same call).
set -u
export TMPDIR=/tmp
REPO=$(pwd)
WD=/dev/shm/abm_trace
mkdir -p $WD gpurun_out
CLI=$REPO/abismal_amd/abismal-amd
IDX=/tmp/abismal_bench/g3100.idx
FA=/tmp/abismal_bench/g3100.fa
[ -f $FA ] || python3 -c "import sys; sys.path.insert(0, '$REPO'); import torch, bench; bench.synth_genome_fasta('$FA', 3100, 1234, torch.device('cuda', 0))"
[ -f $WD/reads_1.fq ] || $CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads $FA > /dev/null
cd /tmp
rm -rf /tmp/prof_cli
ABM_CLI_TRACE=1 ABM_TRACE_HOST=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_cli -- $CLI map -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2> $REPO/gpurun_out/r03_gputrace_cli.err
cat $WD/t.json | head -c 1500; echo
KT=$(find /tmp/prof_cli -name '*kernel_trace.csv' | head -1)
MC=$(find /tmp/prof_cli -name '*memory_copy_trace.csv' | head -1)
python3 - "$KT" "$MC" > $REPO/gpurun_out/r03_gputrace_summary.log <<'PY'
import csv, sys
ev = []
for row in csv.DictReader(open(sys.argv[1])):
    ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), "K q%s %s" % (row.get("Queue_Id", "?"), row["Kernel_Name"][:60])))
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        for row in csv.DictReader(open(sys.argv[2])):
            ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), "C %s %s B" % (row.get("Direction", "?"), row.get("Bytes", "?"))))
    except Exception as e:
        print("no copy trace:", e)
ev.sort()
# the timeline from the first big mapping kernel on
big = [e for e in ev if "map_se_kernel" in e[2] and e[1] - e[0] > 20e6]
t0 = big[0][0] - 50_000_000 if big else ev[0][0]
for s, e, n in ev:
    if s < t0: continue
    print(f"{(s - t0) / 1e6:10.3f} {(e - t0) / 1e6:10.3f} {(e - s) / 1e6:9.3f} ms  {n}")
PY
tail -5 $REPO/gpurun_out/r03_gputrace_cli.err
grep -c . $REPO/gpurun_out/r03_gputrace_summary.log
grep -E "abm host" $REPO/gpurun_out/r03_gputrace_cli.err | tail -40
rm -rf $WD
