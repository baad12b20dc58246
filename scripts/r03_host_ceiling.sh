#!/bin/bash
# what the host pipeline of `abismal-amd map` can carry on this box: -host-ceiling runs (no mapping call) over sinks,
# writer counts and thread counts, with the event trace of one of them.  Needs /tmp/abismal_bench/g3100.{idx,fa}.
set -u
WD=/dev/shm/abm_ceiling
mkdir -p $WD
CLI=abismal_amd/abismal-amd
IDX=/tmp/abismal_bench/g3100.idx
FA=/tmp/abismal_bench/g3100.fa
[ -f $WD/reads_1.fq ] || $CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads $FA > /dev/null
run() {  # label, env..., -- args
  local label=$1; shift
  env "$@" 2>/dev/null >/dev/null
}
one() {
  local label="$1"; shift
  "$@" > /dev/null 2> $WD/err.log
  python3 - "$label" $WD/t.json <<'PY'
import json, sys
t = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:46s} {t['reads'] / t['seconds'] / 1e6:6.2f} M reads/s  {t['seconds']:.3f} s  threads {t['host_threads']}  busy {t['busy_s']}")
PY
}
for th in 32 64 128 224; do
  one "ceiling tmpfs -t $th" $CLI map -host-ceiling -t $th -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
  one "ceiling /dev/null -t $th" $CLI map -host-ceiling -t $th -i $IDX -o /dev/null -timing $WD/t.json $WD/reads_1.fq
done
for w in ; do
  one "ceiling tmpfs -t 64 writers $w" env ABM_CLI_WRITERS=$w $CLI map -host-ceiling -t 64 -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
done
one "ceiling tmpfs -t 64 slices 16k" env ABM_CLI_SLICE_READS=16384 $CLI map -host-ceiling -t 64 -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
one "ceiling tmpfs -t 64 slices 256k" env ABM_CLI_SLICE_READS=262144 $CLI map -host-ceiling -t 64 -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
env ABM_CLI_TRACE=1 $CLI map -host-ceiling -t 64 -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2> gpurun_out/r03_ceiling_trace.log
one "real run tmpfs" $CLI map -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq
one "real run /dev/null" $CLI map -i $IDX -o /dev/null -timing $WD/t.json $WD/reads_1.fq
env ABM_CLI_TRACE=1 ABM_TRACE_HOST=1 $CLI map -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/reads_1.fq > /dev/null 2> gpurun_out/r03_real_trace.log
rm -rf $WD
