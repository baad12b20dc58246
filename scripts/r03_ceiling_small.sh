#!/bin/bash
# host-ceiling runs on a small index (tRex1): what limits the pipeline as host threads are added -- placement of the
# threads (taskset), memory policy (numactl), slice size
set -u
WD=/dev/shm/abm_ceil_small
mkdir -p $WD
CLI=abismal_amd/abismal-amd
FA=tests/golden/tRex1.fa
lscpu | egrep "Model name|Socket|Core|Thread|NUMA|^CPU\(s\)" ; (numactl -H 2>/dev/null | head -12) || echo "no numactl"
free -g | head -2
$CLI idx $FA $WD/t.idx > /dev/null 2>&1
[ -f $WD/reads_1.fq ] || $CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/reads $FA > /dev/null
one() {
  local label="$1"; shift
  "$@" > /dev/null 2> $WD/err.log || tail -2 $WD/err.log
  python3 - "$label" $WD/t.json <<'PY'
import json, sys
t = json.load(open(sys.argv[2]))
print(f"{sys.argv[1]:52s} {t['reads'] / t['seconds'] / 1e6:6.2f} M reads/s  {t['seconds']:.3f} s  busy {t['busy_s']}")
PY
}
ARGS="map -host-ceiling -seed-ext 0,0 -i $WD/t.idx -timing $WD/t.json $WD/reads_1.fq"
for th in 16 32 64 128; do
  one "/dev/null -t $th" $CLI $ARGS -t $th -o /dev/null
done
for th in 64 128; do
  one "/dev/null -t $th taskset 0-63" taskset -c 0-63 $CLI $ARGS -t $th -o /dev/null
  one "/dev/null -t $th taskset 0-127" taskset -c 0-127 $CLI $ARGS -t $th -o /dev/null
  one "/dev/null -t $th numactl interleave" numactl --interleave=all $CLI $ARGS -t $th -o /dev/null
  one "/dev/null -t $th no prewarm" env ABM_CLI_NO_PREWARM=1 $CLI $ARGS -t $th -o /dev/null
done
one "tmpfs -t 32" $CLI $ARGS -t 32 -o $WD/out.sam
one "tmpfs -t 64 taskset 0-63" taskset -c 0-63 $CLI $ARGS -t 64 -o $WD/out.sam
rm -rf $WD
