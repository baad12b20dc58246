#!/bin/bash
# round 4, call 3: event traces of virtual-GPU runs (where does a 40 M-read run spend its 0.6 s?) + the box's foreign load
mkdir -p gpurun_out
WD=/dev/shm/abm_ceiling
{ cat /proc/loadavg; which perf; top -bn1 | head -25; } > gpurun_out/r04_box_load.log 2>&1
python3 scripts/r04_host_ceiling.py --reps 1 --only "1 vGPU, /dev/null" > /dev/null 2>&1   # (makes the index and the 40 M-read FASTQ)
CLI=abismal_amd/abismal-amd
run() { # label args...
  local label=$1; shift
  ABM_CLI_TRACE=1 $CLI map "$@" -i $WD/tRex1.idx -timing $WD/t.json $WD/reads_1.fq 2> gpurun_out/r04_trace_$label.log
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('$label', round(t['reads']/t['seconds']/1e6,2), 'M reads/s', t['seconds'], 'prep', t['host_prepare_s'], t['busy_s'], t['cpu_s'])"
}
run null_t64 -virtual-gpus 8 -t 64 -o /dev/null
run null_t32 -virtual-gpus 8 -t 32 -o /dev/null
run parts8_t64 -virtual-gpus 8 -out-parts 8 -t 64 -o $WD/out.sam
rm -f $WD/out.sam*
run one_t32 -virtual-gpus 1 -t 32 -o $WD/out.sam
rm -f $WD/out.sam*
run parts8_t128 -virtual-gpus 8 -out-parts 8 -t 128 -o $WD/out.sam
cat /proc/loadavg
rm -rf $WD
