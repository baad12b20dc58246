#!/bin/bash
# round 5: abismal-amd map end to end on 40 M reads x 100 bp with 2 / 4 / 16 host workers, SAM text written by the host and by
# the device (an eighth / a quarter / all of the pod's 16 CPUs: one GPU's share of the host at 8 / 4 / 1 GPUs per node)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_KEEP_FASTA=1
OUT=gpurun_out/r05_exp_threads_sweep.log
: > $OUT
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2>&1
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa; CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_threads; rm -rf $WD; mkdir -p $WD
$CLI sim -single -seed 7 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/r $FA > /dev/null 2>&1
cat $WD/r_1.fq $WD/r_1.fq $WD/r_1.fq $WD/r_1.fq > $WD/r40.fq
for t in 2 4 16; do
  for who in device host; do
    if [ $who = device ]; then export ABM_CLI_DEVICE_SAM=1; else export ABM_CLI_DEVICE_SAM=0; fi
    $CLI map -t $t -i $IDX -o $WD/out.sam -s $WD/out.st -timing $WD/t.json $WD/r40.fq 2> $WD/err.log || tail -3 $WD/err.log
    python3 - "$t" "$who" $WD/t.json $WD/out.sam <<'PY' | tee -a $OUT
import json, sys, hashlib, subprocess
t, who, tj, sam = sys.argv[1:5]
d = json.load(open(tj))
md5 = subprocess.run("grep -v '^@PG' %s | md5sum | cut -c1-16" % sam, shell=True, capture_output=True, text=True).stdout.strip()
print("40 M reads, -t %-2s, SAM text by the %-6s: %.2f M reads/s  %.3f s  cpu %s  busy %s  md5 %s" % (t, d["sam_text_by"], d["reads"] / d["seconds"] / 1e6, d["seconds"], d["cpu_s"], {k: round(v, 2) for k, v in d["busy_s"].items()}, md5))
PY
  done
done
rm -rf $WD
