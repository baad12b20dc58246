#!/bin/bash
# round 4, call 22: the multi-GPU code path at the bench's scale on the one device there is -- two and four replicas of the
# sharding on device 0 (`-devices 0,0[,0,0]`), one output file and part files, against the single-GPU run of the same input
mkdir -p gpurun_out
export ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call22_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa
CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_rep; mkdir -p $WD
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/s $FA > /dev/null
{
one() { # label, out files pattern, args
  local label="$1"; shift
  $CLI map -v "$@" -i $IDX -o $WD/out.sam -s $WD/out.st -timing $WD/t.json $WD/s_1.fq 2> $WD/err.log || tail -3 $WD/err.log
  local md5=$(cat $WD/out.sam $WD/out.sam.part* 2>/dev/null | grep -v '^@PG' | md5sum | cut -c1-32)
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('%-44s %6.2f M reads/s  %.3f s  batches per GPU %s  reads per GPU %s  body md5 $md5  stats md5 %s' % ('$label', t['reads']/t['seconds']/1e6, t['seconds'], t['batches_per_gpu'], t['reads_per_gpu'], __import__('hashlib').md5(open('$WD/out.st','rb').read()).hexdigest()[:12]))"
  grep -E "statistics summed|GPU [0-9]" $WD/err.log | sed 's/^/      /'
  rm -f $WD/out.sam $WD/out.sam.part*
}
one "one GPU (-gpus 1)" -gpus 1
one "two replicas on device 0, one file" -devices 0,0
one "two replicas on device 0, two parts" -devices 0,0 -out-parts 2
one "four replicas on device 0, four parts" -devices 0,0,0,0 -out-parts 4
one "one GPU again" -gpus 1
} > gpurun_out/r04_replicas_at_scale.log 2>&1
cat gpurun_out/r04_replicas_at_scale.log
rm -rf $WD
