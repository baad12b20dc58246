#!/bin/bash
# round 4, call 26: the paired-end multi-GPU code path at scale on one device: 2 M pairs 2x150 through one "GPU" and through two replicas with two parts
mkdir -p gpurun_out
export ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call26_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa
CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_rep; mkdir -p $WD
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
{
one() {
  local label="$1"; shift
  $CLI map -v "$@" -i $IDX -o $WD/out.sam -s $WD/out.st -timing $WD/t.json $WD/p_1.fq $WD/p_2.fq 2> $WD/err.log || tail -3 $WD/err.log
  local md5=$(cat $WD/out.sam $WD/out.sam.part* 2>/dev/null | grep -v '^@PG' | md5sum | cut -c1-32)
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('%-44s %6.2f M reads/s  %.3f s  batches per GPU %s  pairs per GPU %s  body md5 $md5  stats md5 %s' % ('$label', t['reads']/t['seconds']/1e6, t['seconds'], t['batches_per_gpu'], t['reads_per_gpu'], __import__('hashlib').md5(open('$WD/out.st','rb').read()).hexdigest()[:12]))"
  rm -f $WD/out.sam $WD/out.sam.part*
}
one "pairs, one GPU (-gpus 1)" -gpus 1
one "pairs, two replicas on device 0, two parts" -devices 0,0 -out-parts 2
one "pairs, two replicas on device 0, one file" -devices 0,0
} > gpurun_out/r04_replicas_at_scale_pe.log 2>&1
cat gpurun_out/r04_replicas_at_scale_pe.log
rm -rf $WD
