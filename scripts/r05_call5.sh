#!/bin/bash
# round 5, call 5: paired-end end to end after the hand-over buffers are reserved up front; the binned-filter probes
# (VERDICT r4 item 6); the seed-extension tables re-priced on the round's kernels (item 7): 7+4 / 6+3 / 5+2 letters,
# single-end (10 M x 100) and paired-end (1 M pairs 2x150), three repetitions each
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
export ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r05_call5_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa; CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_c5; mkdir -p $WD
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
for k in 1 2; do for f in 1 2 3 4; do cat $WD/p_$k.fq; done > $WD/x_$k.fq; done
{
for m in 16 8; do
  for rep in 1 2 3; do
    ABM_TRACE_HOST=1 $CLI map -mappers $m -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/x_1.fq $WD/x_2.fq 2> $WD/err.log || tail -3 $WD/err.log
    python3 -c "
import json; t=json.load(open('$WD/t.json')); print('mappers $m rep $rep: %.2f M reads/s  %.3f s  batches %s  regrown buffers %d' % (t['reads']/t['seconds']/1e6, t['seconds'], t.get('batches_per_gpu'), open('$WD/err.log').read().count('regrown')))"
  done
done
grep regrown $WD/err.log | sort | uniq -c | head
} 2>&1 | tee gpurun_out/r05_pe_e2e_after_reserve.log
rm -rf $WD
bash scripts/r05_binned_probes.sh
{
for rep in 1 2 3; do
  for ext in 7,4 6,3 5,2; do
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split --seed-ext $ext 2> gpurun_out/r05_call5_se.err | tail -1 > gpurun_out/r05_call5_se.json
    python3 -c "
import json; d=json.load(open('gpurun_out/r05_call5_se.json')); print('tables $ext rep $rep  SE %.3f M reads/s  kernel %.1f ms  tables %s' % (d['value']/1e6, d['roofline']['avg_kernel_ms'], d.get('seed_extension_tables')))"
    ABM_EXPERIMENTS=1 ABM_EXT_LETTERS=$ext python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-e2e --no-cpu-baseline 2> gpurun_out/r05_call5_pe.err | tail -1 > gpurun_out/r05_call5_pe.json
    python3 -c "
import json; d=json.load(open('gpurun_out/r05_call5_pe.json')); print('tables $ext rep $rep  PE %.3f M reads/s  %.1f ms/step  alone %s' % (d['value']/1e6, d['ms_per_step'], (d.get('phase_stamps') or {}).get('kernel_ms')))"
  done
done
} 2>&1 | tee gpurun_out/r05_exp_tables_repriced.log
