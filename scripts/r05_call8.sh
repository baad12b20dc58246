#!/bin/bash
# round 5, call 8: the whole GPU suite on the build with (i) the tier-2 heap's first levels in LDS and (ii) single-end SAM
# text written by the kernel; then same-box comparisons: the single-end kernel with the formatter compiled in against the
# build before it, `abismal-amd map` with device text against host formatting at -t 2 / 4 / 16, the pair kernels and the
# 8 M-pair end-to-end run with and without the LDS heap levels
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_call8_tests.log
cat gpurun_out/r05_call8_tests.log
export ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_KEEP_FASTA=1
OLD=$(pwd)/abismal_amd/_ab/libabismal_amd_nocache.so
{
for rep in 1 2; do
  for v in before after; do
    unset ABISMAL_AMD_LIB; [ $v = before ] && export ABISMAL_AMD_LIB=$OLD
    python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2> gpurun_out/r05_call8_se.err | tail -1 > gpurun_out/r05_call8_se.json
    python3 -c "
import json; d=json.load(open('gpurun_out/r05_call8_se.json')); print('SE kernel, formatter compiled in: $v  rep $rep  %.3f M reads/s  kernel %.1f ms' % (d['value']/1e6, d['roofline']['avg_kernel_ms']))"
  done
done
unset ABISMAL_AMD_LIB
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa; CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_c8; mkdir -p $WD
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/s $FA > /dev/null
for f in 1 2 3 4; do cat $WD/s_1.fq; done > $WD/s4.fq
for t in 16 4 2; do
  for how in device host; do
    for rep in 1 2; do
      if [ $how = host ]; then export ABM_CLI_HOST_FORMAT=1; else unset ABM_CLI_HOST_FORMAT; fi
      $CLI map -t $t -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/s4.fq 2> $WD/err.log || tail -3 $WD/err.log
      python3 -c "
import json; t=json.load(open('$WD/t.json')); print('40 M reads, -t $t, SAM text by the $how, rep $rep: %.2f M reads/s  %.3f s  cpu %s  busy %s  md5 ' % (t['reads']/t['seconds']/1e6, t['seconds'], t['cpu_s'], {k: round(v, 2) for k, v in t['busy_s'].items()}), end='')"
      grep -v '^@PG' $WD/out.sam | md5sum | cut -c1-16
    done
  done
done
unset ABM_CLI_HOST_FORMAT
rm -f $WD/s* $WD/out.sam
} 2>&1 | tee gpurun_out/r05_exp_device_sam_text.log
{
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
for k in 1 2; do for f in 1 2 3 4; do cat $WD/p_$k.fq; done > $WD/x_$k.fq; done
for rep in 1 2 3; do
  $CLI map -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/x_1.fq $WD/x_2.fq 2> $WD/err.log || tail -3 $WD/err.log
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('8 M pairs end to end, heap levels in LDS, 16 contexts, rep $rep: %.2f M reads/s  %.3f s' % (t['reads']/t['seconds']/1e6, t['seconds']))"
done
rm -rf $WD
} 2>&1 | tee gpurun_out/r05_exp_heap_levels_in_lds.log
OUT=gpurun_out/r05_exp_heap_levels_in_lds.log FORMS="split@nocache split" REPS=2 scripts/r05_pe_forms.sh
