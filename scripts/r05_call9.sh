#!/bin/bash
# round 5, call 9: device SAM text against host formatting (new test), the pair tests on the final PeSet, the 8 M-pair
# end-to-end run with 16 contexts, then the round's profiles (scripts/r05_profile.sh)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cli_goldens.py tests/test_gpu_se_set.py tests/test_gpu_pe_split.py tests/test_gpu_pe_parity.py -x -q 2>&1 | tail -8 > gpurun_out/r05_call9_tests.log
cat gpurun_out/r05_call9_tests.log
export ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r05_call9_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa; CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_c9; mkdir -p $WD
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
for k in 1 2; do for f in 1 2 3 4; do cat $WD/p_$k.fq; done > $WD/x_$k.fq; done
for rep in 1 2 3; do
  $CLI map -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/x_1.fq $WD/x_2.fq 2> $WD/err.log || tail -3 $WD/err.log
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('8 M pairs end to end, %d contexts, rep $rep: %.2f M reads/s  %.3f s' % (t['mappers_per_gpu'], t['reads']/t['seconds']/1e6, t['seconds']))" | tee -a gpurun_out/r05_pe_e2e_final.log
done
rm -rf $WD
bash scripts/r05_profile.sh > gpurun_out/r05_profile.out 2>&1
tail -3 gpurun_out/r05_profile.out
