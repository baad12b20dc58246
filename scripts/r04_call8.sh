#!/bin/bash
# round 4, call 8: paired-end results written into pinned host memory by the kernels: parity, then end to end
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_params.py tests/test_gpu_cli_goldens.py tests/test_gpu_multi.py "tests/test_gpu_edges_and_properties.py::test_pairs_with_a_long_end" tests/test_gpu_sliced.py -x -q -m gpu > gpurun_out/r04_pe_tests.log 2>&1
tail -5 gpurun_out/r04_pe_tests.log
export ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call8_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa
CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_pe; mkdir -p $WD
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
for e in 1 2; do cat $WD/p_$e.fq $WD/p_$e.fq $WD/p_$e.fq $WD/p_$e.fq > $WD/x_$e.fq; done
one() { # label args
  local label="$1"; shift
  $CLI map "$@" -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/x_1.fq $WD/x_2.fq 2> $WD/err.log || tail -3 $WD/err.log
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('%-40s %6.2f M reads/s  %.3f s  batches %s busy %s' % ('$label', t['reads']/t['seconds']/1e6, t['seconds'], t['batches_per_gpu'], {k: round(v,2) for k,v in t['busy_s'].items()}))"
}
{
one "default (8 mappers, 1 M pairs)"
one "default again"
one "4 mappers, 1 M" -mappers 4
one "12 mappers, 1 M" -mappers 12
one "16 mappers, 1 M" -mappers 16
one "12 mappers, 512 k" -mappers 12 -batch 524288
one "16 mappers, 512 k" -mappers 16 -batch 524288
one "3 mappers, 2 M (round 3)" -mappers 3 -batch 2097152
ABM_TRACE_HOST=1 $CLI map -i $IDX -o $WD/out.sam $WD/x_1.fq $WD/x_2.fq 2>&1 | head -150
} > gpurun_out/r04_pe_e2e_variants.log 2>&1
head -12 gpurun_out/r04_pe_e2e_variants.log
rm -rf $WD
