#!/bin/bash
# host-side timeline of the CLI on 40 M reads (the 10 M-read FASTQ four times over): pipeline events + C ABI sections
set -u
export ABM_BENCH_GENOME_MBP=3100
ABM_BENCH_KEEP_FASTA=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1
W=/dev/shm/abm_trace; mkdir -p $W
./abismal_amd/abismal-amd sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $W/reads /tmp/abismal_bench/g3100.fa > /dev/null
cat $W/reads_1.fq $W/reads_1.fq $W/reads_1.fq $W/reads_1.fq > $W/reads4.fq
for rep in 1 2; do
  ABM_TRACE_HOST=1 ABM_CLI_TRACE=1 ./abismal_amd/abismal-amd map -v -i /tmp/abismal_bench/g3100.idx -o $W/out.sam -timing $W/t.json $W/reads4.fq 2> gpurun_out/cli_trace5_$rep.log
  echo "40M reads rep $rep: $(cat $W/t.json)"
done
grep -E "batch (formed|ready|mapped)|abm host|abismal-amd\]" gpurun_out/cli_trace5_2.log | head -80
ABM_TRACE_HOST=1 ABM_CLI_TRACE=1 ./abismal_amd/abismal-amd map -v -i /tmp/abismal_bench/g3100.idx -o $W/out.sam -timing $W/t.json $W/reads_1.fq 2> gpurun_out/cli_trace5_10m.log
echo "10M reads: $(cat $W/t.json)"
grep -E "batch (formed|ready|mapped)|abm host|abismal-amd\]" gpurun_out/cli_trace5_10m.log | head -40
rm -rf $W
