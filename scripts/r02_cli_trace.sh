#!/bin/bash
# timeline of the CLI on a 10 M-read FASTQ (tmpfs): pipeline events + the C ABI's host sections; batch-size variants;
# a 40 M-read input (4 copies) for the sustained rate
set -u
export ABM_BENCH_GENOME_MBP=3100
ABM_BENCH_KEEP_FASTA=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1   # builds the index, keeps the FASTA
W=/dev/shm/abm_trace; mkdir -p $W
./abismal_amd/abismal-amd sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $W/reads /tmp/abismal_bench/g3100.fa > /dev/null
for b in 16777216 5000000 3400000; do
  for rep in 1 2; do
    ABM_CLI_TRACE=1 ./abismal_amd/abismal-amd map -v -batch $b -i /tmp/abismal_bench/g3100.idx -o $W/out.sam -timing $W/t.json $W/reads_1.fq 2> gpurun_out/cli_trace_b${b}_$rep.log
    echo "batch $b rep $rep: $(cat $W/t.json)"
  done
  grep -E "batch (formed|ready|mapped)|abismal-amd\]" gpurun_out/cli_trace_b${b}_2.log | head -20
done
cat $W/reads_1.fq $W/reads_1.fq $W/reads_1.fq $W/reads_1.fq > $W/reads4.fq; rm $W/reads_1.fq
for b in 16777216 8388608; do
  ABM_CLI_TRACE=1 ./abismal_amd/abismal-amd map -v -batch $b -i /tmp/abismal_bench/g3100.idx -o $W/out.sam -timing $W/t.json $W/reads4.fq 2> gpurun_out/cli_trace4_b${b}.log
  echo "40M reads, batch $b: $(cat $W/t.json)"
  grep -E "batch (formed|ready|mapped)|abismal-amd\]" gpurun_out/cli_trace4_b${b}.log | head -20
done
rm -rf $W
