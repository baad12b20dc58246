#!/bin/bash
# the CLI's first-batch size on a 10 M-read and a 40 M-read run (three repetitions each)
set -u
export ABM_BENCH_GENOME_MBP=3100
ABM_BENCH_KEEP_FASTA=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1
W=/dev/shm/abm_trace; mkdir -p $W
./abismal_amd/abismal-amd sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $W/reads /tmp/abismal_bench/g3100.fa > /dev/null
for fb in 1048576 2097152 3145728 999999999; do
  for rep in 1 2 3; do
    ABM_CLI_FIRST_BATCH=$fb ./abismal_amd/abismal-amd map -i /tmp/abismal_bench/g3100.idx -o $W/out.sam -timing $W/t.json $W/reads_1.fq 2> /dev/null
    echo "first batch $fb, 10M reads, rep $rep: $(python -c "import json; print(json.load(open('$W/t.json'))['seconds'])")"
  done
done
cat $W/reads_1.fq $W/reads_1.fq $W/reads_1.fq $W/reads_1.fq > $W/reads4.fq
for fb in 1048576 999999999; do
  for rep in 1 2 3; do
    ABM_CLI_FIRST_BATCH=$fb ./abismal_amd/abismal-amd map -i /tmp/abismal_bench/g3100.idx -o $W/out.sam -timing $W/t.json $W/reads4.fq 2> /dev/null
    echo "first batch $fb, 40M reads, rep $rep: $(python -c "import json; print(json.load(open('$W/t.json'))['seconds'])")"
  done
done
rm -rf $W
