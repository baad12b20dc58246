# paired-end: how much of the timed region is the drain of the last steps (slots in flight: 16)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
for rep in 1 2; do
  for st in 16 32 64; do
    python bench.py --pe --reads 1000000 --read-len 150 --steps $st --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/pe.json
    python3 -c "
import json
d = json.load(open('/tmp/pe.json'))
print('steps $st rep $rep reads/s', d['value'], 'ms/step', d['ms_per_step'])"
  done
done 2>&1 | tee gpurun_out/r03_exp_pe_steps.log
