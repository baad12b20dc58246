#!/bin/bash
# in-block help (four waves per workgroup) by launch size, now that the costliest reads' replay is cheap
set -u
export ABM_BENCH_GENOME_MBP=3100
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-stage-split --no-e2e > /dev/null 2>&1
for n in 1000000 2000000 4000000; do
  for h in 0 1; do
    ABM_SE_HELP=$h python bench.py --reads $n --steps 4 --warmup 1 --no-e2e --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('help $h reads', d['config']['reads_per_step_per_gpu'], 'ms/step', d['ms_per_step'], d.get('tail_help_per_launch'))"
  done
done
