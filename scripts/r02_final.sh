#!/bin/bash
# round-2 artefacts with the final build: profile round (kernel trace + PMC), the default bench line (with e2e), launch sizes
set -u
export ABM_BENCH_GENOME_MBP=3100
bash scripts/profile_round.sh > gpurun_out/r02_profile_round.log 2>&1
tail -3 gpurun_out/r02_profile_round.log
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -2 gpurun_out/r02_bench_default.err | cut -c1-1500
for n in 1000000 4000000; do python bench.py --reads $n --steps 4 --warmup 1 --no-e2e --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reads', d['config']['reads_per_step_per_gpu'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"; done
python scripts/r02_monster.py 2>&1 | grep -E "^(plain|timed)" | tee gpurun_out/r02_monster_final.log
