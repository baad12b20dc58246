#!/bin/bash
# round 4, call 23: phase shares of the pair kernels' diagnostic builds (current tree)
mkdir -p gpurun_out
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call23_index.err
timeout 900 python3 bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-cpu-baseline --no-e2e --no-other-configs --phase-stamps 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phase_stamps']
print('value %.2f M reads/s' % (d['value']/1e6)); print('kernel_ms', p['kernel_ms'])
for t,x in enumerate(p['tiers']): print('tier', t+1, x['counts'], x['share'])
for h in p['by_set_size']: print(h)
" > gpurun_out/r04_pe_phase_shares.log 2>&1
cat gpurun_out/r04_pe_phase_shares.log
