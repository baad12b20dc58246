#!/bin/bash
# round 4, call 11: pairs predicted to outgrow tier 1's lists (by the weight class of their first seed buckets) sent to tier 2 at once
mkdir -p gpurun_out
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call11_index.err
{
for c in 0 26 24 22 20 18 16 0; do
  ABM_EXPERIMENTS=1 ABM_PE_BIG_CLASS=$c timeout 600 python3 bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-cpu-baseline --no-e2e --no-other-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('big_class $c: %.2f M reads/s  tier1 %.1f ms  tier2 %.1f ms per launch  tier-2 share of candidates %.3f  lines/pair accounted %s' % (d['value']/1e6, r['tier1_ms_per_launch'], r['tier2_ms_per_launch'], r['tier2_share_of_candidates'], r['lines_per_pair_by_source']['accounted']))"
done
} > gpurun_out/r04_pe_big_class.log 2>&1
cat gpurun_out/r04_pe_big_class.log
