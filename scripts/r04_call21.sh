#!/bin/bash
# round 4, call 21: pair kernels with less LDS per wave (tier-1 lists of 64 entries, position cache of 128): more waves per CU
mkdir -p gpurun_out
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call21_index.err
{
for rep in 1 2 3; do
  for v in base c64p8 c128p7 c64p7; do
    cp abismal_amd/_ab/libabismal_amd_$v.so abismal_amd/libabismal_amd.so
    timeout 600 python3 bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --no-cpu-baseline --no-e2e --no-other-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v rep $rep: %.2f M reads/s  tier1 %.1f ms  tier2 %.1f ms per launch (overlapped)  status %s' % (d['value']/1e6, r['tier1_ms_per_launch'], r['tier2_ms_per_launch'], d['kernel_status']))"
  done
done
cp abismal_amd/_ab/libabismal_amd_base.so abismal_amd/libabismal_amd.so
} > gpurun_out/r04_exp_pe_lds.log 2>&1
cat gpurun_out/r04_exp_pe_lds.log
