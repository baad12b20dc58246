#!/bin/bash
# round 4, call 16: the host ceiling once more with the round's final binary (SAM lines by pointer writes, fast BGZF deflate)
mkdir -p gpurun_out
timeout 1500 python3 scripts/r04_host_ceiling.py --reps 5 --final --out gpurun_out/r04_host_ceiling_final.json > gpurun_out/r04_host_ceiling_final.log 2>&1
cat gpurun_out/r04_host_ceiling_final.log | cut -c1-260
rm -rf /dev/shm/abm_ceiling
