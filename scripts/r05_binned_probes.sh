#!/bin/bash
# round 5: the two probes that decide whether a batch-wide, region-binned filter pass could beat the
# single-end kernel's one-line-per-window gathers: (a) random 32-byte gathers from L2-resident tables against the 4 GB
# table (requests/s), (b) the rate at which waves partition 8-byte records into 256 bins with LDS-staged full-line flushes
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/hip/sector_probe.hip -o /tmp/sector_probe || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/hip/partition_probe.hip -o /tmp/partition_probe || exit 1
{
echo "== (a) random 32-byte gathers, four in flight per lane, 20 waves per CU"
/tmp/sector_probe
echo "== (b) partition into 256 bins"
/tmp/partition_probe
} 2>&1 | tee gpurun_out/r05_exp_binned_filter_probes.log
