#!/bin/bash
# HBM read requests of map_se_kernel for the default bench workload (one launch), current build
set -u
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp && rm -rf /tmp/prof_pmc
(cd "$REPO" && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > /tmp/pmc.log 2>&1)
CC=$(find /tmp/prof_pmc -name '*counter_collection.csv' | head -1)
grep map_se_kernel "$CC" | awk -F, '{print $(NF-3), $(NF-2), ($NF-$(NF-1))/1e6 " ms"}'
