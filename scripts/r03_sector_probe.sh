#!/bin/bash
# random 32-byte gathers by load flavour / allocation type: rate, and the L2's memory-side requests per gather
set -u
export TMPDIR=/tmp
REPO=$(pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 tests/hip/sector_probe.hip -o /tmp/sector_probe || exit 1
/tmp/sector_probe
rm -rf /tmp/prof_sp
(cd /tmp && rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/prof_sp -- /tmp/sector_probe > /tmp/sp.log 2>&1)
CC=$(find /tmp/prof_sp -name '*counter_collection.csv' | head -1)
python3 - "$CC" <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for row in csv.DictReader(open(sys.argv[1])):
    key = (row["Dispatch_Id"], row["Kernel_Name"][:40])
    rows.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
gathers = {0: 256 * 20 * 64 * 75 * 4, 1: 256 * 20 * 64 * 300 * 4}
for i, ((d, k), c) in enumerate(rows.items()):
    n = gathers[i % 2]
    print(d, k, {x: f"{v:.4g}" for x, v in c.items()}, "RDREQ per gather %.2f, of which 32-byte %.2f" % (c.get("TCC_EA0_RDREQ_sum", 0) / n, c.get("TCC_EA0_RDREQ_32B_sum", 0) / n))
PY
