#!/bin/bash
# experiment: paired-end rate for (tier-1 list capacity, waves per SIMD) variants; rebuilds the library on the box
for v in "$@"; do
  cap=${v%%:*}; w=${v##*:}
  touch abismal_amd/csrc/*.hip
  make -C abismal_amd/csrc -j8 EXTRA="-DABM_PE_WAVES_PER_SIMD=$w -DABM_PE_TIER1_CAP=$cap" 2>&1 | grep -E "error" 
  python bench.py --pe --genome-mbp 1000 --reads 1000000 --read-len 150 --no-cpu-baseline --steps 4 --warmup 2 --phase-stamps 2>/dev/null | tail -1 > /tmp/pe.json
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/pe.json')); print("cap:waves", sys.argv[1], "reads/s", d["value"], "ms/step", d["ms_per_step"], "tiers ms", d["phase_stamps"]["kernel_ms"])
PY
done
# leave the tree building the default kernel again
touch abismal_amd/csrc/*.hip; make -C abismal_amd/csrc -j8 2>&1 | grep -E "error"
