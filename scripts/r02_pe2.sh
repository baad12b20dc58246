#!/bin/bash
export ABM_EXPERIMENTS=1  # (the experiment variables below are honoured only with this set: abm_api.hip, experiment_env)
set -u
export ABM_BENCH_GENOME_MBP=3100
python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_scale_parity.py tests/test_gpu_params.py tests/test_gpu_cli_goldens.py -q -x 2>&1 | tail -4
python bench.py --pe --reads 1000000 --read-len 150 --cpu-sample 1200000 --no-e2e 2> gpurun_out/r02_pe4.err | tail -1 > gpurun_out/r02_pe4.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_pe4.json"))
r = d["roofline"]
print("PE coop: reads/s", d["value"], "ms/step", d["ms_per_step"], "tier ms", r.get("tier1_ms_per_launch"), r.get("tier2_ms_per_launch")); print("cpu", d["cpu_baseline"])
PY
ABM_COOP_WINDOWS=0 python bench.py --pe --reads 1000000 --read-len 150 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('PE one lane per window: reads/s', d['value'], 'ms/step', d['ms_per_step'], 'tier ms', r.get('tier1_ms_per_launch'), r.get('tier2_ms_per_launch'))"
