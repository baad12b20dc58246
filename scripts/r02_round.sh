#!/bin/bash
export ABM_EXPERIMENTS=1  # (the experiment variables below are honoured only with this set: abm_api.hip, experiment_env)
# the bit-plane filter for real: GPU suite, default bench line (1 M-read parity sample), launch sizes, variants
set -u
export ABM_BENCH_GENOME_MBP=3100
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 3 --warmup 1 --no-e2e --cpu-sample 1000000 2> gpurun_out/r02_real.err | tail -1 > gpurun_out/r02_real.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_real.json"))
r = d["roofline"]
print("10M:", d["value"], "reads/s; ms/step", d["ms_per_step"], "kernel", r["avg_kernel_ms"], "frac", r["frac"], d["filter_genome"])
print("shares", d.get("phase_shares_diagnostic")); print("cpu", {k: v for k, v in d["cpu_baseline"].items() if "identical" in k or k == "value"})
PY
for n in 1000000 4000000; do python bench.py --reads $n --steps 4 --warmup 1 --no-e2e --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('reads', d['config']['reads_per_step_per_gpu'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"; done
ABM_PLANES_COPIES=1 python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one copy: ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"
cp abismal_amd/_ab/libabismal_amd_r4.so abismal_amd/libabismal_amd.so
python bench.py --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('4 rounds: ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'])"
cp abismal_amd/_ab/libabismal_amd_new.so abismal_amd/libabismal_amd.so
