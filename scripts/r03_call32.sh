# paired-end: which register budget per tier with the direct narrowing on, and how many slots in flight
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100 ABM_EXPERIMENTS=1
run() {  # label, env assignments..., then -- bench args
  local label="$1"; shift
  env "$@" python bench.py --pe --reads 1000000 --read-len 150 --steps 12 --warmup 12 --no-cpu-baseline $EXTRA 2>/dev/null | tail -1 > /tmp/pe.json
  python3 -c "
import json
d = json.load(open('/tmp/pe.json'))
print('$label reads/s', d['value'], 'ms/step', d['ms_per_step'])"
}
for rep in 1 2; do
  EXTRA="" run "default(t1=3,t2=4) rep $rep" X=1
  EXTRA="" run "t1=4,t2=4 rep $rep" ABM_PE_WPS=4
  EXTRA="" run "t1=3,t2=3 rep $rep" ABM_PE_WPS2=3
  EXTRA="--streams 8" run "streams 8 rep $rep" X=1
  EXTRA="--streams 16" run "streams 16 rep $rep" X=1
done 2>&1 | tee gpurun_out/r03_exp_pe_budgets_and_slots.log
