# paired-end: tier-1 list capacity 64 / 128 / 256 entries (16 slots)
set -u
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
for rep in 1 2 3; do
  for v in cap64 cap48 cap32 cap96; do
    EXTRA="--no-cpu-baseline"; [ $rep = 1 ] && EXTRA="--cpu-sample 50000"
    ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_$v.so python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 $EXTRA 2>/dev/null | tail -1 > /tmp/pe.json
    python3 -c "
import json
d = json.load(open('/tmp/pe.json'))
c = d.get('cpu_baseline') or {}
print('$v rep $rep reads/s', d['value'], 'ms/step', d['ms_per_step'], 'tier2 share of candidates', d['roofline'].get('tier2_share_of_candidates'), c.get('pairs_hits_fallbacks_cigars_vs_oracle', ''))"
  done
done 2>&1 | tee gpurun_out/r03_exp_pe_tier1_cap_small.log
