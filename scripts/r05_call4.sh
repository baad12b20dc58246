#!/bin/bash
# round 5, call 4: what the tier-2 tail is made of (per-pair phase cycles of the diagnostic kernels), the paired-end end-to-end
# rate with the split kernels (8 M pairs; mapper contexts 8 / 16), and the box's tmpfs / memory limits for the full-size runs
mkdir -p gpurun_out
export ABM_BENCH_GENOME_MBP=3100
{
echo "== box"; df -h /dev/shm /tmp | sed 's/^/   /'; cat /sys/fs/cgroup/memory.max 2>/dev/null | sed 's/^/   memory.max /'; cat /sys/fs/cgroup/cpu.max 2>/dev/null | sed 's/^/   cpu.max /'; free -g | sed 's/^/   /'
} > gpurun_out/r05_box.log 2>&1
cat gpurun_out/r05_box.log
python bench.py --pe --reads 1000000 --read-len 150 --steps 16 --warmup 16 --cpu-sample 200000 --e2e-reads 2000000 --e2e-check 20000 --e2e-copies 4 2> gpurun_out/r05_call4_pe.err | tail -1 > gpurun_out/r05_call4_pe.json
python3 - <<'PY' | tee gpurun_out/r05_pe_tail_anatomy.log
import json
d = json.load(open('gpurun_out/r05_call4_pe.json'))
print("value %.3f M reads/s, e2e %s (over kernel %s), e2e runs %s" % (d["value"] / 1e6, d.get("e2e_reads_per_s"), d.get("e2e_over_kernel"), (d.get("e2e") or {}).get("seconds_of_each_run")))
print("cli", (d.get("e2e") or {}).get("cli"))
print("parity", (d.get("cpu_baseline") or {}).get("pairs_hits_fallbacks_cigars_vs_oracle"), (d.get("e2e") or {}).get("parity"))
ps = d["phase_stamps"]
print("alone ms", ps["kernel_ms"])
for row in ps["by_set_size"]:
    print(row)
for row in ps["slowest_pairs"]:
    print(row)
PY
# mapper contexts per GPU: 8 against the default 16 on the same 8 M pairs
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa; CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_c4; mkdir -p $WD
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
for k in 1 2; do for f in 1 2 3 4; do cat $WD/p_$k.fq; done > $WD/x_$k.fq; done
{
for m in 16 8 4; do
  for rep in 1 2; do
    $CLI map -mappers $m -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/x_1.fq $WD/x_2.fq 2> $WD/err.log || tail -3 $WD/err.log
    python3 -c "
import json; t=json.load(open('$WD/t.json')); print('mappers $m rep $rep: %.2f M reads/s  %.3f s  batches %s' % (t['reads']/t['seconds']/1e6, t['seconds'], t.get('batches_per_gpu')))"
  done
done
} 2>&1 | tee gpurun_out/r05_pe_e2e_mappers.log
rm -rf $WD
