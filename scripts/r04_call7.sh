#!/bin/bash
# round 4, call 7: paired-end end to end (how many batches in flight, how large) and BGZF input, on the bench index
mkdir -p gpurun_out
export ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r04_call7_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa
CLI=abismal_amd/abismal-amd
WD=/dev/shm/abm_pe; mkdir -p $WD
$CLI sim -seed 1 -n 2000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/p $FA > /dev/null
for e in 1 2; do cat $WD/p_$e.fq $WD/p_$e.fq $WD/p_$e.fq $WD/p_$e.fq > $WD/x_$e.fq; done
one() { # label args
  local label="$1"; shift
  $CLI map "$@" -i $IDX -o $WD/out.sam -timing $WD/t.json $WD/x_1.fq $WD/x_2.fq 2> $WD/err.log || tail -3 $WD/err.log
  python3 -c "
import json; t=json.load(open('$WD/t.json')); print('%-40s %6.2f M reads/s  %.3f s  batches %s busy %s' % ('$label', t['reads']/t['seconds']/1e6, t['seconds'], t['batches_per_gpu'], {k: round(v,2) for k,v in t['busy_s'].items()}))"
}
{
one "default (8 mappers, 1 M pairs)"
one "default again"
one "4 mappers, 1 M" -mappers 4
one "12 mappers, 1 M" -mappers 12
one "8 mappers, 512 k" -batch 524288
one "12 mappers, 512 k" -mappers 12 -batch 524288
one "16 mappers, 512 k" -mappers 16 -batch 524288
one "3 mappers, 2 M (round 3)" -mappers 3 -batch 2097152
} > gpurun_out/r04_pe_e2e_variants.log 2>&1
cat gpurun_out/r04_pe_e2e_variants.log
rm -f $WD/x_*.fq $WD/out.sam
# BGZF input, single-end
$CLI sim -single -seed 1 -n 10000000 -l 100 -m 0.01 -b 0.98 -o $WD/s $FA > /dev/null
python3 - $WD/s_1.fq $WD/s.fq.bgz <<'PY'
import sys, zlib, struct
from concurrent.futures import ThreadPoolExecutor
def blk(d):
    co = zlib.compressobj(1, zlib.DEFLATED, -15); z = co.compress(d) + co.flush()
    return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(z) + 25) + z + struct.pack("<II", zlib.crc32(d), len(d))
with open(sys.argv[1], "rb") as f, open(sys.argv[2], "wb") as o, ThreadPoolExecutor(16) as pool:
    while True:
        big = f.read(64 << 20)
        if not big: break
        for b in pool.map(blk, [big[k:k + 0xff00] for k in range(0, len(big), 0xff00)]): o.write(b)
    o.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
PY
{
for rep in 1 2 3; do
$CLI map -i $IDX -o $WD/bz.sam -timing $WD/t.json $WD/s.fq.bgz 2> $WD/err.log || tail -3 $WD/err.log
python3 -c "
import json; t=json.load(open('$WD/t.json')); print('BGZF input  %6.2f M reads/s  %.3f s  busy %s cpu %s throttled %s' % (t['reads']/t['seconds']/1e6, t['seconds'], {k: round(v,2) for k,v in t['busy_s'].items()}, t['cpu_s'], t['throttled_s']))"
done
$CLI map -i $IDX -o $WD/pl.sam -timing $WD/t.json $WD/s_1.fq 2> /dev/null
python3 -c "
import json; t=json.load(open('$WD/t.json')); print('plain input %6.2f M reads/s  %.3f s' % (t['reads']/t['seconds']/1e6, t['seconds']))"
cmp <(grep -v '^@PG' $WD/bz.sam) <(grep -v '^@PG' $WD/pl.sam) && echo "SAM bodies identical"
} > gpurun_out/r04_bgzf_input.log 2>&1
cat gpurun_out/r04_bgzf_input.log
rm -rf $WD
