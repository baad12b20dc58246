#!/bin/bash
# round 5: BASELINE configs 3, 5 and 4 once at their stated sizes, on the one device there is.
#   config 3: 50 M pairs 2x150 (sim -n 50000000 -l 150 -min-fraglen 150 -max-fraglen 500), abismal-amd map, SAM prefix md5 == oracle CLI
#   config 5: 50 M reads 150 bp random PBAT (sim -single -R), map -R, SAM prefix md5 == oracle CLI
#   config 4: 200 M reads x 100 bp through eight replicas on device 0 with eight part files, through -gpus 1, and through eight
#             virtual GPUs: bodies compared byte for byte, statistics identical, time to each region's first batch, peak RSS, pinned memory
# The three simulators run side by side in the background from the start (one thread each); outputs on tmpfs are deleted as soon as compared.
set -u
mkdir -p gpurun_out
LOG=gpurun_out/r05_full_size.log
export ABM_BENCH_GENOME_MBP=3100 ABM_BENCH_KEEP_FASTA=1
python3 bench.py --steps 1 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split > /dev/null 2> gpurun_out/r05_full_size_index.err
IDX=/tmp/abismal_bench/g3100.idx; FA=/tmp/abismal_bench/g3100.fa; CLI=abismal_amd/abismal-amd
ORACLE=oracle/_build/abismal_oracle
[ -x $ORACLE ] || make -C oracle > /dev/null 2>&1
WD=/dev/shm/abm_full; rm -rf $WD; mkdir -p $WD
T0=$(date +%s)
( S=$(date +%s); $CLI sim -seed 1 -n 50000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/c3 $FA > /dev/null 2> $WD/sim3.err; echo "sim config 3 (50 M pairs 2x150): $(( $(date +%s) - S )) s" > $WD/sim3.log ) &
P3=$!
( S=$(date +%s); $CLI sim -single -R -seed 2 -n 50000000 -l 150 -min-fraglen 150 -max-fraglen 500 -m 0.01 -b 0.98 -o $WD/c5 $FA > /dev/null 2> $WD/sim5.err; echo "sim config 5 (50 M reads x 150, -R): $(( $(date +%s) - S )) s" > $WD/sim5.log ) &
P5=$!
( S=$(date +%s); $CLI sim -single -seed 3 -n 200000000 -l 100 -m 0.01 -b 0.98 -o $WD/c4 $FA > /dev/null 2> $WD/sim4.err; echo "sim config 4 (200 M reads x 100): $(( $(date +%s) - S )) s" > $WD/sim4.log ) &
P4=$!
body_md5() { grep -v '^@PG' "$@" | md5sum | cut -c1-32; }
row() {  # label, timing json
  python3 - "$1" "$2" <<'PY'
import json, sys
t = json.load(open(sys.argv[2]))
print("%-58s %7.2f M reads/s  %8.3f s  %d reads  batches/GPU %s  first batch of each region at %s s  count done %.2f s  peak RSS %d MB  pinned %s MB  host threads %d" % (
    sys.argv[1], t["reads"] / t["seconds"] / 1e6, t["seconds"], t["reads"], t["batches_per_gpu"], [round(x, 2) for x in t.get("region_first_batch_s", [])],
    t.get("count_done_s", -1), t.get("peak_rss_mb", -1), t.get("pinned_mb"), t["host_threads"]))
PY
}
exec > >(tee $LOG) 2>&1
echo "== box: $(nproc) hardware threads, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null), tmpfs $(df -h /dev/shm | tail -1 | awk '{print $2}')"
# ---- config 5
wait $P5; cat $WD/sim5.log | tail -1
ls -la $WD/c5_1.fq | awk '{print "   FASTQ bytes", $5}'
$CLI map -R -i $IDX -o $WD/c5.sam -s $WD/c5.st -timing $WD/t.json $WD/c5_1.fq 2> $WD/err.log || tail -3 $WD/err.log
row "config 5: 50 M reads x 150 bp random PBAT, map -R" $WD/t.json
head -800000 $WD/c5_1.fq > $WD/c5p.fq
$CLI map -R -i $IDX -o $WD/c5p.sam $WD/c5p.fq 2> /dev/null
$ORACLE map -R -t 64 -i $IDX -o $WD/c5o.sam $WD/c5p.fq 2> /dev/null
N=$(grep -vc '^@' $WD/c5p.sam)
echo "   parity, first 200000 reads: product $(body_md5 $WD/c5p.sam)  oracle CLI $(body_md5 $WD/c5o.sam)  the full run's first $N records $(grep -v '^@' $WD/c5.sam | head -$N | md5sum | cut -c1-32) / $(grep -v '^@' $WD/c5p.sam | md5sum | cut -c1-32)"
rm -f $WD/c5*
# ---- config 3
wait $P3; cat $WD/sim3.log | tail -1
ls -la $WD/c3_1.fq $WD/c3_2.fq | awk '{print "   FASTQ bytes", $5}'
$CLI map -i $IDX -o $WD/c3.sam -s $WD/c3.st -timing $WD/t.json $WD/c3_1.fq $WD/c3_2.fq 2> $WD/err.log || tail -3 $WD/err.log
row "config 3: 50 M pairs 2x150, map (8 contexts)" $WD/t.json
ls -la $WD/c3.sam | awk '{print "   SAM bytes", $5}'
head -80000 $WD/c3_1.fq > $WD/c3p_1.fq; head -80000 $WD/c3_2.fq > $WD/c3p_2.fq
$CLI map -i $IDX -o $WD/c3p.sam $WD/c3p_1.fq $WD/c3p_2.fq 2> /dev/null
$ORACLE map -t 64 -i $IDX -o $WD/c3o.sam $WD/c3p_1.fq $WD/c3p_2.fq 2> /dev/null
N=$(grep -vc '^@' $WD/c3p.sam)
echo "   parity, first 20000 pairs: product $(body_md5 $WD/c3p.sam)  oracle CLI $(body_md5 $WD/c3o.sam)  the full run's first $N records $(grep -v '^@' $WD/c3.sam | head -$N | md5sum | cut -c1-32) / $(grep -v '^@' $WD/c3p.sam | md5sum | cut -c1-32)"
rm -f $WD/c3.sam
$CLI map -mappers 16 -i $IDX -o $WD/c3.sam -timing $WD/t.json $WD/c3_1.fq $WD/c3_2.fq 2> $WD/err.log || tail -3 $WD/err.log
row "config 3 again with 16 contexts" $WD/t.json
rm -f $WD/c3*
# ---- config 4
wait $P4; cat $WD/sim4.log | tail -1
ls -la $WD/c4_1.fq | awk '{print "   FASTQ bytes", $5}'
$CLI map -devices 0,0,0,0,0,0,0,0 -out-parts 8 -i $IDX -o $WD/c4.sam -s $WD/c4.st -timing $WD/t.json $WD/c4_1.fq 2> $WD/err.log || tail -3 $WD/err.log
row "config 4: 200 M reads x 100 bp, 8 replicas on device 0, 8 parts" $WD/t.json
$CLI map -gpus 1 -i $IDX -o $WD/c4one.sam -s $WD/c4one.st -timing $WD/t.json $WD/c4_1.fq 2> $WD/err.log || tail -3 $WD/err.log
row "config 4 input through -gpus 1, one file" $WD/t.json
H1=$(grep -c '^@' $WD/c4one.sam); H2=$(grep -c '^@' $WD/c4.sam.part000)
if cmp -s <(tail -n +$((H1 + 1)) $WD/c4one.sam) <(cat $WD/c4.sam.part00? | tail -n +$((H2 + 1))); then echo "   SAM body of the 8 parts (cat) == the one file of -gpus 1, byte for byte ($(stat -c %s $WD/c4one.sam) bytes)"; else echo "   SAM BODIES DIFFER"; fi
if cmp -s $WD/c4.st $WD/c4one.st; then echo "   statistics identical"; else echo "   STATISTICS DIFFER"; fi
rm -f $WD/c4one.sam $WD/c4.sam.part*
$CLI map -virtual-gpus 8 -out-parts 8 -i $IDX -o $WD/c4v.sam -timing $WD/t.json $WD/c4_1.fq 2> $WD/err.log || tail -3 $WD/err.log
row "config 4 input around 8 virtual GPUs, 8 parts (host pipeline alone)" $WD/t.json
echo "== wall clock of the whole script: $(( $(date +%s) - T0 )) s"
rm -rf $WD
