#!/bin/bash
# last pass of the round with the committed build: GPU suite, profile round, default bench line (with e2e), PE line
set -u
export ABM_BENCH_GENOME_MBP=3100
python -m pytest tests -m gpu -q 2>&1 | tail -3
bash scripts/profile_round.sh > gpurun_out/r02_profile_round.log 2>&1
tail -2 gpurun_out/r02_profile_round.log
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -1 gpurun_out/r02_bench_default.err | cut -c1-700
python bench.py --pe --reads 1000000 --read-len 150 --cpu-sample 1200000 --no-e2e 2> gpurun_out/r02_pe_final.err | tail -1 > gpurun_out/r02_pe_final.json
python -c "
import json; d=json.load(open('gpurun_out/r02_pe_final.json')); print('PE', d['value'], d['ms_per_step'])"
