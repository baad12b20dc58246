# device-side timeline of the CLI run (10 M reads)
set -u
mkdir -p gpurun_out
( ABM_BENCH_KEEP_FASTA=1 timeout 900 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r03_call14_prep.json 2> gpurun_out/r03_call14_prep.err )
tail -2 gpurun_out/r03_call14_prep.err
bash scripts/r03_cli_gputrace.sh 2>&1 | tail -60
