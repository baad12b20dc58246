# device-side timeline of the CLI run + paired-end phase shares (diagnostic kernels)
set -u
mkdir -p gpurun_out
( timeout 900 python bench.py --no-e2e --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r03_call13_prep.json 2> gpurun_out/r03_call13_prep.err )
tail -2 gpurun_out/r03_call13_prep.err
bash scripts/r03_cli_gputrace.sh 2>&1 | tail -60
( timeout 900 python bench.py --pe --reads 1000000 --read-len 150 --steps 12 --warmup 12 --no-cpu-baseline --phase-stamps > gpurun_out/r03_call13_pe.json 2> gpurun_out/r03_call13_pe.err )
tail -c 2500 gpurun_out/r03_call13_pe.json
