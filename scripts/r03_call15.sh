# after the h_arena reservation fix: e2e (median of three + sustained), then the device-side timeline
set -u
mkdir -p gpurun_out
( ABM_BENCH_KEEP_FASTA=1 timeout 1500 python bench.py --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/r03_call15_bench.json 2> gpurun_out/r03_call15_bench.err )
grep "e2e" gpurun_out/r03_call15_bench.err | cut -c1-900
bash scripts/r03_cli_gputrace.sh 2>&1 | tail -30
grep -c regrown gpurun_out/r03_gputrace_cli.err
