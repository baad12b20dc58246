# Round-3 verification of the final build: the whole GPU suite, the default bench line, then the round profile.
set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call31_tests.log 2>&1
tail -3 gpurun_out/r03_call31_tests.log
( timeout 1500 python bench.py > gpurun_out/r03_call31_bench.json 2> gpurun_out/r03_call31_bench.err )
tail -c 3000 gpurun_out/r03_call31_bench.json
tail -5 gpurun_out/r03_call31_bench.err
bash scripts/profile_round.sh 2>&1 | tail -20
bash scripts/r03_pe_pmc.sh 2>&1 | tee gpurun_out/r03_pe_pmc_direct.log | tail -3
