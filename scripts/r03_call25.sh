# Round-3 verification of the final build: the whole GPU suite, the default bench line, then the round profile.
set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r03_call25_tests.log 2>&1
tail -3 gpurun_out/r03_call25_tests.log
( timeout 1500 python bench.py > gpurun_out/r03_call25_bench.json 2> gpurun_out/r03_call25_bench.err )
tail -c 3000 gpurun_out/r03_call25_bench.json
tail -5 gpurun_out/r03_call25_bench.err
bash scripts/profile_round.sh 2>&1 | tail -20
( timeout 600 python bench.py --pe --reads 1000000 --read-len 100 --steps 12 --warmup 12 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.load(sys.stdin); print('pe 2x100 reads/s', d['value'])" )
