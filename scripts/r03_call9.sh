set -u
mkdir -p gpurun_out
cp abismal_amd/_ab/libabismal_amd_d1.so abismal_amd/libabismal_amd.so
( ABM_EXPERIMENTS=1 ABM_DIRECT_MIN=64 timeout 1800 python -m pytest tests/test_gpu_se_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_scale_parity.py tests/test_gpu_params.py tests/test_gpu_edges_and_properties.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r03_call9_tests.log 2>&1
tail -5 gpurun_out/r03_call9_tests.log
VARIANTS="d0 d1" bash scripts/r03_ab.sh 2>&1 | tee gpurun_out/r03_call9_ab.log
cp abismal_amd/_ab/libabismal_amd_d1.so abismal_amd/libabismal_amd.so
export ABM_BENCH_GENOME_MBP=3100
for m in 256 4096 16384; do
  ABM_EXPERIMENTS=1 ABM_DIRECT_MIN=$m python bench.py --steps 4 --warmup 1 --no-e2e --no-other-configs --no-cpu-baseline --no-stage-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('d1 direct_min $m', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['avg_kernel_ms'], 'probes', d['work_per_read']['search_probes'])" | tee -a gpurun_out/r03_call9_ab.log
done
