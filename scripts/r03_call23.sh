# paired-end kernels: one looped copy of the orientation code, and fewer waves per SIMD (more registers)
set -u
mkdir -p gpurun_out
( ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_loop4.so timeout 1800 python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_se_set.py tests/test_gpu_params.py tests/test_gpu_seed_extension.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r03_call23_tests.log 2>&1
tail -3 gpurun_out/r03_call23_tests.log
VARIANTS="peold4 loop4 loop3 pew3 pew2" bash scripts/r03_pe_ab.sh 2>&1 | tee gpurun_out/r03_exp_pe_loop_and_registers.log
