# counters of the final kernels: SQ/TCC of map_se_kernel, memory-side requests of the paired-end kernels
set -u
mkdir -p gpurun_out
bash scripts/r03_pmc.sh 2>&1 | tee gpurun_out/r03_pmc_sq_tcc_final.log
bash scripts/r03_pe_pmc.sh 2>&1 | tee gpurun_out/r03_pe_pmc.log
