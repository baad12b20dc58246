mkdir -p gpurun_out
bash scripts/r03_sector_probe.sh 2>&1 | grep -v "warning\|hipEvent\|\^\||" | tee gpurun_out/r03_sector_probe.log
