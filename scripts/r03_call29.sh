# direct narrowing of big ranges in the pair kernels: parity (incl. thresholds), rate
set -u
mkdir -p gpurun_out
( timeout 1800 python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_scale_parity.py tests/test_gpu_params.py tests/test_gpu_edges_and_properties.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r03_call29_tests.log 2>&1
tail -3 gpurun_out/r03_call29_tests.log
( ABISMAL_AMD_LIB=$(pwd)/abismal_amd/_ab/libabismal_amd_direct256.so timeout 1800 python -m pytest tests/test_gpu_pe_parity.py tests/test_gpu_seed_extension.py tests/test_gpu_scale_parity.py -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r03_call29_tests_256.log 2>&1
tail -3 gpurun_out/r03_call29_tests_256.log
VARIANTS="nodirect direct1024 direct256" bash scripts/r03_pe_ab.sh 2>&1 | tee gpurun_out/r03_exp_pe_direct_narrowing.log
python3 - <<'PY'
import json
for v in ("nodirect", "direct1024", "direct256"):
    try:
        d = json.load(open(f"gpurun_out/r03_pe_ab_{v}_1.json"))
        print(v, "work per pair", d["roofline"]["work_per_pair"])
    except Exception as e:
        print(v, e)
PY
