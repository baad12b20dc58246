"""The candidate sets of the kernels (single-end: lane-resident; paired-end: in memory) against libstdc++'s heap
calls (what se_candidates / pe_candidates are made of), on the GPU: small HIP programs that include the kernel
source, built here with hipcc."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["se_set_check", "pe_set_check", "wave_prims_check"])
def test_candidate_set_equals_libstdcxx_heap(tmp_path, prog):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / prog
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "hip", prog + ".hip"), "-o", str(exe)], check=True, timeout=1500)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
