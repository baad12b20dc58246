"""The lane-resident candidate set of the single-end kernel against libstdc++'s heap calls (what se_candidates
is made of), on the GPU: a small HIP program that includes the kernel source, built here with hipcc."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_se_set_equals_libstdcxx_heap(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = tmp_path / "se_set_check"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "hip", "se_set_check.hip"), "-o", str(exe)], check=True, timeout=900)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
