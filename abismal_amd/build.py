"""In-tree build of libabismal_amd.so (hipcc, gfx950)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose: bool = False) -> str:
    """Compile every HIP/C++ source under csrc/ into abismal_amd/libabismal_amd.so."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libabismal_amd.so failed")
    return os.path.join(_HERE, "libabismal_amd.so")
