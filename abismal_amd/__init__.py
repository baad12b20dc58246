"""abismal_amd — MI355X-native implementation of abismal's mapping hot path.

The product is the C-ABI shared library ``libabismal_amd.so`` (HIP kernels for
gfx950 + C++ host code, sources in ``abismal_amd/csrc``, interface in
``include/abismal_amd.h``).  This package is the thin ctypes binding used by the
tests and by ``bench.py``; it never computes anything itself and raises if the
library is missing.
"""
from .api import (  # noqa: F401
    AbismalAmdError, Index, index_build, Context, Params, lib_path, load_library,
    SE_T_RICH, SE_A_RICH, SE_RANDOM, PE_NORMAL, PE_PBAT, PE_RANDOM,
    HIT_DTYPE, PAIR_DTYPE, EXPORTED_SYMBOLS,
)
from .build import build  # noqa: F401
