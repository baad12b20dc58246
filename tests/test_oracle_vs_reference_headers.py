"""CPU: the oracle's aligner, encodings and hashes against the REAL reference code
that is header-only and therefore compiles here (oracle/_ref/libref_probe.so, built
from /root/reference/src/AbismalAlign.hpp etc. where they lie).  Skipped when the
probe library is absent (e.g. a machine without the reference tree and without the
prebuilt file)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_probe.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/libref_probe.so not built")


@pytest.fixture(scope="module")
def libs(oracle):
    ref = C.CDLL(REF)
    orc = oracle.lib
    for lib, pre in ((ref, "ref_"), (orc, "abo_")):
        f = getattr(lib, pre + "align")
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int,
                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("roll2", "roll3"):
            getattr(lib, pre + name).restype = C.c_uint32
    return ref, orc


def test_encoding_tables_and_symbols(libs):
    ref, orc = libs
    for c in range(128):
        for a in (0, 1):
            assert ref.ref_read_code(c, a) == orc.abo_read_code(c, a), chr(c)
        assert ref.ref_genome_code(c) == orc.abo_genome_code(c), chr(c)
    assert ref.ref_genome_code(ord("N")) == 0  # not 15: see DESIGN.md
    for nt in range(16):
        assert ref.ref_get_bit(nt) == orc.abo_get_bit(nt)
        for conv in (0, 1):
            assert ref.ref_trit(nt, conv) == orc.abo_trit(nt, conv)


def test_rolling_hashes(libs):
    ref, orc = libs
    rng = np.random.default_rng(0)
    for conv in (0, 1):
        k2r = k2o = k3r = k3o = 0
        for nt in rng.integers(0, 16, 3000).tolist():
            k2r, k2o = ref.ref_roll2(k2r, nt), orc.abo_roll2(k2o, nt)
            k3r, k3o = ref.ref_roll3(k3r, nt, conv), orc.abo_roll3(k3o, nt, conv)
            assert k2r == k2o and k3r == k3o


def _align(lib, pre, genome, q, diffs, md, pos, tb):
    cig = np.zeros(600, dtype=np.uint32)
    n = C.c_uint32(0); alen = C.c_uint32(0); npos = C.c_uint32(0); nm = C.c_int(0)
    s = getattr(lib, pre + "align")(genome.ctypes.data, len(genome), q.ctypes.data, len(q), diffs, md, pos, int(tb),
                                    cig.ctypes.data, len(cig), C.byref(n), C.byref(alen), C.byref(npos), C.byref(nm))
    return (s, cig[:n.value].tolist(), alen.value, npos.value, nm.value) if tb else (s,)


def test_aligner_matches_reference(libs):
    ref, orc = libs
    rng = np.random.default_rng(1)
    n_bases = 40000
    gn = rng.choice([1, 2, 4, 8], n_bases).astype(np.uint64)
    gn[5000:5050] = 0  # an N stretch (nibble 0)
    genome = np.zeros((n_bases + 15) // 16 + 2, dtype=np.uint64)
    for k in range(16):
        genome[: len(gn[k::16])] |= gn[k::16] << np.uint64(4 * k)
    t_rich = {1: 1, 2: 2, 4: 4, 8: 10}
    checked = 0
    for trial in range(600):
        L = int(rng.choice([44, 60, 100, 150, 251]))
        pos = int(rng.integers(200, n_bases - 400))
        src = gn[pos:pos + L + 12].tolist()
        out, i = [], 0
        rate = float(rng.choice([0.0, 0.02, 0.06, 0.15]))
        while len(out) < L:
            r = rng.random()
            if r < rate / 3:
                out.append(int(rng.choice([1, 2, 4, 8]))); i += 1
            elif r < 2 * rate / 3:
                out.append(int(rng.choice([1, 2, 4, 8])))
            elif r < rate:
                i += 1
            else:
                out.append(src[i]); i += 1
        q = np.array([t_rich.get(x, 0) for x in out[:L]], dtype=np.uint8)
        if trial % 7 == 0:
            q[rng.integers(0, L, 3)] = 0  # read Ns
        md = int(0.1 * L)
        for diffs in (1, 2, int(rng.integers(1, 40)), 0):
            tp = pos + int(rng.integers(-3, 4))
            a = _align(ref, "ref_", genome, q, diffs, md, tp, False)
            b = _align(orc, "abo_", genome, q, diffs, md, tp, False)
            assert a == b, (trial, L, diffs, a, b)
            a = _align(ref, "ref_", genome, q, diffs, md, tp, True)
            b = _align(orc, "abo_", genome, q, diffs, md, tp, True)
            assert a == b, (trial, L, diffs, a, b)
            checked += 1
    assert checked == 2400
