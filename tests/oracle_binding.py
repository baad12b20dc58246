"""ctypes binding of oracle/_build/liboracle.so — the CPU restatement used as the
parity checker.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
CLI = os.path.join(ORACLE_DIR, "_build", "abismal_oracle")

HIT_DTYPE = np.dtype([("diffs", "<i2"), ("flags", "<u2"), ("pos", "<u4")])
PAIR_DTYPE = np.dtype([("aln_score", "<i2"), ("reserved", "<i2"), ("r1", HIT_DTYPE), ("r2", HIT_DTYPE)])
WORK_KEYS = ["reads", "seed_iters", "search_probes", "candidates", "words", "set_updates", "aligns",
             "aligns_tb", "dp_cells"]


def build_oracle():
    r = subprocess.run(["make", "-C", ORACLE_DIR, "-j4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout)


def _blob(reads):
    bs = [r if isinstance(r, bytes) else r.encode() for r in reads]
    off = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    return np.frombuffer(b"".join(bs) + b"\0", dtype=np.uint8).copy(), off


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp = C.c_void_p
        lib.abo_last_error.restype = C.c_char_p
        lib.abo_index_build.argtypes = [C.c_char_p, C.c_char_p, C.c_uint]
        lib.abo_index_load.argtypes = [C.c_char_p]
        lib.abo_index_load.restype = vp
        lib.abo_index_free.argtypes = [vp]
        lib.abo_mapper_new.argtypes = [vp, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32, C.c_int]
        lib.abo_mapper_new.restype = vp
        lib.abo_mapper_free.argtypes = [vp]
        lib.abo_simulate.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                     C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_double]
        lib.abo_map_se.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, vp, vp, C.c_uint32, vp, C.c_uint, vp]
        lib.abo_map_pe.argtypes = [vp, C.c_int, C.c_uint64, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_uint32, vp,
                                   vp, C.c_uint, vp]

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + self.lib.abo_last_error().decode())

    def index_build(self, fasta, out, threads=1, targets=None, window=20):
        if targets or window != 20:
            self.lib.abo_index_build_opts.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, C.c_char_p, C.c_uint]
            self._chk(self.lib.abo_index_build_opts(os.fsencode(fasta), os.fsencode(targets or ""), window, os.fsencode(out), threads))
        else:
            self._chk(self.lib.abo_index_build(os.fsencode(fasta), os.fsencode(out), threads))

    def index_load(self, path):
        h = self.lib.abo_index_load(os.fsencode(path))
        if not h:
            raise RuntimeError("oracle: " + self.lib.abo_last_error().decode())
        return h

    def index_free(self, h):
        self.lib.abo_index_free(h)

    def simulate(self, fasta, prefix, n_reads, single_end=False, pbat=False, random_pbat=False, read_len=100,
                 min_frag=100, max_frag=250, seed=1, mut=0.01, bis=0.98):
        self._chk(self.lib.abo_simulate(os.fsencode(fasta), os.fsencode(prefix), int(single_end), int(pbat),
                                        int(random_pbat), read_len, min_frag, max_frag, n_reads, seed, mut, bis))

    def map_se(self, index, reads, mode=0, max_candidates=0, valid_frac=0.1, threads=1, cig_stride=None):
        blob, off = _blob(reads)
        n = len(off) - 1
        ml = int((off[1:] - off[:-1]).max()) if n else 0
        cig_stride = cig_stride or (ml + 2)
        mp = self.lib.abo_mapper_new(index, max_candidates, valid_frac, 32, 3000, 0)
        res = np.zeros(n, dtype=HIT_DTYPE)
        cig = np.zeros(max(1, n * cig_stride), dtype=np.uint32)
        cig_n = np.zeros(n, dtype=np.uint32)
        work = np.zeros(9, dtype=np.uint64)
        try:
            self._chk(self.lib.abo_map_se(mp, mode, n, blob.ctypes.data, off.ctypes.data, res.ctypes.data,
                                          cig.ctypes.data, cig_stride, cig_n.ctypes.data, threads,
                                          work.ctypes.data))
        finally:
            self.lib.abo_mapper_free(mp)
        return res, cig.reshape(n, cig_stride) if n else cig.reshape(0, cig_stride), cig_n, dict(
            zip(WORK_KEYS, [int(x) for x in work]))

    def map_pe(self, index, reads1, reads2, mode=0, max_candidates=0, valid_frac=0.1, min_frag=32, max_frag=3000,
               allow_ambig=False, threads=1):
        b1, o1 = _blob(reads1)
        b2, o2 = _blob(reads2)
        n = len(o1) - 1
        ml = max(int((o1[1:] - o1[:-1]).max()) if n else 0, int((o2[1:] - o2[:-1]).max()) if n else 0)
        st = ml + 2
        mp = self.lib.abo_mapper_new(index, max_candidates, valid_frac, min_frag, max_frag, int(allow_ambig))
        pairs = np.zeros(n, dtype=PAIR_DTYPE)
        se1 = np.zeros(n, dtype=HIT_DTYPE)
        se2 = np.zeros(n, dtype=HIT_DTYPE)
        c1 = np.zeros(max(1, n * st), dtype=np.uint32)
        c2 = np.zeros(max(1, n * st), dtype=np.uint32)
        n1 = np.zeros(n, dtype=np.uint32)
        n2 = np.zeros(n, dtype=np.uint32)
        work = np.zeros(9, dtype=np.uint64)
        try:
            self._chk(self.lib.abo_map_pe(mp, mode, n, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data,
                                          o2.ctypes.data, pairs.ctypes.data, se1.ctypes.data, se2.ctypes.data,
                                          c1.ctypes.data, c2.ctypes.data, st, n1.ctypes.data, n2.ctypes.data,
                                          threads, work.ctypes.data))
        finally:
            self.lib.abo_mapper_free(mp)
        return pairs, se1, se2, (c1.reshape(n, st), n1), (c2.reshape(n, st), n2), dict(
            zip(WORK_KEYS, [int(x) for x in work]))


def load(build=False):
    if build or not os.path.exists(LIB):
        build_oracle()
    return Oracle(C.CDLL(LIB))


def read_fastq_like_readloader(path):
    """FASTQ -> (names, reads) with the trimming/skip rules of ReadLoader
    (src/abismal.cpp:164-201): reads with < 44 non-N bases become empty; others
    lose trailing Ns and everything before the first A/C/G/T."""
    names, reads = [], []
    with open(path) as f:
        for k, line in enumerate(f):
            line = line.rstrip("\n")
            if k % 4 == 0:
                cut = min([i for i in (line.find(" "), line.find("\t")) if i >= 0] or [len(line)])
                names.append(line[1:cut])
            elif k % 4 == 1:
                if sum(1 for c in line if c != "N") < 44:
                    line = ""
                else:
                    line = line.rstrip("N")
                    first = min([i for i in (line.find(b) for b in "ACGT") if i >= 0])
                    line = line[first:]
                reads.append(line)
    return names, reads
