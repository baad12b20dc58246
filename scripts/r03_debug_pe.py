"""paired-end on the repeat-rich test genome with and without seed-extension tables: mismatches against the oracle
(diagnosis of tests/test_gpu_seed_extension.py::test_pe_with_and_without_tables)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abismal_amd as A
from tests import oracle_binding as ob, synth
from tests.test_gpu_pe_parity import compare_pe

o = ob.load(build=not os.path.exists(ob.LIB))
wd = tempfile.mkdtemp()
fa, idx = os.path.join(wd, "rep.fa"), os.path.join(wd, "rep.idx")
synth.repeat_rich_genome(fa)
A.index_build(fa, idx, 8)
oix = o.index_load(idx)
has_ext = hasattr(A.load_library(), "abm_index_set_seed_extension")
for seed in (3, 4):
    r1, r2 = synth.mutated_pairs(fa, 3000, 100, seed=seed)
    orc = o.map_pe(oix, r1, r2, mode=0, threads=8)
    for letters in ([(0, 0), (2, 1), (3, 2), (0, 0)] if has_ext else [None]):
        ix = A.Index(idx, seed_extension=letters) if has_ext else A.Index(idx)
        ctx = A.Context(ix, 0)
        for rep in range(2):
            try:
                compare_pe(ctx.map_pe(r1, r2, mode=0), orc, f"seed {seed} tables {letters} call {rep}")
                print("seed", seed, "tables", letters, "call", rep, "identical", flush=True)
            except AssertionError as e:
                print(str(e)[:400], flush=True)
        ctx.close(); ix.close()
