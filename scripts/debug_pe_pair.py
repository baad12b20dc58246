"""Debug helper: map one simulated tRex1 pair on the GPU (index of the pair as argv[1])."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import abismal_amd.api as A
from tests import oracle_binding as ob
o = ob.load(build=True)
wd = tempfile.mkdtemp()
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tRex1.fa")
idx = os.path.join(wd, "t.idx")
A.index_build(gold, idx)
prefix = os.path.join(wd, "pe")
o.simulate(gold, prefix, 10000)
_, r1 = ob.read_fastq_like_readloader(prefix + "_1.fq")
_, r2 = ob.read_fastq_like_readloader(prefix + "_2.fq")
k = int(sys.argv[1])
ix = A.Index(idx)
ctx = A.Context(ix)
print(ctx.map_pe([r1[k]], [r2[k]], mode=0)[0])
