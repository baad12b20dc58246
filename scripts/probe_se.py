import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import oracle_binding as ob
import abismal_amd as A
o = ob.load(build=True)
wd = "/tmp/abw"; os.makedirs(wd, exist_ok=True)
t=time.time(); o.index_build("tests/golden/tRex1.fa", wd+"/t.idx", 8); print("idx build", time.time()-t)
o.simulate("tests/golden/tRex1.fa", wd+"/r", 100000, single_end=True, seed=7)
names, reads = ob.read_fastq_like_readloader(wd+"/r_1.fq")
oix = o.index_load(wd+"/t.idx")
ix = A.Index(wd+"/t.idx"); ctx = A.Context(ix, 0)
for mode in (0,1,2):
    t=time.time(); ores, ocig, ocn, work = o.map_se(oix, reads, mode=mode, threads=16); to=time.time()-t
    t=time.time(); res, cig, coff = ctx.map_se(reads, mode=mode); tg=time.time()-t
    t=time.time(); res, cig, coff = ctx.map_se(reads, mode=mode); tg2=time.time()-t
    same = (res["pos"]==ores["pos"]).mean()
    m = res["pos"]!=0
    dsame = (res["diffs"][m]==ores["diffs"][m]).mean(); fsame=(res["flags"][m]==ores["flags"][m]).mean()
    print(f"mode {mode}: n={len(reads)} oracle {to:.2f}s gpu {tg:.3f}s/{tg2:.3f}s mapped gpu={m.sum()} oracle={(ores['pos']!=0).sum()} pos-eq={same:.5f} diffs-eq={dsame:.5f} flags-eq={fsame:.5f} ambig={(res['flags'][m]&0x100!=0).sum()}")
    print("  oracle work", work, "gpu work", ctx.take_work())
