import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import abismal_amd as A
from tests import oracle_binding as ob, synth
o = ob.load(build=True)
wd = "/tmp/dbg"; os.makedirs(wd, exist_ok=True)
fa = wd + "/iupac.fa"; idx = wd + "/iupac.idx"
synth.repeat_rich_genome(fa, seed=21, n_chroms=2, chrom_len=600_000, iupac=60000)
A.index_build(fa, idx, 8)
reads = synth.trim_like_readloader(synth.mutated_reads(fa, 6000, 100, seed=5, mut=0.04))
ix = A.Index(idx); ctx = A.Context(ix, 0); oix = o.index_load(idx)
o_res, o_cig, o_n, _ = o.map_se(oix, reads, mode=0, threads=8)
res, cig, off = ctx.map_se(reads, mode=0)
bad = [i for i in range(len(reads)) if int(res[i]["pos"]) != int(o_res[i]["pos"])]
print("bad", bad)
names, starts, gw = None, None, None
import bench
names, starts, gw = bench.read_index_genome(idx)
def gnib(k): return int((int(gw[k >> 4]) >> ((k & 15) * 4)) & 15)
dec = "ZACMGRSVTWYHKDBN"
for i in bad[:4]:
    h = o_res[i]; print(i, "oracle", int(h["diffs"]), hex(int(h["flags"])), int(h["pos"]), [hex(x) for x in o_cig[i, :int(o_n[i])]], "gpu", int(res[i]["diffs"]), hex(int(res[i]["flags"])), int(res[i]["pos"]))
    p = int(h["pos"]); print("  read  ", reads[i]); print("  genome", "".join(dec[gnib(p + k)] for k in range(-2, 104)))
    # single-read rerun
    r1, c1, o1 = ctx.map_se([reads[i]], mode=0); print("  gpu alone:", int(r1[0]["diffs"]), hex(int(r1[0]["flags"])), int(r1[0]["pos"]))
print("---- experiments")
for i in bad:
    for vf in (0.1, 0.2, 0.5):
        r1, c1, o1 = ctx.map_se([reads[i]], mode=0, params=A.Params(valid_frac=vf))
        oo, oc, on, _ = o.map_se(oix, [reads[i]], mode=0, valid_frac=vf)
        print(i, "len", len(reads[i]), "vf", vf, "gpu", int(r1[0]["diffs"]), hex(int(r1[0]["flags"])), int(r1[0]["pos"]), [hex(x) for x in c1[:4]], "| oracle", int(oo[0]["diffs"]), hex(int(oo[0]["flags"])), int(oo[0]["pos"]), [hex(x) for x in oc[0,:int(on[0])]][:4])
print("---- sweep")
def rn(c, arich=False):
    return {"A": 5 if arich else 1, "C": 2, "G": 4, "T": 8 if arich else 10}.get(c, 0)
comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
for i in (900, 3218, 3866):
    h = o_res[i]; p = int(h["pos"]); fl = int(h["flags"]); r = reads[i]
    if fl & 0x10:
        r = "".join(comp.get(c, "N") for c in reversed(r)); ar = True
    else:
        ar = False
    d = sum(1 - bin(rn(c, ar) & gnib(p + k)).count("1") for k, c in enumerate(r))
    print(i, "hamming at oracle pos:", d, "read non-ACGT:", sum(1 for c in reads[i] if c not in "ACGT"))
    for vf in (0.1, 0.12, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45):
        r1, c1, o1 = ctx.map_se([reads[i]], mode=0, params=A.Params(valid_frac=vf))
        oo, oc, on, _ = o.map_se(oix, [reads[i]], mode=0, valid_frac=vf)
        print("   vf", vf, "gpu", int(r1[0]["diffs"]), int(r1[0]["pos"]), "oracle", int(oo[0]["diffs"]), int(oo[0]["pos"]))
