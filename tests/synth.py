"""Small synthetic genomes/reads for parity tests (numpy, deterministic)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


def repeat_rich_genome(path, seed=3, n_chroms=3, chrom_len=1_500_000, iupac=0):
    """Repeat families at low divergence, purine-only tracts, homopolymers and N
    runs (short ones get LCG-filled by the indexer, long ones are excluded): the
    inputs that drive bucket narrowing, full candidate heaps and tie handling."""
    rng = np.random.default_rng(seed)
    fams = [ACGT[rng.integers(0, 4, ln)] for ln in (320, 900, 2500)]
    with open(path, "wb") as f:
        for c in range(n_chroms):
            seq = ACGT[rng.integers(0, 4, chrom_len)].copy()
            for fam, copies, div in zip(fams, (600, 250, 80), (0.03, 0.02, 0.05)):
                for at in rng.integers(0, chrom_len - len(fam), copies):
                    piece = fam.copy()
                    m = rng.random(len(fam)) < div
                    piece[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
                    if rng.random() < 0.5:
                        piece = COMP[piece[::-1]]
                    seq[at:at + len(fam)] = piece
            for at in rng.integers(0, chrom_len - 500, 150):  # A/G-only and C/T-only tracts
                alpha = np.frombuffer(b"AG" if rng.random() < 0.5 else b"CT", dtype=np.uint8)
                seq[at:at + 400] = alpha[rng.integers(0, 2, 400)]
            for at in rng.integers(0, chrom_len - 300, 40):   # homopolymers / dinucleotide repeats
                motif = ACGT[rng.integers(0, 4, int(rng.integers(1, 3)))]
                seq[at:at + 200] = np.resize(motif, 200)
            if iupac:                                          # ambiguity codes: multi-bit genome nibbles
                at = rng.integers(0, chrom_len, iupac)
                seq[at] = np.frombuffer(b"RYMKSWBDHV", dtype=np.uint8)[rng.integers(0, 10, iupac)]
            seq[1000:1100] = ord("N")                          # short run (<=256): filled
            seq[chrom_len // 2: chrom_len // 2 + 3000] = ord("N")  # long run: excluded
            if c == 0:
                seq[:50] = ord("N")
            if c % 2 == 1:                                     # some soft-masked sequence
                seq[5000:9000] = np.char.lower(seq[5000:9000].view("S1")).view(np.uint8)
            f.write(b">chr%d some description\n" % (c + 1))
            f.write(b"\n".join(bytes(seq[i:i + 70]) for i in range(0, chrom_len, 70)) + b"\n")


def mutated_reads(fasta, n, L, seed, mut=0.02, bis=0.95, pbat_frac=0.0, n_frac=0.02):
    """Reads drawn from a FASTA with substitutions/indels and bisulfite conversion;
    a fraction carry N bases (leading/trailing/internal) and a few are too short."""
    rng = np.random.default_rng(seed)
    chroms = []
    for rec in open(fasta, "rb").read().split(b">")[1:]:
        chroms.append(np.frombuffer(rec.split(b"\n", 1)[1].replace(b"\n", b"").upper(), dtype=np.uint8))
    reads = []
    for _ in range(n):
        ch = chroms[int(rng.integers(0, len(chroms)))]
        ln = L if rng.random() > 0.05 else int(rng.integers(30, L))
        at = int(rng.integers(0, len(ch) - ln - 20))
        frag = ch[at:at + ln + 20].copy()
        if rng.random() < 0.5:
            frag = COMP[frag[::-1]]
        out = []
        i = 0
        while len(out) < ln and i < len(frag):
            r = rng.random()
            if r < mut / 3:
                out.append(ACGT[rng.integers(0, 4)]); i += 1
            elif r < 2 * mut / 3:
                out.append(ACGT[rng.integers(0, 4)])
            elif r < mut:
                i += 1
            else:
                out.append(frag[i]); i += 1
        s = np.array(out[:ln], dtype=np.uint8)
        ga = rng.random() < pbat_frac
        src, dst = (ord("G"), ord("A")) if ga else (ord("C"), ord("T"))
        conv = (s == src) & (rng.random(len(s)) < bis)
        s[conv] = dst
        if rng.random() < n_frac:
            k = int(rng.integers(1, 8))
            where = rng.random()
            if where < 0.33:
                s[:k] = ord("N")
            elif where < 0.66:
                s[-k:] = ord("N")
            else:
                j = int(rng.integers(0, len(s) - k)); s[j:j + k] = ord("N")
        reads.append(bytes(s))
    return reads


def trim_like_readloader(reads):
    """ReadLoader::load_reads trimming/skip rule (src/abismal.cpp:187-195) on raw sequences."""
    out = []
    for r in reads:
        if isinstance(r, bytes):
            r = r.decode()
        if sum(1 for c in r if c != "N") < 44:
            out.append("")
            continue
        r = r.rstrip("N")
        first = min(i for i in (r.find(b) for b in "ACGT") if i >= 0)
        out.append(r[first:])
    return out


def mutated_pairs(fasta, n, L, seed, mut=0.02, bis=0.95, frag=(120, 600)):
    """Read pairs from random fragments: read 1 = fragment start, read 2 = revcomp of its end;
    conversion C->T on the fragment strand (so read 2 looks G->A), with mutations and some Ns."""
    rng = np.random.default_rng(seed)
    chroms = []
    for rec in open(fasta, "rb").read().split(b">")[1:]:
        chroms.append(np.frombuffer(rec.split(b"\n", 1)[1].replace(b"\n", b"").upper(), dtype=np.uint8))
    out1, out2 = [], []
    for _ in range(n):
        ch = chroms[int(rng.integers(0, len(chroms)))]
        fl = int(rng.integers(max(frag[0], L), frag[1]))
        at = int(rng.integers(0, len(ch) - fl))
        f = ch[at:at + fl].copy()
        if rng.random() < 0.5:
            f = COMP[f[::-1]]
        m = rng.random(fl) < mut
        f[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
        if rng.random() < 0.2:  # a small deletion or insertion
            j = int(rng.integers(10, fl - 10))
            f = np.concatenate([f[:j], f[j + 2:]]) if rng.random() < 0.5 else np.concatenate([f[:j], ACGT[rng.integers(0, 4, 2)], f[j:]])
        conv = (f == ord("C")) & (rng.random(len(f)) < bis)
        f[conv] = ord("T")
        a = f[:L].copy()
        b = COMP[f[::-1]][:L].copy()
        if rng.random() < 0.03:
            a[:3] = ord("N")
        if rng.random() < 0.03:
            b[-4:] = ord("N")
        if rng.random() < 0.02:
            a = a[:30]
        out1.append(bytes(a))
        out2.append(bytes(b))
    return out1, out2
