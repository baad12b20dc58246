#!/usr/bin/env python3
"""Static instruction census of one kernel from hipcc's assembly (-S -gline-tables-only).

Usage: isa_profile.py file.s kernel-symbol-substring [--lines]

Attributes every instruction of the kernel to the source line of the innermost `.loc` in force and
sums by the enclosing source function (found by scanning the source files for function headers), split
into VALU / SALU / VMEM / LDS / scratch.  Static counts, not executed ones: useful to see what a loop
body is made of and to compare two builds of the same source.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("scratch_") or (op.startswith("buffer_") and "offen" in op):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    return "other"


def function_ranges(path):
    """[(first_line, name)] of the device functions / lambdas of a source file, by a crude header scan"""
    out = []
    try:
        src = open(path).read().split("\n")
    except OSError:
        return out
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|static|inline|__host__)[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(")
    lam = re.compile(r"^\s*auto\s+([A-Za-z_][A-Za-z0-9_]*)\s*=\s*\[")
    for n, line in enumerate(src, 1):
        m = pat.match(line)
        if m and m.group(1) not in ("if", "for", "while", "return", "__launch_bounds__"):
            out.append((n, m.group(1)))
            continue
        m = lam.match(line)
        if m:
            out.append((n, "λ" + m.group(1)))
    return out


def main():
    path, want = sys.argv[1], sys.argv[2]
    by_line = "--lines" in sys.argv
    files = {}
    counts = collections.defaultdict(lambda: collections.Counter())
    inside = False
    cur = (0, 0)
    file_re = re.compile(r'^\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"')
    file_re1 = re.compile(r'^\s*\.file\s+(\d+)\s+"([^"]*)"')
    loc_re = re.compile(r"^\s*\.loc\s+(\d+)\s+(\d+)")
    for line in open(path):
        m = file_re.match(line)
        if m:
            files[int(m.group(1))] = m.group(2).rstrip("/") + "/" + m.group(3)
            continue
        m = file_re1.match(line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if not inside:
            if line.startswith("_Z") and want in line and line.rstrip().split(":")[0].endswith(("E", "_")) is not None and ":" in line:
                inside = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = loc_re.match(line)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        s = line.strip()
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        op = s.split()[0]
        counts[cur][classify(op)] += 1
    ranges = {fid: function_ranges(p) for fid, p in files.items() if "/root/repo" in p}
    agg = collections.defaultdict(lambda: collections.Counter())
    for (fid, ln), c in counts.items():
        name = "?"
        if fid in ranges:
            for first, fn in ranges[fid]:
                if first <= ln:
                    name = fn
                else:
                    break
            key = f"{files[fid].split('/')[-1]}:{name}" if not by_line else f"{files[fid].split('/')[-1]}:{ln} ({name})"
        else:
            key = files.get(fid, "?").split("/")[-1]
        agg[key].update(c)
    total = collections.Counter()
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))
    print(f"{'where':58s} {'valu':>6s} {'salu':>6s} {'vmem':>5s} {'lds':>5s} {'scr':>5s}")
    for key, c in rows:
        total.update(c)
        if sum(c.values()) >= (3 if by_line else 1):
            print(f"{key:58s} {c['valu']:6d} {c['salu']:6d} {c['vmem']:5d} {c['lds']:5d} {c['scratch']:5d}")
    print(f"{'TOTAL':58s} {total['valu']:6d} {total['salu']:6d} {total['vmem']:5d} {total['lds']:5d} {total['scratch']:5d}")


if __name__ == "__main__":
    main()
