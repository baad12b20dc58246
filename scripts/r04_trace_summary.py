#!/usr/bin/env python3
"""Summary of an ABM_CLI_TRACE=1 log of `abismal-amd map`: per kind of event the first / last time and a histogram over
the run (events per 50 ms), and the durations the events carry (parse, format, write: microseconds)."""
import collections
import re
import sys

ev = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"\[abm cli\] t=\s*([\d.]+) ms (.+?)\s+(\d+) (\d+)\s*$", ln)
    if m:
        ev[m.group(2).strip()].append((float(m.group(1)), int(m.group(3)), int(m.group(4))))
if not ev:
    sys.exit("no events")
end = max(t for v in ev.values() for t, _, _ in v)
step = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
nb = int(end // step) + 1
print(f"run: {end:.1f} ms; buckets of {step:g} ms")
for kind in ("cut", "parsed", "batch formed", "batch ready", "batch mapped", "formatted", "written", "waited on join", "threads joined", "output closed"):
    v = ev.get(kind)
    if not v:
        continue
    hist = [0] * nb
    for t, _, _ in v:
        hist[int(t // step)] += 1
    extra = ""
    if kind in ("parsed", "formatted", "written"):
        us = sorted(b for _, _, b in v)
        extra = f"  task us: median {us[len(us) // 2]}, p90 {us[len(us) * 9 // 10]}, max {us[-1]}, sum {sum(us) / 1e6:.2f} s"
    if kind in ("batch formed",):
        extra = "  sizes: " + " ".join(str(b) for _, _, b in v[:40])
    print(f"{kind:15s} n={len(v):5d} first {v[0][0]:8.1f} last {v[-1][0]:8.1f}{extra}")
    print("                " + " ".join(f"{h:3d}" for h in hist))
