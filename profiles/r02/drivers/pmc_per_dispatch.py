#!/usr/bin/env python3
"""Prints per-dispatch PMC values (no averaging) for kernels matching a pattern: usage pmc_per_dispatch.py DIR PATTERN [N]"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 8
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = collections.OrderedDict()
with open(f) as fh:
    for r in csv.DictReader(fh):
        if pat not in r["Kernel_Name"]:
            continue
        rows.setdefault(r["Dispatch_Id"], {"grid": r.get("Grid_Size", "?")})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (k, v) in enumerate(rows.items()):
    if i >= nmax:
        break
    print(k, " ".join(f"{a}={b:.4g}" if a != "grid" else f"grid={b}" for a, b in v.items()))
