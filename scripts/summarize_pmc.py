#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc csv output (one row per dispatch and counter) into per-kernel means."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for sub in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(sub):
        continue
    files = glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?")
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value", 0) or 0)
                a = agg[short][c]
                a[0] += v
                a[1] += 1
    print(f"== {os.path.basename(sub)}")
    for kname, ctrs in sorted(agg.items()):
        n = max(v[1] for v in ctrs.values())
        print(f"  {kname}  dispatches={n}")
        for c, (tot, cnt) in sorted(ctrs.items()):
            print(f"      {c:28s} mean={tot / max(cnt, 1):.6g} total={tot:.6g}")
