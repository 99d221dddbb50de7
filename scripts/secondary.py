#!/usr/bin/env python3
"""The two secondary workloads of bench.py (BASELINE configs[3] sparse and configs[4] k = 64) on their own, for
`rocprofv3 --kernel-trace --stats -- python3 scripts/secondary.py` (profiles/r05/secondary_kernel_stats.csv)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import nmfk_jl_amd as NMFk

ctx = NMFk.Context(0)
which = sys.argv[1:] or ["cfg4", "cfg5"]
out = {}
if "cfg4" in which:
    out["cfg4"] = bench.secondary_cfg4(NMFk, ctx)
if "cfg5" in which:
    out["cfg5"] = bench.secondary_cfg5(NMFk, ctx)
print(json.dumps(out))
