#!/usr/bin/env python3
"""Registers, spills and LDS of the kernels inside libnmfk_hip.so whose (mangled) name contains a pattern:
`python scripts/kernel_resources.py sp_wblk` (reads the code objects' metadata notes)."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_lint_pk_opsel import LLVM, code_objects

so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfk.jl_amd", "libnmfk_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    for co in code_objects(so, tmp):
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or pat not in name.group(1):
                continue
            g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]
            print(f"{name.group(1)[:90]:90s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} "
                  f"lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
