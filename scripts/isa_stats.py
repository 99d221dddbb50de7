#!/usr/bin/env python3
"""Instruction statistics of the kernels inside libnmfk_hip.so whose mangled name contains a pattern; with a second argument the
kernel's disassembly is written to that file.  `python scripts/isa_stats.py wide2_step_kernelILi4ELi2ELi0ELb1 /tmp/k.s`"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_lint_pk_opsel import LLVM, code_objects
so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfk.jl_amd", "libnmfk_hip.so")
pat = sys.argv[1]
with tempfile.TemporaryDirectory() as tmp:
    for co in code_objects(so, tmp):
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
        for part in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
            m = re.match(r"[0-9a-f]{16} <(\S+)>", part)
            if not m or pat not in m.group(1):
                continue
            c = lambda p: len(re.findall(p, part))
            print(m.group(1)[:80], "mfma", c(r"v_mfma"), "mov", c(r"v_mov_b32"), "perm", c("v_perm_b32"), "and", c("v_and_b32"), "sub", c("v_sub_f32"), "rcp", c("v_rcp"),
                  "ds_r128", c("ds_read_b128"), "ds_r64", c(r"ds_read2?_b64"), "ds_w16", c("ds_write_b16"), "ds_w32", c("ds_write_b32"), "nop", c("s_nop"),
                  "waitcnt", c("s_waitcnt"), "lines", part.count("\n"))
            if len(sys.argv) > 2:
                open(sys.argv[2], "w").write(part)
