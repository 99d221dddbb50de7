#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel of libnmfk_hip.so: `python scripts/loop_mix.py <mangled-name-part> [min_instrs]`.
Prints every backward branch's body size and opcode histogram (how the VALU budget of a loop is spent)."""
import collections, os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_lint_pk_opsel import LLVM, code_objects

so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmfk.jl_amd", "libnmfk_hip.so")
pat, least = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
with tempfile.TemporaryDirectory() as tmp:
    for co in code_objects(so, tmp):
        t = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", t, re.S | re.M):
            if pat not in m.group(1):
                continue
            ins = [(int(l.split("//")[1].split(":")[0], 16), l.split("//")[0].strip()) for l in m.group(2).splitlines() if "//" in l]
            amap = {a: i for i, (a, _) in enumerate(ins)}
            print(m.group(1)[:100], len(ins), "instructions")
            for i, (a, s) in enumerate(ins):
                mm = re.match(r"s_c?branch\S*\s+(?:\S+,\s*)?(\d+)$", s)
                if not mm:
                    continue
                off = int(mm.group(1))
                off -= 65536 if off > 32767 else 0
                tgt = a + 4 + 4 * off
                if tgt < a and tgt in amap and i + 1 - amap[tgt] >= least:
                    body = ins[amap[tgt]:i + 1]
                    c = collections.Counter(x.split()[0] for _, x in body)
                    print(f"  loop of {len(body)}: " + ", ".join(f"{k} {v}" for k, v in c.most_common(24)))
