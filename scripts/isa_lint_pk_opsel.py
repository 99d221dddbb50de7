#!/usr/bin/env python3
"""ISA lint for the gfx950 hazard of DESIGN.md ("Known hazard"): a packed fp32 VALU instruction (v_pk_fma_f32,
v_pk_mul_f32, v_pk_add_f32) whose src1 is a VGPR pair read with op_sel[1] = 1 -- the low result half takes the ODD
register of src1 -- returns wrong low halves in the lanes 48-63 while another wave on the CU issues 128-bit-operand
matrix instructions (tools/hazard/pk_victim.hip G=1, G=4; tools/hazard/burner.hip mode 0).  The same select on src0, src2 or on an
SGPR pair, and op_sel_hi, are safe.  This script compiles every kernel TU of libnmfk_hip.so to gfx950 assembly and lists
the kernels that contain the unsafe form.  Exit code 1 if any does.
usage: isa_lint_pk_opsel.py [--tu file.hip ...] [extra hipcc flags]
       isa_lint_pk_opsel.py --so libnmfk_hip.so      the SHIPPED code: every gfx950 code object embedded in the built
                                                     library is disassembled (llvm-objdump) and scanned; the Makefile runs
                                                     this after linking and removes the library on a hit"""
import os, re, struct, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmfk.jl_amd", "csrc")
TUS = ["nmfk_step_f32.hip", "nmfk_step_f64.hip", "nmfk_step_hyb.hip", "nmfk_cluster.hip", "nmfk_kmeans.hip", "nmfk_api.hip", "nmfk_comm.hip"]
VF = {"nmfk_step_f32.hip", "nmfk_step_hyb.hip"}  # (Makefile: -mllvm -amdgpu-mfma-vgpr-form)
PK = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)$")


def unsafe(line):
    m = PK.match(line)
    if not m:
        return False
    ops = m.group(2)
    sel = re.search(r"\bop_sel:\[([01,]+)\]", ops)
    if not sel:
        return False
    bits = sel.group(1).split(",")
    if len(bits) < 2 or bits[1] != "1":
        return False
    args = [a.strip() for a in re.split(r",\s*(?![^\[]*\])", ops.split(" op_sel")[0])]
    return len(args) >= 3 and re.match(r"v(\[|\d)", args[2]) is not None  # dst, src0, src1, ...: a VGPR (not vcc)


def scan(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur and unsafe(line):
            out.setdefault(cur, []).append(line.strip())
    return out


LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so, tmp):
    """the gfx950 code objects inside a host ELF: its .hip_fatbin section is a sequence of clang offload bundles
    (magic, number of entries, then (offset, size, length of the target id, target id) per entry)"""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "copy.so")],
                   check=True)
    blob = open(fat, "rb").read()
    magic, out = b"__CLANG_OFFLOAD_BUNDLE__", []
    for mm in re.finditer(magic, blob):
        base = mm.start()
        p = base + len(magic)
        (count,) = struct.unpack_from("<Q", blob, p)
        p += 8
        for _ in range(count):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                path = os.path.join(tmp, f"co{len(out)}.co")
                open(path, "wb").write(blob[base + off:base + off + size])
                out.append(path)
    return out


def scan_disassembly(text):
    """llvm-objdump -d output: `<symbol>:` labels and one instruction per line, followed by `// address: encoding`"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            continue
        ins = line.split("//")[0]
        if cur and unsafe(ins):
            out.setdefault(cur, []).append(ins.strip())
    return out


def lint_so(so):
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        cos = code_objects(so, tmp)
        if not cos:
            print(f"{so}: no gfx950 code object found")
            return 1
        npk = 0
        for co in cos:
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True,
                                  text=True).stdout
            npk += len(re.findall(r"\bv_pk_(?:fma|mul|add)_f32\b", text))
            for k, v in scan_disassembly(text).items():
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                print(f"    {name[:110]}: {len(v)}   e.g. {v[0]}")
                bad += len(v)
        print(f"{os.path.basename(so)}: {bad} unsafe packed instruction(s) among {npk} packed fp32 instructions in {len(cos)} code objects")
    return 1 if bad else 0


def compile_tu(tu, extra, tmp):
    s = os.path.join(tmp, tu + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-S"] + extra
    if tu in VF:
        cmd += ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    subprocess.run(cmd + [os.path.join(CSRC, tu), "-o", s], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return tu, scan(s)


def main():
    extra, tus = [], []
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--so":
            return lint_so(args.pop(0))
        if a == "--tu":
            tus.append(args.pop(0))
        else:
            extra.append(a)
    tus = tus or TUS
    bad = 0
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(4) as ex:
        for tu, hits in ex.map(lambda t: compile_tu(t, extra, tmp), tus):
            n = sum(len(v) for v in hits.values())
            print(f"{tu}: {n} unsafe packed instruction(s) in {len(hits)} kernel(s)")
            for k, v in hits.items():
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                print(f"    {name[:110]}: {len(v)}   e.g. {v[0]}")
            bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
