#!/usr/bin/env python3
"""Instruction mix of the loops of a gfx950 kernel, from the compiled object (no GPU needed):

    python tools/isa_loops.py strique_amd/lib/obj/viterbi_kernels.o 'viterbi_kernelILi4ELi2ELi65ELi13ELi2ELb0ELb1ELb0ELb0E'

Unbundles the device code object (llvm-objdump --offloading), disassembles it and prints, for every backward
branch of the kernels whose mangled name contains the pattern, the instructions between target and branch by class."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    local = os.path.join(tmp, os.path.basename(obj))
    subprocess.check_call(["cp", obj, local])
    subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
    return subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, co)]).decode().split("\n")


def kernels(lines):
    out, cur = {}, None
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", l)
        if m:
            cur = m.group(1); out[cur] = []
        elif cur:
            m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-F]+):", l)
            if m:
                out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def loops(ins):
    res = []
    for ad, op, args in ins:
        if op.startswith("s_cbranch") or op == "s_branch":
            try:
                off = int(args.split()[0])
            except (ValueError, IndexError):
                continue
            if off > 32767:
                off -= 65536
            tgt = ad + 4 + off * 4
            if tgt < ad:
                res.append([x for x in ins if tgt <= x[0] <= ad])
    return res


def classify(op):
    if op.startswith("v_") and "f64" in op:
        return "valu_f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_"):
        return "salu"
    return "vmem"


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    for name, ins in kernels(disassemble(obj)).items():
        if pat not in name:
            continue
        print(name[:110], "instructions:", len(ins))
        for sub in sorted(loops(ins), key=len):
            cls = collections.Counter(classify(x[1]) for x in sub)
            ops = collections.Counter(x[1] for x in sub)
            print("  loop of %4d: %s | %s" % (len(sub), dict(cls), ", ".join("%s %d" % kv for kv in ops.most_common(top))))


if __name__ == "__main__":
    main()
