#!/usr/bin/env python3
"""Loops of a kernel in the gfx950 code object of a built library or object file: for every backward
branch, the number of VALU / SALU / LDS / VMEM instructions and waits between its target and the
branch, and how many v_rcp_f64 (one per data point or pair of points in the pulse models).  What the
instruction counts per data point in DESIGN.md 16 were read from.  No GPU needed.

    python tools/loop_insts.py <model_N.o | lib.so> '<substring of the demangled kernel name>'
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(path):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.o")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co])
    return subprocess.run(["c++filt"], input=txt, stdout=subprocess.PIPE).stdout.decode()


def kernels(txt):
    cur, body = None, {}
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur and line.strip():
            body[cur].append(line)
    return body


def classify(op, c):
    if op.startswith("v_"):
        c["valu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        c["vmem"] += 1
    elif op.startswith("s_waitcnt"):
        c["wait"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    if op.startswith("v_rcp_f64"):
        c["rcp"] += 1


def main():
    path, want = sys.argv[1], sys.argv[2]
    for name, lines in kernels(disassemble(path)).items():
        if want not in name:
            continue
        insts = []
        for line in lines:
            m = re.match(r"^\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if m:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
        index = {a: i for i, (a, _, _) in enumerate(insts)}
        print(name, len(insts), "instructions")
        for i, (addr, op, args) in enumerate(insts):
            if not (op.startswith("s_cbranch") or op == "s_branch"):
                continue
            m = re.match(r"(\d+)", args)            # the branch offset in dwords, as objdump prints it
            if not m:
                continue
            simm = int(m.group(1))
            if simm >= 32768:
                simm -= 65536
            target = addr + 4 + 4 * simm
            if target >= addr or target not in index:
                continue
            c = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "rcp": 0, "wait": 0}
            for _, o, _ in insts[index[target]:i + 1]:
                classify(o, c)
            if c["valu"] >= 20:
                print("  loop %06x..%06x: %s" % (target, addr, c))


if __name__ == "__main__":
    main()
