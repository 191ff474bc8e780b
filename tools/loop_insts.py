#!/usr/bin/env python3
"""Innermost loops of a kernel in a disassembly (llvm-objdump -d --no-show-raw-insn of the gfx950 code
object): for every backward branch, the number of VALU / SALU / LDS / VMEM instructions between the
target and the branch, and how many v_rcp_f64 (one per data point in the pulse models; per sine: v_rndne).

    python tools/loop_insts.py <model_N.o | lib.so> '<substring of the demangled kernel name>'
"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"

def disassemble(path):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.o")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co]).decode()
    return subprocess.run(["c++filt"], input=txt.encode(), stdout=subprocess.PIPE).stdout.decode()

def main():
    path, want = sys.argv[1], sys.argv[2]
    txt = disassemble(path)
    cur, body = None, {}
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if cur and line.strip():
            body[cur].append(line)
    for name, lines in body.items():
        if want not in name or name.startswith("L") or not lines:
            continue
        insts = []
        for l in lines:
            m = re.match(r"^\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
            if m:
                insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
        addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
        print(name, len(insts), "instructions")
        for i, (a, op, args) in enumerate(insts):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", args)
                lab = None
                # objdump prints the target as a symbol+offset or an absolute address
                m2 = re.search(r"(\d+)\s*$", args)
                tgt = None
                for t in re.findall(r"0x([0-9a-f]+)", l if False else args):
                    pass
                # compute from the encoded simm16 when shown as a plain number
                if m2 and not m:
                    simm = int(m2.group(1))
                    if simm >= 32768:
                        simm -= 65536
                    tgt = a + 4 + 4 * simm
                if tgt is None or tgt >= a or tgt not in addr_index:
                    continue
                j = addr_index[tgt]
                seg = insts[j:i + 1]
                inner = not any((o.startswith("s_cbranch") or o == "s_branch") and k < len(seg) - 1 and False for k, (_, o, _) in enumerate(seg))
                c = {"valu": 0, "salu": 0, "lds": 0, "vmem": 0, "rcp": 0, "wait": 0}
                for _, o, _ in seg:
                    if o.startswith("v_"):
                        c["valu"] += 1
                    elif o.startswith("ds_"):
                        c["lds"] += 1
                    elif o.startswith(("global_", "buffer_", "flat_", "scratch_")):
                        c["vmem"] += 1
                    elif o.startswith("s_waitcnt"):
                        c["wait"] += 1
                    elif o.startswith("s_"):
                        c["salu"] += 1
                    if o == "v_rcp_f64_e32" or o == "v_rcp_f64_e64":
                        c["rcp"] += 1
                if c["valu"] >= 20:
                    print("  loop %06x..%06x: %s" % (tgt, a, c))

main()
