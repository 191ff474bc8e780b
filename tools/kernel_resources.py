#!/usr/bin/env python3
"""Registers, spills, LDS and scratch per kernel, read from the gfx950 code object inside a built
library (default: apemost_amd/libapemost_hip.so).  No GPU needed.

    python tools/kernel_resources.py [lib.so | object.o] [substring of the demangled name]

Occupancy bound by registers on gfx950: 512 VGPRs per SIMD lane slot -> waves/SIMD =
floor(512 / vgprs rounded up to 8), at most 8."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(lib):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "co.o")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co]).decode()
    out, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur.get("name"):
            out.append(cur)
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "agpr_count",
                 "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            if k == "name" and "name" in cur and "vgpr_count" in cur:
                out.append(cur)
                cur = {}
            cur[k] = v
    if cur.get("name"):
        out.append(cur)
    names = [k["name"] for k in out]
    dem = subprocess.run(["c++filt"], input="\n".join(names).encode(), stdout=subprocess.PIPE).stdout
    for k, d in zip(out, dem.decode().splitlines()):
        k["demangled"] = d
    return out


def main():
    args = [a for a in sys.argv[1:]]
    # the product library is linked from one object per translation unit (apemost_amd/build.py), each
    # with a code object of its own: read them object by object
    obj_dir = os.path.join(ROOT, "apemost_amd", "csrc", "obj")
    libs = sorted(os.path.join(obj_dir, f) for f in os.listdir(obj_dir) if f.endswith(".o")) if os.path.isdir(obj_dir) else []
    if args and (args[0].endswith(".so") or args[0].endswith(".o")):
        libs = [args.pop(0)]
    needle = args[0] if args else ""
    print("%-64s %5s %5s %5s %6s %6s %7s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "waves/SIMD"))
    for k in [k for lib in libs for k in kernels(lib)]:
        if needle not in k["demangled"]:
            continue
        v = int(k.get("vgpr_count", 0))
        # unified register file: arch VGPRs + AGPRs, allocated in blocks of 8
        total = (v + 7) // 8 * 8
        occ = min(8, 512 // max(total, 1))
        name = re.sub(r"^void ", "", k["demangled"])
        name = re.sub(r"\(.*$", "", name)
        print("%-64s %5s %5s %5s %6s %6s %7s %8d" % (name[:64], k.get("vgpr_count"), k.get("agpr_count", "-"),
                                                     k.get("sgpr_count"), k.get("vgpr_spill_count"),
                                                     k.get("sgpr_spill_count"), k.get("private_segment_fixed_size"), occ))


if __name__ == "__main__":
    main()
