"""Builds the in-tree native libraries with hipcc / gcc (no JIT cache: the .so files
travel to the GPU box with the repo snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.environ.get("APEMOST_HIP_LIB") or os.path.join(HERE, "libapemost_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
             "-Wno-unused-value"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, "apemost_hip.hip")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "apemost_hip.h")]   # the one translation unit and everything it includes
    if force or _stale(HIP_LIB, srcs):
        cmd = [HIPCC] + HIP_FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", HIP_LIB,
                                     srcs[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_stamps(verbose=False, wave=0):
    """diagnostic twin of the library with in-kernel s_memtime stamps (tools/stamp_profile.py);
    `wave` selects the wave of workgroup 0 whose step segments are timed, -1 = a timeline of all waves"""
    out = os.path.join(HERE, "libapemost_hip_stamps%s.so" % ("" if wave == 0 else "_tl" if wave < 0 else "_w%d" % wave))
    cmd = [HIPCC] + HIP_FLAGS + ["-DAPEMOST_STAMPS", "-DAPEMOST_STAMP_WAVE=%d" % wave, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out,
                                 os.path.join(CSRC, "apemost_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_dev_stamps(models, waves, verbose=False):
    """development build of the stamps twin (in-kernel cycle counters)"""
    return build_dev(models, waves, out=os.path.join(HERE, "libapemost_hip_stamps.so"), extra=["-DAPEMOST_STAMPS"],
                     verbose=verbose)


def build_dev(models, waves, out=None, extra=(), verbose=False):
    """development build: only the given models (ids) and workgroup shapes (wave counts) are
    instantiated, which compiles in seconds instead of minutes.  Never the product library:
    select it with APEMOST_HIP_LIB=<out>."""
    out = out or os.path.join(HERE, "libapemost_hip_dev.so")
    mm = sum(1 << m for m in models)
    wm = sum(1 << w for w in waves)
    cmd = [HIPCC] + HIP_FLAGS + ["-DAPEMOST_DEV_MODELS=%d" % mm, "-DAPEMOST_DEV_WAVES=%d" % wm] + list(extra) + [
        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out, os.path.join(CSRC, "apemost_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    return [build_hip(force, verbose)]


if __name__ == "__main__":
    for p in build_all(force=True, verbose=True):
        print("built", p)
