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


def _sources_sha1(paths):
    import hashlib
    h = hashlib.sha1()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


LAST_BUILD = {"compiled": [], "linked": False}   # what the last build_hip() call did (reported by __graft_entry__.build)


OBJ = os.path.join(CSRC, "obj")
MODEL_TUS = (0, 1, 2, 3, 8, 9, 10, 11)   # the built-in models and their variant instantiations (pt_device.h kVariantModel)


def _sources():
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [
        os.path.join(ROOT, "include", "apemost_hip.h")]
    return os.path.join(CSRC, "apemost_hip.hip"), os.path.join(CSRC, "apemost_model.hip"), headers


def build_hip(force=False, verbose=False):
    """libapemost_hip.so from nine translation units compiled side by side: the host side of the C ABI
    (apemost_hip.hip) and one unit per likelihood model holding all of that model's kernels
    (apemost_model.hip with -DAPEMOST_TU_MODEL=k).  Only what is older than its sources is rebuilt."""
    from concurrent.futures import ThreadPoolExecutor
    abi_src, model_src, headers = _sources()
    os.makedirs(OBJ, exist_ok=True)
    # The objects and the library travel with a snapshot of the tree (gpurun), where file times need not mean
    # what they meant here: the hash of the sources the library was built from is kept beside it, and a library
    # whose stamp differs from the sources present is rebuilt whatever the file times say.
    stamp = HIP_LIB + ".sources_sha1"
    want = _sources_sha1([abi_src, model_src] + headers)
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if have != want and os.path.exists(HIP_LIB):
        force = True
    compile_flags = [f for f in HIP_FLAGS if f != "-shared"] + ["-c", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    jobs = []
    abi_obj = os.path.join(OBJ, "abi.o")
    if force or _stale(abi_obj, [abi_src] + headers):
        jobs.append([HIPCC] + compile_flags + ["-o", abi_obj, abi_src])
    objs = [abi_obj]
    for k in MODEL_TUS:
        obj = os.path.join(OBJ, "model_%d.o" % k)
        objs.append(obj)
        if force or _stale(obj, [model_src] + headers):
            jobs.append([HIPCC] + compile_flags + ["-DAPEMOST_TU_MODEL=%d" % k, "-o", obj, model_src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        workers = max(1, min(len(jobs), os.cpu_count() or 1, int(os.environ.get("APEMOST_BUILD_JOBS", "8"))))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(run, jobs))
    link = bool(jobs) or force or _stale(HIP_LIB, objs)
    if link:
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs + ["-ldl"])
    if link or have != want:
        open(stamp, "w").write(want + "\n")
    LAST_BUILD["compiled"] = [os.path.basename(j[j.index("-o") + 1]) for j in jobs]
    LAST_BUILD["linked"] = link
    return HIP_LIB


def build_stamps(verbose=False, wave=0):
    """diagnostic twin of the library with in-kernel s_memtime stamps (tools/stamp_profile.py);
    `wave` selects the wave of workgroup 0 whose step segments are timed, -1 = a timeline of all waves"""
    out = os.path.join(HERE, "libapemost_hip_stamps%s.so" % ("" if wave == 0 else "_tl" if wave < 0 else "_w%d" % wave))
    cmd = [HIPCC] + HIP_FLAGS + ["-DAPEMOST_SINGLE_TU", "-DAPEMOST_STAMPS", "-DAPEMOST_STAMP_WAVE=%d" % wave, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out,
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
    cmd = [HIPCC] + HIP_FLAGS + ["-DAPEMOST_SINGLE_TU", "-DAPEMOST_DEV_MODELS=%d" % mm, "-DAPEMOST_DEV_WAVES=%d" % wm] + list(extra) + [
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
