"""Compile check for a user-supplied device likelihood (include/apemost_device_model.h) without a GPU:
the same translation unit the engine hands to hiprtc when a sampler of APEMOST_MODEL_USER is created
(apemost_hip.hip user_model_build) -- '#include "pt_kernels.h"' + the user's file, the four kernels of
each workgroup shape (1, 2, 4, 8 waves per chain) named as template instantiations -- compiled for gfx950 through libhiprtc with ctypes.

    python -m apemost_amd.device_model my_model.hip
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODEL_USER, VARIANT = 4, 8   # APEMOST_MODEL_USER, pt_device.h kVariantModel


def _hiprtc():
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise OSError("libhiprtc.so not found")


def compile_check(path, variant=False, arch="gfx950"):
    """-> (ok, compiler log, code bytes)"""
    rtc = _hiprtc()
    src = ('#define APEMOST_USER_MODEL 1\n#include "pt_kernels.h"\n#line 1 "%s"\n%s\n' % (path, open(path).read())).encode()
    prog = C.c_void_p()
    if rtc.hiprtcCreateProgram(C.byref(prog), src, b"apemost_user_model.hip", 0, None, None) != 0:
        raise RuntimeError("hiprtcCreateProgram failed")
    km = MODEL_USER + (VARIANT if variant else 0)
    for w in (1, 2, 4, 8):          # the workgroup shapes the engine instantiates (apemost_hip.hip kUserWaves)
        prod = "true" if w >= 4 else "false"
        for name in ("apemost::pt_round_kernel<%d, %d, false, %s>" % (km, w, prod),
                     "apemost::pt_calibrate_kernel<%d, %d, false, %s>" % (km, w, prod),
                     "apemost::pt_calc_model_kernel<%d, %d, false>" % (MODEL_USER, w),
                     "apemost::pt_loglike_kernel<%d, %d, false>" % (MODEL_USER, w)):
            rtc.hiprtcAddNameExpression(prog, name.encode())
    opts = [b"--offload-arch=" + arch.encode(), b"-O3", b"-ffp-contract=off", b"-std=c++17",
            b"-I" + os.path.join(HERE, "csrc").encode(), b"-I" + os.path.join(ROOT, "include").encode(), b"-I/opt/rocm/include"]
    rc = rtc.hiprtcCompileProgram(prog, len(opts), (C.c_char_p * len(opts))(*opts))
    n = C.c_size_t(0)
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    log = C.create_string_buffer(n.value + 1)
    if n.value:
        rtc.hiprtcGetProgramLog(prog, log)
    size = C.c_size_t(0)
    if rc == 0:
        rtc.hiprtcGetCodeSize(prog, C.byref(size))
    rtc.hiprtcDestroyProgram(C.byref(prog))
    return rc == 0, log.value.decode("utf-8", "replace"), size.value


if __name__ == "__main__":
    ok_all = True
    for p in sys.argv[1:]:
        ok, log, size = compile_check(p)
        print("%s: %s%s" % (p, "compiles (%d bytes of gfx950 code)" % size if ok else "DOES NOT COMPILE", "\n" + log if log.strip() else ""))
        ok_all = ok_all and ok
    sys.exit(0 if ok_all else 1)
