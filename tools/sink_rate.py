#!/usr/bin/env python3
"""End-to-end rate of the C host's run phase per sample sink (BASELINE config 2: simplesin, 128
chains x 1024 points): Metropolis steps/s from process start to exit, against the kernel-only rate
of bench.py.  Writes into a scratch directory (default /dev/shm, a tmpfs).

    python tools/sink_rate.py [iterations] [scratch dir]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apemost_amd import build, workloads as wl  # noqa: E402
from tests import hostlib  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    scratch = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
    n_beta = 128
    build.build_hip()
    top = tempfile.mkdtemp(dir=scratch, prefix="apemost_sink_")
    bins = tempfile.mkdtemp(prefix="apemost_sink_bin_")   # executables: /dev/shm is usually mounted noexec
    try:
        w = wl.simplesin(n_data=1024, n_chain=n_beta)
        exe = hostlib.make(os.path.join(bins, "sine.exe"), strict="-std=c99 -O2",
                           ccflags="-DN_BETA=%d -DBURN_IN_ITERATIONS=2000 -DMAX_ITERATIONS=%d" % (n_beta, iters))
        base = os.path.join(top, "calib")
        os.mkdir(base)
        open(os.path.join(base, "params"), "w").write(w.params_file_text())
        open(os.path.join(base, "data"), "w").write(w.data_file_text())
        env = dict(os.environ, APEMOST_SEED="1")
        for phase in ("calibrate_first", "calibrate_rest"):
            subprocess.check_call([exe, phase], cwd=base, env=env, stdout=subprocess.DEVNULL)
        for mode, n in (("binary", iters), ("binary:all", iters), ("binary,thin:10", iters), ("text,thin:100", iters), ("text", iters // 30)):
            work = os.path.join(top, mode.replace(",", "_").replace(":", ""))
            shutil.copytree(base, work)
            exe_n = exe
            if n != iters:
                exe_n = hostlib.make(os.path.join(bins, "sine_short.exe"), strict="-std=c99 -O2",
                                     ccflags="-DN_BETA=%d -DMAX_ITERATIONS=%d" % (n_beta, n))
            t0 = time.time()
            subprocess.check_call([exe_n, "run"], cwd=work, env=dict(env, APEMOST_DUMP=mode), stdout=subprocess.DEVNULL)
            dt = time.time() - t0
            size = sum(os.path.getsize(os.path.join(work, f)) for f in os.listdir(work))
            print("APEMOST_DUMP=%-16s %9d iterations x %d chains in %6.2f s = %.3g steps/s end to end, %.2f GB written"
                  % (mode, n, n_beta, dt, n * n_beta / dt, size / 1e9), flush=True)
            shutil.rmtree(work)
    finally:
        shutil.rmtree(top, ignore_errors=True)
        shutil.rmtree(bins, ignore_errors=True)


if __name__ == "__main__":
    main()
