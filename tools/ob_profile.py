#!/usr/bin/env python3
"""Diagnostic for the one-barrier round kernel: per wave of workgroup 0, the cycles per step between
leaving a step's barrier and arriving at the next (the wave's own work); the wave with the largest
figure is the one the others wait for.  Uses the -DAPEMOST_STAMPS twin of the library.
    python tools/ob_profile.py [model] [n_chain] [n_data] [waves]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["APEMOST_HIP_LIB"] = os.environ.get("APEMOST_STAMP_LIB") or os.path.join(ROOT, "apemost_amd", "libapemost_hip_stamps.so")

import numpy as np  # noqa: E402
from apemost_amd import capi, workloads as wl  # noqa: E402
from apemost_amd.sampler import HipSampler, get_chain_beta  # noqa: E402
from apemost_amd.state import LadderState  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "simplesin"
    n_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    n_data = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    waves = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    w = wl.by_name(name, n_data=n_data, n_chain=n_chain)
    st = LadderState.from_params(n_chain, w.start, w.pmin, w.pmax, w.step * 0.3)
    for i in range(n_chain):
        st.beta[i] = get_chain_beta(0, i, n_chain, 0.02)
        st.step[i] = np.minimum(st.step[i] * st.beta[i] ** -0.5, w.pmax - w.pmin)
    L = capi.lib()
    L.apemost_hip_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=1, waves_per_chain=waves)
    s.set_state(st)
    out = (C.c_uint64 * 16)()
    s.run_sampler(20, 15)
    L.apemost_hip_debug_stamps(out)
    n_steps = 200 * 15
    s.run_sampler(200, 15)
    L.apemost_hip_debug_stamps(out)
    print("one-barrier kernel, %s, %d chains x %d points, %d likelihood waves: cycles per step (workgroup 0)" % (name, n_chain, n_data, waves))
    print("   whole step (owner, barrier to barrier)  %7.0f" % (out[15] / n_steps))
    for hw in range(waves + 4):
        role = "likelihood" if hw < waves else "owner" if hw == waves else "producer"
        print("   wave %2d %-10s busy %7.0f" % (hw, role, out[hw] / n_steps))
    if s.ob_helper:
        print("   wave %2d %-10s busy %7.0f" % (waves + 4, "helper", out[14] / n_steps))
    if os.environ.get("APEMOST_STAMP_PHASES"):
        # a twin built with -DAPEMOST_STAMP_PHASES: the first producer's step by phase (each phase every third step)
        for i, nm in enumerate(["Philox blocks, polar test", "logarithm", "division, square root, store"]):
            print("   producer phase %d (%s): %7.0f ticks" % (i, nm, out[8 + i] / (n_steps / 3.0)))
    elif waves == 4:
        names = ["LDS batch, partial sums, decision", "finish, counters, best point, sample row", "the proposal in flight (choose)",
                 "next candidates, both prepared proposals", "the prior of the proposal in flight", "threshold, flags"]
        print("   the owner's step in segments (ticks between the points it reaches):")
        for i, nm in enumerate(names):
            print("      %-45s %7.0f" % (nm, out[8 + i] / n_steps))
    s.close()


if __name__ == "__main__":
    main()
