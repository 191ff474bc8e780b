#!/usr/bin/env python3
"""Diagnostic: where does one Metropolis step of the round kernel spend its cycles?
Uses the -DAPEMOST_STAMPS twin of the library (never the product build); prints the share of
each step segment for workgroup 0.  Shares, not lengths, are meaningful (the stamps cost cycles)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMP_WAVE = int(os.environ.get("APEMOST_STAMP_WAVE", "0"))  # build_stamps(wave=...) made the twin
os.environ["APEMOST_HIP_LIB"] = os.environ.get("APEMOST_STAMP_LIB") or os.path.join(
    ROOT, "apemost_amd", "libapemost_hip_stamps%s.so" % ("" if STAMP_WAVE == 0 else "_tl" if STAMP_WAVE < 0 else "_w%d" % STAMP_WAVE))

import numpy as np  # noqa: E402
from apemost_amd import capi, workloads as wl  # noqa: E402
from apemost_amd.sampler import HipSampler, get_chain_beta  # noqa: E402
from apemost_amd.state import LadderState  # noqa: E402

SEG = ["bookkeeping", "refill", "propose", "barrier A", "likelihood terms", "partials+finish", "wave reduce",
       "barrier B", "accept", "-", "-", "duty window"]
# stamps in the order a step passes them
ORDER = [0, 1, 2, 3, 4, 6, 7, 5, 8, 11]
NAME = {0: "start", 1: "refill", 2: "proposed", 3: "A out", 4: "terms", 6: "reduced", 7: "B out",
        5: "finish", 8: "accept", 11: "end"}


def timeline(L, waves):
    """APEMOST_STAMP_WAVE=-1 twin: cycles since the step's first stamp, per wave"""
    out = (C.c_uint64 * (4 * 16 * 12))()
    steps, points = C.c_int(), C.c_int()
    L.apemost_hip_debug_timeline(out, C.byref(steps), C.byref(points))
    for st in range(steps.value):
        rows = [[out[(st * 16 + w) * points.value + i] for i in range(points.value)] for w in range(waves)]
        t0 = min(v for r in rows for v in r if v)
        print(" step %d   %s" % (st, " ".join("%8s" % NAME[i] for i in ORDER)))
        for w in range(waves):
            print("   wave %d %s" % (w, " ".join("%8d" % (rows[w][i] - t0) if rows[w][i] else "       -" for i in ORDER)))



def main():
    waves_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 8]
    name = sys.argv[2] if len(sys.argv) > 2 else "simplesin"
    n_chain = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    n_data = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    w = wl.by_name(name, n_data=n_data, n_chain=n_chain)
    st = LadderState.from_params(n_chain, w.start, w.pmin, w.pmax, w.step * 0.3)
    for i in range(n_chain):
        st.beta[i] = get_chain_beta(0, i, n_chain, 0.02)
        st.step[i] = np.minimum(st.step[i] * st.beta[i] ** -0.5, w.pmax - w.pmin)
    L = capi.lib()
    L.apemost_hip_debug_stamps.argtypes = [C.POINTER(C.c_uint64)]
    for waves in waves_list:
        s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=1, waves_per_chain=waves)
        s.set_state(st)
        s.run_sampler(20, 15)
        out = (C.c_uint64 * 16)()
        L.apemost_hip_debug_stamps(out)
        n_steps = 200 * 15
        s.run_sampler(200, 15)
        L.apemost_hip_debug_stamps(out)
        if STAMP_WAVE < 0:
            timeline(L, waves)
            s.close()
            continue
        tot = sum(out[:10]) + out[11]
        print("waves=%d  %s  cycles/step=%.0f" % (waves, name, tot / n_steps))
        for i, nm in enumerate(SEG):
            if nm == "-":
                continue
            print("   %-20s %8.0f cyc/step  %5.1f %%" % (nm, out[i] / n_steps, 100.0 * out[i] / tot))
        s.close()


if __name__ == "__main__":
    main()
