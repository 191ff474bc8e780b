#!/usr/bin/env python3
"""Diagnostic: run-phase rate of a pulse ladder (any number of modes) with and without the one-barrier kernel's
helper wavefront (APEMOST_OB_HELPER=0/1), at the geometry the engine chooses.  One GPU.
    python tools/helper_rate.py [n_data] [rounds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from apemost_amd import workloads as wl  # noqa: E402
from apemost_amd.sampler import HipSampler, get_chain_beta  # noqa: E402
from apemost_amd.state import LadderState  # noqa: E402


def pulse_modes(n_modes, n_data, n_chain, seed=7):
    rs = np.random.RandomState(seed)
    nu = np.linspace(10, 12, n_data)
    modes = [(10.2 + 1.6 * k / max(1, n_modes - 1), 4.0 - 0.4 * k) for k in range(n_modes)]
    y = wl._lorentz(nu, 5.0, modes, 0.05)
    d = y * rs.exponential(1.0, n_data)
    params = [(5.0, 0.1, 50, "lifetime", -1), (0.05, 0, 1, "p1", -1)]
    for k, (f, h) in enumerate(modes):
        params += [(f, 10, 12, "freq%d" % k, -1), (h, 0, 20, "height%d" % k, -1)]
    return wl.Workload("pulse", wl.MODEL_PULSE, params, np.stack([nu, d], 1), n_chain, 1)


def rate(w, n_chain, rounds, helper):
    os.environ["APEMOST_OB_HELPER"] = "1" if helper else "0"
    st = LadderState.from_params(n_chain, w.start, w.pmin, w.pmax, w.step * 0.3)
    for i in range(n_chain):
        st.beta[i] = get_chain_beta(0, i, n_chain, 0.02)
        st.step[i] = np.minimum(st.step[i] * st.beta[i] ** -0.5, w.pmax - w.pmin)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=3)
    s.set_state(st)
    s.run_sampler(rounds, 1)
    s.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        s.run_sampler(rounds * 4, 1)
        s.synchronize()
        best = max(best, rounds * 4 * n_chain / (time.perf_counter() - t0))
    geo = (s.geometry[0], s.ob_helper)
    s.close()
    return best, geo


def main():
    n_data = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    print("pulse, %d points, one swap attempt per step: steps/s without / with the helper wavefront (likelihood waves)" % n_data)
    for n_modes in (1, 2, 3, 4, 6):
        for n_chain in (64, 128, 256):
            w = pulse_modes(n_modes, n_data, n_chain)
            r0, g0 = rate(w, n_chain, rounds, False)
            r1, g1 = rate(w, n_chain, rounds, True)
            assert g1[1] and not g0[1]
            print("  %d modes (n_par %2d), %3d chains, %d waves:  %.3e  %.3e  (%+.1f %%)" % (n_modes, w.n_par, n_chain, g1[0], r0, r1, 100 * (r1 / r0 - 1)))


if __name__ == "__main__":
    main()
