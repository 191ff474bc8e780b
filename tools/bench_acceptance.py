#!/usr/bin/env python3
"""Diagnostic: acceptance rate per chain of the bench workload (config 2 start state)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from apemost_amd import workloads as wl
from apemost_amd.sampler import HipSampler, get_chain_beta
from apemost_amd.state import LadderState
n = 128
w = wl.by_name("simplesin", n_data=1024, n_chain=n)
st = LadderState.from_params(n, w.start, w.pmin, w.pmax, w.step * 0.3)
for i in range(n):
    st.beta[i] = get_chain_beta(0, i, n, 0.02)
    st.step[i] = np.minimum(st.step[i] * st.beta[i] ** -0.5, w.pmax - w.pmin)
s = HipSampler(w.model, w.n_par, n, w.data, seed=1)
s.set_state(st)
s.run_sampler(2000, 15)
a = s.get_state()
r = a.accept / (a.accept + a.reject)
print("acceptance: mean %.3f  min %.3f  max %.3f; cold chain %.3f hot chain %.3f" % (r.mean(), r.min(), r.max(), r[0], r[-1]))
