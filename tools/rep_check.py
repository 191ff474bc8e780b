#!/usr/bin/env python3
"""Diagnostic: run-to-run bit identity of a BASELINE config 3 run (four repetitions), with the
fields that differ if any do."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from apemost_amd import workloads as wl
from apemost_amd.sampler import HipSampler
from helpers import make_pair
w = wl.by_name("sine3", n_data=8192, n_chain=1024)
st, _, _ = make_pair(w, 1024, seed=13)
outs = []
for rep in range(4):
    s = HipSampler(w.model, w.n_par, 1024, w.data, seed=13)
    s.set_state(st)
    d = torch.zeros((6, 1024, w.n_par + 2), dtype=torch.float64, device="cuda")
    s.run_sampler(6, 1, d.data_ptr()); s.synchronize()
    outs.append((s.get_state(), d.cpu().numpy())); print("waves", s.waves_per_chain if hasattr(s, "waves_per_chain") else "?")
    s.close()
a, sa = outs[0]
for i, (b, sb) in enumerate(outs[1:]):
    dif = np.argwhere(sa != sb)
    print("rep", i + 1, "sample diffs", len(dif), dif[:5].tolist(), "params", int((a.params != b.params).sum()), "prob", int((a.prob != b.prob).sum()),
          "accept", int((a.accept != b.accept).sum()), "swapcount", int((a.swapcount != b.swapcount).sum()))
    if len(dif):
        k = tuple(dif[0]); print("  e.g.", sa[k], sb[k])
