#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json from the CPU oracle (oracle/apemost_oracle.c).

The reference cannot be run here (it needs GSL, absent in the image), so these vectors are
outputs of the restatement, which is itself pinned by tests/test_oracle_pins.py and
tests/test_oracle_workflow.py; they freeze it against regressions and give the GPU tests fixed
expected values.  Inputs are fully specified below (no files are read), so the script can be
re-run anywhere:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from apemost_amd import workloads as wl  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_pair  # noqa: E402


def main():
    out = {}
    # G1: likelihood known answers on the synthetic BASELINE data shapes (small n)
    rs = np.random.RandomState(2024)
    ll = []
    for name, n_data in (("simplesin", 200), ("sine3", 150), ("pulse", 120), ("pulse_vrot", 90)):
        w = wl.by_name(name, n_data=n_data, n_chain=2)
        for _ in range(4):
            p = w.pmin + (w.pmax - w.pmin) * rs.uniform(0.1, 0.9, w.n_par)
            beta = float(rs.uniform(0.05, 1.0))
            prob, prior = orc.loglike(w.model, p, w.data, beta=beta)
            ll.append({"workload": name, "n_data": n_data, "params": [float(v) for v in p], "beta": beta,
                       "prob": prob, "prior": prior})
    out["loglike"] = ll
    # G2: beta ladders (src/parallel_tempering_beta.c:53-90)
    out["ladders"] = [{"kind": k, "n_beta": n, "beta_0": b0,
                       "beta": [orc.get_chain_beta(k, i, n, b0) for i in range(n)]}
                      for k in range(7) for n, b0 in ((2, 0.3), (8, 0.01), (20, 0.001))]
    # G3: swap decisions (src/parallel_tempering_interaction.c:25-42, :92)
    L = orc.lib()
    swaps = []
    import ctypes as C
    for _ in range(40):
        ab, bb = float(rs.uniform(0.05, 1)), float(rs.uniform(0.01, 1))
        ap, bp = float(rs.uniform(-3000, -100)), float(rs.uniform(-3000, -100))
        u, u2 = float(rs.uniform()), float(rs.uniform(1e-6, 1))
        n_beta = int(rs.choice([2, 8, 128, 2048]))
        r = C.c_double(0)
        sw = L.orc_swap_decision(ab, bb, ap, bp, float(np.log(u2)), C.byref(r))
        swaps.append({"a_beta": ab, "b_beta": bb, "a_prob": ap, "b_prob": bp, "u": u, "u2": u2, "n_beta": n_beta,
                      "pair": L.orc_swap_pair_index(u, n_beta), "r": r.value, "swapped": bool(sw)})
    out["swaps"] = swaps
    # G4: accept rule (src/markov_chain.c:282-311): (prob_old, prob_new, ln u) -> accepted, drew
    acc = []
    for po, pn in ((-10.0, -10.0), (-10.0, -9.0), (-10.0, -10.5), (-10.0, -30.0), (-1e10, -5.0), (-5.0, float("nan"))):
        for lu in (-0.1, -0.6, -25.0):
            accepted = (pn == po) or (pn > po) or (lu < pn - po)
            acc.append({"prob_old": po, "prob_new": pn if pn == pn else "nan", "log_u": lu, "accept": bool(accepted),
                        "drew": not (pn == po or pn > po)})
    out["accept"] = acc
    # G5: a short trajectory per model in the engine's tick-addressed stream mode
    traj = []
    for name, n_data in (("simplesin", 96), ("pulse", 80), ("pulse_vrot", 72), ("sine3", 64)):
        w = wl.by_name(name, n_data=n_data, n_chain=6)
        st, lad, rng = make_pair(w, 6, seed=20240)
        samples = orc.run_sampler(lad, rng, 12, 5, record=True)
        traj.append({"workload": name, "n_data": n_data, "n_chain": 6, "seed": 20240, "n_rounds": 12, "n_swap": 5,
                     "accept": [int(v) for v in lad.accept], "swapcount": [int(v) for v in lad.swapcount],
                     "ticks": [int(v) for v in rng.ticks], "params": lad.params.tolist(), "prob": lad.prob.tolist(),
                     "prob_best": lad.prob_best.tolist(), "last_rows": samples[-1].tolist()})
    out["trajectories"] = traj
    # G6: RNG addressing
    out["rng"] = {"philox_seed5_sub3_first8": [int(v) for v in orc.philox_stream(5, 3, 8)],
                  "attempts": [list(orc.gaussian_attempt(11, 7, 2, 1000, q)) for q in range(6)],
                  "accept_log_u": [orc.accept_log_uniform(11, 7, 4, t) for t in (0, 1, 999)]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
