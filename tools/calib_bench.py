import sys, time
sys.path.insert(0, '.')
import numpy as np
from apemost_amd import capi, workloads as wl
from apemost_amd.sampler import HipSampler
from apemost_amd.state import LadderState
for name, n_chain, n_data in (("simplesin", 128, 1024), ("pulse_vrot", 512, 4096)):
    w = wl.by_name(name, n_data=n_data, n_chain=n_chain)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=3)
    st = LadderState.from_params(n_chain, w.start, w.pmin, w.pmax, w.step)
    s.set_state(st)
    cfg = capi.calib_defaults(burn_in_iterations=2000)
    t = time.time(); rc = s.calibrate_first(cfg); t1 = time.time() - t
    st = s.get_state()
    st.params_best[0] = st.params[0]
    t = time.time(); status, beta_0, fac = s.calibrate_rest(cfg); t2 = time.time() - t
    st = s.get_state()
    print(name, n_chain, "chains: calibrate_first %.2fs rc=%d, calibrate_rest %.2fs status=%d beta_0=%.4g" % (t1, rc, t2, status, beta_0 or -1))
    print("   ticks(min/max)", st.ticks[1:].min(), st.ticks[1:].max(), "steps chain0", st.step[0], "hot", st.step[-1])
    s.close()
