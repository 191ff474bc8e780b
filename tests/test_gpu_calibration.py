"""GPU parity of the calibration phases against the CPU oracle (STREAMS mode = same tick-addressed
draws): burn_in (src/markov_chain.c:34-79), markov_chain_calibrate_orig
(src/markov_chain_calibrate.c:1039-1180), markov_chain_step_for (src/markov_chain.c:317-333),
calibrate_first / calibrate_rest (src/parallel_tempering.c:78-207) and the beta ladder
(src/parallel_tempering_beta.c:53-102).  Integer results (status, sweep counts, counters, ticks)
bit-exact; step widths, betas, factors and positions to rel 1e-9."""
import numpy as np
import pytest

from apemost_amd import capi, workloads as wl
from apemost_amd.sampler import HipSampler
from apemost_amd.state import LadderState
from oracle import oracle as orc
from tests.helpers import assert_match, make_pair, small_workloads, to_oracle

pytestmark = pytest.mark.gpu

BURN, LIMIT = 600, 20000


def _cfgs(burn=BURN, limit=LIMIT):
    return (capi.calib_defaults(burn_in_iterations=burn, iter_limit=limit),
            orc.calib_defaults(burn_in_iterations=burn, iter_limit=limit))


@pytest.mark.parametrize("name", ["simplesin", "sine3", "pulse", "pulse_vrot"])
@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_calibration_matches_oracle(name, waves):
    """pt_calibrate_kernel for every model and every workgroup shape: waves 4 and 8 are the
    PRODUCER variants (candidate sets prepared ahead by waves 1-3 while step widths and the
    current point change under them), which is what bench.py launches before its timed region"""
    w = small_workloads()[name]
    n_chain = 4
    st, lad, rng = make_pair(w, n_chain, seed=5, init_prob=True)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=5, waves_per_chain=waves)
    assert s.geometry[0] == waves
    s.set_state(st)
    dcfg, ocfg = _cfgs()
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    dev = s.get_state()
    for c in range(n_chain):
        st_o, it_o = orc.markov_chain_calibrate(lad, rng, c, ocfg)
        assert status[c] == st_o and iters[c] == it_o, (c, status[c], st_o, iters[c], it_o)
    assert iters.min() >= 200
    assert_match(dev, lad, rng, what="calibrate %s waves=%d" % (name, waves))
    s.close()


@pytest.mark.parametrize("waves", [4, 8])
def test_calibration_with_rows_in_registers_matches_oracle(waves):
    """BASELINE config 2's shape: 1024 points = one pass of the interleaved loop, so every lane keeps
    its data rows in registers (Engine::cache_rows) -- exactly <SIMPLESIN, 8, LDS, PROD>, the kernel
    the bench calibrates its ladder with"""
    w = wl.simplesin(n_data=1024, n_chain=128)
    n_chain = 3
    st, lad, rng = make_pair(w, n_chain, seed=12, init_prob=True)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=12, waves_per_chain=waves, lds_policy=1)
    assert s.geometry == (waves, True)
    s.set_state(st)
    dcfg, ocfg = _cfgs(burn=400)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    dev = s.get_state()
    for c in range(n_chain):
        st_o, it_o = orc.markov_chain_calibrate(lad, rng, c, ocfg)
        assert status[c] == st_o and iters[c] == it_o, (c, status[c], st_o, iters[c], it_o)
    assert_match(dev, lad, rng, what="calibrate config-2 shape waves=%d" % waves)
    s.close()


@pytest.mark.parametrize("name,waves", [("simplesin", 1), ("simplesin", 8), ("pulse_vrot", 4), ("sine3", 2)])
def test_burn_in_only_matches_oracle(name, waves):
    """burn_in_only = 1 is the -DSKIP_CALIBRATE_ALLCHAINS path of calibrate_rest
    (src/parallel_tempering.c:190-196): burn_in() alone, step widths restored afterwards"""
    w = small_workloads()[name]
    n_chain = 5
    st, lad, rng = make_pair(w, n_chain, seed=21, init_prob=True)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=21, waves_per_chain=waves)
    s.set_state(st)
    dcfg, _ = _cfgs(burn=1000)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg, burn_in_only=True)
    dev = s.get_state()
    assert not status.any() and not iters.any()
    for c in range(n_chain):
        orc.burn_in(lad, rng, c, 1000)
    assert np.array_equal(dev.step, st.step)                     # restored bit for bit
    assert np.all(dev.ticks == 1000) and np.all(dev.accept + dev.reject == 1000)
    assert_match(dev, lad, rng, what="burn_in %s" % name)
    s.close()


@pytest.mark.parametrize("name,waves", [("simplesin", 1), ("simplesin", 4), ("pulse", 2), ("sine3", 8)])
def test_launch_round_for_matches_oracle_step_for(name, waves):
    """apemost_hip_launch_round_for = n x markov_chain_step_for(m, p): only parameter p moves, only
    its counters count, m->accept / m->reject stay (quirk Q5); what the host layer's
    markov_chain_step_for() calls"""
    import torch
    w = small_workloads()[name]
    n_chain = 6
    st, lad, rng = make_pair(w, n_chain, seed=33, init_prob=True)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=33, waves_per_chain=waves)
    s.set_state(st)
    plan = [(p % w.n_par, 3 + p) for p in range(w.n_par + 2)] + [(w.n_par - 1, 40), (0, 40)]
    for p, n in plan:
        d = torch.zeros((n, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
        s.markov_chain_step_for(p, n, d.data_ptr())
        s.synchronize()
        ref = np.zeros((n, n_chain, w.n_par + 2))
        for k in range(n):
            for c in range(n_chain):
                orc.step_for(lad, rng, c, p)
                orc.check_best(lad, c)
                lad.n_iter[c] += 1
                ref[k, c, :w.n_par] = lad.params[c]
                ref[k, c, w.n_par] = lad.prob[c]
                ref[k, c, w.n_par + 1] = lad.prob[c] - lad.prior[c]
        np.testing.assert_allclose(d.cpu().numpy(), ref, rtol=1e-9, atol=1e-300)
    dev = s.get_state()
    assert not dev.accept.any() and not dev.reject.any()
    assert (dev.params_accepts + dev.params_rejects).sum() == n_chain * sum(n for _, n in plan)
    assert s.round == (0, False)                                 # single-parameter rounds never owe a swap
    assert_match(dev, lad, rng, what="step_for %s" % name)
    with pytest.raises(capi.ApemostHipError):
        s.markov_chain_step_for(w.n_par, 1)
    s.close()


@pytest.mark.parametrize("name", ["simplesin", "pulse_vrot"])
@pytest.mark.parametrize("skip", [False, True])
def test_calibrate_first_and_rest_match_oracle(name, skip):
    """HipSampler.calibrate_first / calibrate_rest (device kernels + the host's ladder arithmetic)
    against orc_calibrate_first / orc_calibrate_rest: beta_0, every beta, the stepwidth factors,
    the seeded start points and the calibrated widths of the whole ladder"""
    w = small_workloads()[name]
    n_beta = 6
    st = LadderState.from_params(n_beta, w.start, w.pmin, w.pmax, w.step)
    lad = orc.Ladder(w.model, n_beta, w.n_par, w.data)
    to_oracle(st, lad)
    rng = orc.Rng(orc.RNG_STREAMS, 9, lad)
    s = HipSampler(w.model, w.n_par, n_beta, w.data, seed=9)
    s.set_state(st)
    dcfg, ocfg = _cfgs(limit=100000)       # the cold chain needs ~22 000 sweeps from the params-file widths
    assert s.calibrate_first(dcfg) == orc.calibrate_first(lad, rng, ocfg) == 0
    assert_match(s.get_state(), lad, rng, what="calibrate_first " + name)
    status, beta_0, factors = s.calibrate_rest(dcfg, skip_calibrate_allchains=skip)
    o_status, o_beta_0, o_factors = orc.calibrate_rest(lad, rng, ocfg, skip_calibrate_allchains=skip)
    assert status == o_status == 0
    assert abs(beta_0 - o_beta_0) <= 1e-9 * o_beta_0 and 0 < beta_0 < 1
    np.testing.assert_allclose(factors, o_factors, rtol=1e-9)
    dev = s.get_state()
    assert dev.beta[0] == 1.0 and abs(dev.beta[-1] - beta_0) < 1e-15 and np.all(np.diff(dev.beta) < 0)
    assert_match(dev, lad, rng, what="calibrate_rest " + name)
    s.close()


def test_config5_shard_size_with_device_calibration():
    """BASELINE config 5 at one GPU's share: pulse_vrot, 2048 of 16384 beta-chains x 65536 data
    points, step-size calibration on the device, then sampling.  Checked against the oracle on a
    sub-ladder (first, middle, last chain of the shard): calibration status, sweep counts, final
    step widths; the run must be bit-reproducible."""
    import torch
    n_local, n_global, lo = 2048, 16384, 4096
    w = wl.pulse_vrot(n_data=65536, n_chain=n_global)
    st, _, _ = make_pair(w, n_local, seed=31, chain_offset=lo, n_global=n_global)
    dcfg = capi.calib_defaults(burn_in_iterations=200, iter_limit=600, no_rescaling_limit=1, max_ar_deviation=0.2)
    ocfg = orc.calib_defaults(burn_in_iterations=200, iter_limit=600, no_rescaling_limit=1, max_ar_deviation=0.2)
    outs = []
    for rep in range(2):
        s = HipSampler(w.model, 7, n_local, w.data, seed=31, chain_offset=lo, n_chains_global=n_global)
        s.set_state(st)
        s.calc_model(0, n_local)
        status, iters = s.markov_chain_calibrate(0, n_local, dcfg)
        cal = s.get_state()
        d = torch.zeros((4, n_local, 9), dtype=torch.float64, device="cuda")
        s.launch_round(4, False, d.data_ptr())      # a shard: four steps, no swap attempt in between
        s.synchronize()
        outs.append((status, iters, cal, s.get_state(), d.cpu().numpy()))
        s.close()
    (sa, ia, ca, ra, da), (sb, ib, cb, rb, db) = outs
    assert np.array_equal(sa, sb) and np.array_equal(ia, ib) and np.array_equal(da, db)
    for f in ("step", "params", "prob", "params_best", "ticks"):
        assert np.array_equal(getattr(ca, f), getattr(cb, f)) and np.array_equal(getattr(ra, f), getattr(rb, f)), f
    assert np.all(ia >= 200) and set(np.unique(sa)) <= {0, 2}
    # (a chain that ran into the iteration limit keeps the counters of its last 200-step block)
    assert np.all((ra.accept + ra.reject) - (ca.accept + ca.reject) == 4) and np.all(ra.n_iter - ca.n_iter == 4)
    assert np.all(da[..., :7] >= w.pmin) and np.all(da[..., :7] <= w.pmax) and np.all(np.isfinite(da))
    # sub-ladder against the oracle: each chain is independent during calibration
    for c in (0, n_local // 2, n_local - 1):
        one = st.slice(c, c + 1)
        lad = orc.Ladder(w.model, 1, 7, w.data, chain_offset=lo + c)
        to_oracle(one, lad)
        rng = orc.Rng(orc.RNG_STREAMS, 31, lad)
        orc.calc_model(lad, 0)
        st_o, it_o = orc.markov_chain_calibrate(lad, rng, 0, ocfg)
        assert (sa[c], ia[c]) == (st_o, it_o), (c, sa[c], st_o, ia[c], it_o)
        np.testing.assert_allclose(ca.step[c], lad.step[0], rtol=1e-9)
        np.testing.assert_allclose(ca.params[c], lad.params[0], rtol=1e-9)
        np.testing.assert_allclose(ca.prob_best[c], lad.prob_best[0], rtol=1e-9)
        assert ca.ticks[c] == rng.ticks[0]
