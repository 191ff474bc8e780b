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


@pytest.mark.parametrize("name", ["pulse", "pulse_vrot"])
@pytest.mark.parametrize("waves", [4, 8])
def test_one_barrier_calibration_without_the_helper_wavefront_matches_oracle(name, waves, monkeypatch):
    """pt_calibrate_ob_kernel<..., HELPER = false> for the models with a prior: what ladders of more than one chain per CU
    calibrate in (257-512 chains), where the owner computes the proposal's prior itself.  Small ladders get the helper
    by default (the test above); APEMOST_OB_HELPER=0 selects the other kernel, which must give the same chains."""
    monkeypatch.setenv("APEMOST_OB_HELPER", "0")
    w = small_workloads()[name]
    n_chain = 4
    st, lad, rng = make_pair(w, n_chain, seed=5, init_prob=True)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=5, waves_per_chain=waves)
    assert s.geometry[0] == waves and s.launch_policy[0] and not s.ob_helper
    s.set_state(st)
    dcfg, ocfg = _cfgs()
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    dev = s.get_state()
    for c in range(n_chain):
        st_o, it_o = orc.markov_chain_calibrate(lad, rng, c, ocfg)
        assert status[c] == st_o and iters[c] == it_o, (c, status[c], st_o, iters[c], it_o)
    assert_match(dev, lad, rng, what="calibrate %s waves=%d without helper" % (name, waves))
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


@pytest.mark.parametrize("name,waves", [("simplesin", 1), ("simplesin", 4), ("pulse", 2), ("sine3", 8), ("pulse", 1), ("pulse_vrot", 1)])
def test_launch_round_for_matches_oracle_step_for(name, waves):
    """apemost_hip_launch_round_for = n x markov_chain_step_for(m, p): only parameter p moves, only
    its counters count, m->accept / m->reject stay (quirk Q5); what the host layer's
    markov_chain_step_for() calls.  The one-wave kernels of the pulse models take single-parameter
    updates of the additive parameter (p = 1) without walking the data vector (Engine::kOffsetShortcut):
    the rows must be the oracle's all the same."""
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


# ---- round 3: the calibration as a sequence of launches -----------------------------------------
def _oracle_calibration(w, n_chain, seed, ocfg, progress_path=None):
    st, lad, rng = make_pair(w, n_chain, seed=seed, init_prob=True)
    orc.set_progress_path(progress_path)
    try:
        res = [orc.markov_chain_calibrate(lad, rng, c, ocfg) for c in range(n_chain)]
    finally:
        orc.set_progress_path(None)
    return st, lad, rng, res


@pytest.mark.parametrize("name,waves", [("simplesin", 1), ("simplesin", 4), ("pulse", 8), ("pulse_vrot", 2), ("sine3", 4)])
def test_calibration_cut_at_every_block_matches_oracle(name, waves, monkeypatch):
    """APEMOST_CALIB_SEGMENT_EVALS=1 ends every launch after ONE block (200 burn-in steps, 200 sweeps, or
    the 200 steps behind them): the whole state of a calibration -- stage, burn-in position, sweeps,
    rat_limit, the no-rescaling count, the saved step widths, and the chain -- goes through HBM between
    any two blocks.  Same oracle run as one launch would have to match; waves 4 and 8 are the
    one-barrier kernel (single-parameter sweeps with both next proposals prepared ahead)."""
    monkeypatch.setenv("APEMOST_CALIB_SEGMENT_EVALS", "1")
    w = small_workloads()[name]
    n_chain = 3
    dcfg, ocfg = _cfgs()
    st, lad, rng, res = _oracle_calibration(w, n_chain, 17, ocfg)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=17, waves_per_chain=waves)
    s.set_state(st)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    segments, evals, by_waves = s.calibrate_stats()
    assert [(int(a), int(b)) for a, b in zip(status, iters)] == res
    assert_match(s.get_state(), lad, rng, what="segmented calibrate %s waves=%d" % (name, waves))
    blocks = max(3 + 2 * (it // 200) for _, it in res)       # burn-in 600 = 3 blocks, then two per readjustment
    assert segments == blocks and list(by_waves) == [waves]
    assert evals == sum(600 + it * w.n_par + (it // 200) * 200 for _, it in res)
    s.close()


@pytest.mark.parametrize("name,waves", [("simplesin", 2), ("simplesin", 4), ("pulse", 4), ("sine3", 8)])
def test_two_phase_calibration_of_several_waves_cut_at_every_block_matches_oracle(name, waves, monkeypatch):
    """pt_calibrate_kernel (the two-phase step) with more than one wave per chain, a launch per block:
    every launch goes through the transitions that take no step (INIT -> BURN1, BURN1 -> BURN2, BURN2 ->
    SWEEP), where all waves read the LDS record that thread 0 rewrites; they read it behind a barrier
    of its own since round 4.  Waves 4 and 8 reach this kernel through
    APEMOST_HIP_FLAG_TWO_BARRIER_STEP (PRODUCER variants)."""
    monkeypatch.setenv("APEMOST_CALIB_SEGMENT_EVALS", "1")
    w = small_workloads()[name]
    n_chain = 3
    dcfg, ocfg = _cfgs()
    st, lad, rng, res = _oracle_calibration(w, n_chain, 19, ocfg)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=19, waves_per_chain=waves,
                   flags=capi.FLAG_TWO_BARRIER_STEP)
    s.set_state(st)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    assert [(int(a), int(b)) for a, b in zip(status, iters)] == res
    assert_match(s.get_state(), lad, rng, what="segmented two-phase calibrate %s waves=%d" % (name, waves))
    s.close()


def test_tail_of_a_calibration_gets_more_waves_per_chain_and_still_matches_oracle(monkeypatch):
    """The survivors of a calibration are launched again with the workgroup shape the engine would
    choose for a ladder of that many chains.  800 chains start with one wave each; the chains that
    need the most sweeps end the calibration with more (two from 768 chains down, the four-wave
    one-barrier kernel from 512 down).  Whatever the cut,
    status and sweep counts equal the oracle's for every chain checked (first to finish, median,
    last to finish, and a sample in between)."""
    monkeypatch.setenv("APEMOST_CALIB_SEGMENT_EVALS", "1000")     # a readjustment per launch
    w = wl.simplesin(n_data=512, n_chain=800)
    n_chain = 800
    dcfg, ocfg = _cfgs(burn=400)
    st, _, _ = make_pair(w, n_chain, seed=23, init_prob=False)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=23)
    assert s.geometry[0] == 1
    s.set_state(st)
    s.calc_model(0, n_chain)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    dev = s.get_state()
    segments, _, by_waves = s.calibrate_stats()
    assert by_waves.get(1, 0) > 0 and len(by_waves) >= 2 and segments == sum(by_waves.values()), by_waves
    order = np.argsort(iters, kind="stable")
    picks = sorted(set([int(order[0]), int(order[n_chain // 2]), int(order[-1]), int(order[-2])] + list(range(0, n_chain, 97))))
    for c in picks:
        one = st.slice(c, c + 1)
        lad = orc.Ladder(w.model, 1, 4, w.data, chain_offset=c)
        to_oracle(one, lad)
        rng = orc.Rng(orc.RNG_STREAMS, 23, lad)
        orc.calc_model(lad, 0)
        assert (int(status[c]), int(iters[c])) == orc.markov_chain_calibrate(lad, rng, 0, ocfg), c
        np.testing.assert_allclose(dev.step[c], lad.step[0], rtol=1e-9)
        np.testing.assert_allclose(dev.params[c], lad.params[0], rtol=1e-9)
        np.testing.assert_allclose(dev.prob_best[c], lad.prob_best[0], rtol=1e-9)
        assert dev.ticks[c] == rng.ticks[0]
    s.close()


@pytest.mark.parametrize("waves", [1, 4])
def test_calibration_progress_log_equals_the_oracles_file(waves, tmp_path):
    """calibration_progress.data (src/markov_chain_calibrate.c:1052, 1141-1146): the reference reopens
    the file "w" for every chain, so after a single-threaded pass over the chains it holds the lines
    of the last one.  The device logs that chain's readjustments; the text must equal the file the
    oracle writes, byte for byte (step widths move by exact factors, accept rates are k/200)."""
    w = small_workloads()["simplesin"]
    n_chain = 3
    dcfg, ocfg = _cfgs()
    path = tmp_path / "calibration_progress.data"
    st, lad, rng, res = _oracle_calibration(w, n_chain, 29, ocfg, progress_path=path)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=29, waves_per_chain=waves)
    s.set_state(st)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    assert [(int(a), int(b)) for a, b in zip(status, iters)] == res
    text = s.calibration_progress_text()
    assert text == path.read_text()
    assert len(text.splitlines()) == 4 * (res[-1][1] // 200) and text.splitlines()[0].startswith("0\t200\t")
    # any chain of the range can be the logged one
    mid = capi.calib_defaults(burn_in_iterations=BURN, iter_limit=LIMIT, progress_chain=1)
    s.set_state(st)
    s.set_state(LadderState(n_chain, 4), ("ticks",))       # rewind the RNG addresses
    s.markov_chain_calibrate(0, n_chain, mid)
    rows = s.calibrate_progress()
    assert rows.shape == (res[1][1] // 200, 9) and rows[-1, 0] == res[1][1]
    with pytest.raises(capi.ApemostHipError):
        s.markov_chain_calibrate(0, 2, capi.calib_defaults(progress_chain=2))   # outside the range
    s.close()


def test_calibration_can_be_polled_and_cancelled(monkeypatch):
    """begin / poll / cancel / end: poll never blocks and reports the chains still calibrating; after
    cancel the chains that were not done come back with status -1 in a consistent state (whole
    blocks), and the sampler calibrates again afterwards"""
    import ctypes as C
    import time
    monkeypatch.setenv("APEMOST_CALIB_SEGMENT_EVALS", "1")
    w = small_workloads()["simplesin"]
    n_chain = 4
    st, lad, rng = make_pair(w, n_chain, seed=41, init_prob=True)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=41)
    s.set_state(st)
    dcfg, ocfg = _cfgs()
    L = s.L
    capi.check(L.apemost_hip_calibrate_begin(s._h, 0, n_chain, C.byref(dcfg), 0))
    assert L.apemost_hip_calibrate_begin(s._h, 0, n_chain, C.byref(dcfg), 0) == capi.ERR_INVALID
    active = C.c_int32(-1)
    seen = 0
    while s.calibrate_stats()[0] < 6:                  # a few segments on
        capi.check(L.apemost_hip_calibrate_poll(s._h, C.byref(active)))
        assert 0 < active.value <= n_chain
        seen += 1
        time.sleep(0.001)
    capi.check(L.apemost_hip_calibrate_cancel(s._h))
    status = np.zeros(n_chain, dtype=np.int32)
    iters = np.zeros(n_chain, dtype=np.uint64)
    rc = L.apemost_hip_calibrate_end(s._h, status.ctypes.data_as(C.POINTER(C.c_int32)),
                                     iters.ctypes.data_as(C.POINTER(C.c_uint64)))
    assert rc == capi.ERR_CALIBRATION and np.all(status == -1) and b"cancelled" in L.apemost_hip_last_error()
    part = s.get_state()
    assert np.all(part.ticks % 200 == 0) and np.all(part.ticks >= 600) and np.all(iters % 200 == 0)
    assert L.apemost_hip_calibrate_poll(s._h, C.byref(active)) == capi.ERR_INVALID      # nothing open
    # a fresh calibration on the same sampler equals the oracle's
    s.set_state(st)
    s.set_state(LadderState(n_chain, 4), ("ticks",))
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    for c in range(n_chain):
        assert (int(status[c]), int(iters[c])) == orc.markov_chain_calibrate(lad, rng, c, ocfg)
    assert_match(s.get_state(), lad, rng, what="calibrate after cancel")
    s.close()


def test_data_vector_between_the_two_kernels_lds_limits_is_staged_correctly():
    """n_data = 3100 with the data vector forced into LDS: 15680 + 49600 B for the two-phase kernels is
    under 64 KiB, 16704 + 49600 B for the one-barrier kernels is over it, so only the latter need the
    opt-in for large dynamic LDS (the opt-in was decided on the two-phase footprint alone)"""
    import torch
    w = wl.simplesin(n_data=3100, n_chain=8)
    n_chain = 8
    st, lad, rng = make_pair(w, n_chain, seed=3, init_prob=True)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=3, waves_per_chain=4, lds_policy=1)
    assert s.geometry == (4, True)
    s.set_state(st)
    d = torch.zeros((3 * 9, n_chain, 6), dtype=torch.float64, device="cuda")
    s.run_sampler(3, 9, d.data_ptr())
    s.synchronize()
    ref = orc.run_sampler(lad, rng, 3, 9, record=True)
    assert_match(s.get_state(), lad, rng, what="n_data 3100 in LDS")
    np.testing.assert_allclose(d.cpu().numpy(), ref, rtol=1e-9)
    dcfg, ocfg = _cfgs(burn=200)
    status, iters = s.markov_chain_calibrate(0, 2, dcfg)
    for c in range(2):
        assert (int(status[c]), int(iters[c])) == orc.markov_chain_calibrate(lad, rng, c, ocfg)
    s.close()
