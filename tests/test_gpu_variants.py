"""The reference's compile-time variants of the hot path, carried as apemost_hip_config.flags:
-DPROPOSAL_LOGISTIC / -DPROPOSAL_UNIFORM (get_next_random_jump, src/mcmc_gettersetter.c:290-305),
-DRANDOMSWAP (parallel_tempering_decide_swap_random, src/parallel_tempering_interaction.c:47-64,
130-131) and -DADAPT (adapt(), src/parallel_tempering.c:282-301, 404) -- every kernel that proposes
or swaps, against the oracle run with the same variant."""
import numpy as np
import pytest

from apemost_amd import capi, workloads as wl
from apemost_amd.sampler import HipSampler
from oracle import oracle as orc
from tests.helpers import assert_match, make_pair, small_workloads

pytestmark = pytest.mark.gpu
LAWS = {"logistic": (capi.FLAG_PROPOSAL_LOGISTIC, orc.PROPOSAL_LOGISTIC),
        "uniform": (capi.FLAG_PROPOSAL_UNIFORM, orc.PROPOSAL_UNIFORM)}


def _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=0, **kw):
    import torch
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=seed, waves_per_chain=waves, flags=flags, **kw)
    s.set_state(st)
    d = torch.zeros((n_rounds * n_swap, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
    s.run_sampler(n_rounds, n_swap, d.data_ptr())
    s.synchronize()
    out = s.get_state(), d.cpu().numpy()
    s.close()
    return out


@pytest.mark.parametrize("law", sorted(LAWS))
@pytest.mark.parametrize("name,waves", [("simplesin", 1), ("simplesin", 4), ("simplesin", 8), ("sine3", 2),
                                        ("pulse", 8), ("pulse", 1), ("pulse_vrot", 4), ("pulse_vrot", 8)])
def test_proposal_laws_match_oracle(law, name, waves):
    """run_sampler under a non-Gaussian proposal: the classic kernels (their own candidate refill at
    1-2 waves) and the one-barrier kernel (4 and 8 waves; also forced back to the two-phase step with
    its producer waves) against the oracle's do_step_for with the same law"""
    flag, kind = LAWS[law]
    w = small_workloads()[name]
    n_chain, n_rounds, n_swap, seed = 8, 40, 9, 61
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    lad.proposal = kind
    dev, samples = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=flag)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True, n_threads=8)
    assert_match(dev, lad, rng, what="%s %s waves=%d" % (law, name, waves))
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    assert 0 < dev.accept.sum() < dev.n_iter.sum()
    if waves in (4, 8):
        two, s2 = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=flag | capi.FLAG_TWO_BARRIER_STEP)
        assert np.array_equal(two.params, dev.params) and np.array_equal(s2, samples)


@pytest.mark.parametrize("law", sorted(LAWS))
@pytest.mark.parametrize("waves", [1, 8])
def test_proposal_laws_redraw_and_wrap(law, waves):
    """step widths of six times the prior box: the prepared attempts mostly leave the box and the
    workgroup redraws (attempt indices continue past the prepared ones); with the phase circular
    (-DCIRCULAR_PARAMS) its first usable jump wraps instead.  The logistic law has heavy tails, the
    flat law never rejects an attempt by itself -- both paths differ from the Gaussian's."""
    flag, kind = LAWS[law]
    w = small_workloads()["simplesin"]
    for circular in (0, 1 << 2):
        st, lad, rng = make_pair(w, 4, seed=3)
        st.step[:] = (w.pmax - w.pmin) * 6.0
        lad.step[:] = st.step
        lad.circular, lad.proposal = circular, kind
        dev, samples = _run(w, st, 4, 12, 5, waves, 3, flags=flag, circular_params=circular)
        ref = orc.run_sampler(lad, rng, 12, 5, record=True)
        assert_match(dev, lad, rng, what="%s wide steps waves=%d circular=%d" % (law, waves, circular))
        np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("law", sorted(LAWS))
def test_calibration_under_a_proposal_law_matches_oracle(law):
    """markov_chain_calibrate (burn-in + step-width search) proposes through the same
    get_next_random_jump: the calibration kernel against orc_markov_chain_calibrate"""
    flag, kind = LAWS[law]
    w = small_workloads()["simplesin"]
    n_chain, seed = 4, 17
    cfg = capi.calib_defaults(burn_in_iterations=300, iter_limit=100000)
    ocfg = orc.calib_defaults(burn_in_iterations=300, iter_limit=100000)
    for waves in (1, 4):
        st, lad, rng = make_pair(w, n_chain, seed=seed, init_prob=True)
        lad.proposal = kind
        s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=seed, waves_per_chain=waves, flags=flag)
        s.set_state(st)
        status, iters = s.markov_chain_calibrate(0, n_chain, cfg)
        dev = s.get_state()
        s.close()
        for c in range(n_chain):
            o_status, o_iters = orc.markov_chain_calibrate(lad, rng, c, ocfg)
            assert (status[c], iters[c]) == (o_status, o_iters), (law, waves, c)
        assert_match(dev, lad, rng, what="calibrate %s waves=%d" % (law, waves))


@pytest.mark.parametrize("waves,flags", [(8, 0), (4, 0), (1, 0), (4, capi.FLAG_SINGLE_ROUND_LAUNCHES)])
def test_randomswap_schedule_matches_oracle(waves, flags):
    """-DRANDOMSWAP: one more uniform ahead of the pair choice; in-launch swaps and swaps fused into
    the next launch's start read the same shifted words"""
    w = small_workloads()["simplesin"]
    n_chain, n_rounds, n_swap, seed = 8, 120, 3, 23
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    lad.randomswap = 1
    dev, samples = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=flags | capi.FLAG_RANDOMSWAP)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(dev, lad, rng, what="randomswap waves=%d" % waves)
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    assert dev.swapcount.sum() > 0
    # and it is a different schedule from the default one
    st0, lad0, rng0 = make_pair(w, n_chain, seed=seed)
    orc.run_sampler(lad0, rng0, n_rounds, n_swap)
    assert not np.array_equal(lad0.swapcount, lad.swapcount)
    # the pair the host predicts for a sharded ladder is the one the kernels use
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=seed, flags=capi.FLAG_RANDOMSWAP)
    words = orc.philox_stream(seed, 1 << 63, 8)
    assert s.swap_pair(0) == int(n_chain * 1000 * (words[1] / 2.0 ** 32)) % (n_chain - 1)
    assert s.swap_pair(1) == int(n_chain * 1000 * (words[5] / 2.0 ** 32)) % (n_chain - 1)
    s.close()


@pytest.mark.parametrize("waves", [1, 4, 8])
def test_adapt_nudges_step_widths_like_the_reference(waves):
    """-DADAPT: after 20000 counted updates (summed over the parameters) every round ends with the
    chain's step widths scaled by 0.99 or 1/0.99 towards the target ratio of accepts to REJECTS, and
    past 100000 the counters restart -- both thresholds are crossed here"""
    w = small_workloads()["simplesin"]
    n_chain, n_rounds, n_swap, seed = 8, 420, 70, 41          # 29400 steps x 4 parameters per chain
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    lad.adapt, lad.adapt_target = 1, 0.5
    step0 = st.step.copy()
    dev, samples = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=capi.FLAG_ADAPT, adapt_target=0.5)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True, n_threads=8)
    assert_match(dev, lad, rng, what="adapt waves=%d" % waves)
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    assert not np.allclose(dev.step, step0)                                   # widths moved ...
    assert (dev.accept + dev.reject < dev.n_iter).all()                       # ... and the counters restarted
    # a different target gives a different chain (the knob is wired through)
    st2, lad2, rng2 = make_pair(w, n_chain, seed=seed)
    lad2.adapt, lad2.adapt_target = 1, 0.23
    dev2, _ = _run(w, st2, n_chain, n_rounds, n_swap, waves, seed, flags=capi.FLAG_ADAPT, adapt_target=0.23)
    orc.run_sampler(lad2, rng2, n_rounds, n_swap, n_threads=8)
    assert_match(dev2, lad2, rng2, what="adapt target 0.23 waves=%d" % waves)
    assert not np.allclose(dev2.step, dev.step)


@pytest.mark.parametrize("name,waves", [("simplesin", 1), ("simplesin", 4), ("pulse", 4), ("pulse_vrot", 2)])
def test_rwm_moves_step_widths_like_the_restated_reference(name, waves):
    """-DRWM (src/parallel_tempering.c:268-281 + rmw_adapt_stepwidth, src/markov_chain.c:342-367; the reference's
    own call site does not compile, include/apemost_hip.h states what the engine makes of it): every round ends
    with one more step per chain that moves the counters and the RNG tick but neither n_iter nor the best point
    nor the sample rows, and with every step width moved by U / sqrt(n_iter) (min(1, exp(dprob)) - target) range.
    Against the oracle's restatement (orc_rwm): counters, ticks and n_iter exact, widths and rows 1e-9; together
    with -DADAPT (adapt() runs both blocks, RWM first)."""
    w = small_workloads()[name]
    n_chain, n_rounds, n_swap, seed = 6, 90, 4, 77
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    lad.rwm, lad.adapt_target = 1, 0.5
    step0 = st.step.copy()
    dev, samples = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=capi.FLAG_RWM)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(dev, lad, rng, what="rwm %s waves=%d" % (name, waves))
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    assert np.all(dev.n_iter == n_rounds * n_swap) and np.all(dev.ticks == n_rounds * (n_swap + 1))
    assert np.all(dev.accept + dev.reject == n_rounds * (n_swap + 1))
    assert not np.allclose(dev.step, step0)
    if name == "simplesin" and waves == 4:
        st2, lad2, rng2 = make_pair(w, n_chain, seed=seed)
        lad2.rwm, lad2.adapt, lad2.adapt_target = 1, 1, 0.3
        dev2, _ = _run(w, st2, n_chain, 2300, 3, waves, seed, flags=capi.FLAG_RWM | capi.FLAG_ADAPT, adapt_target=0.3)
        orc.run_sampler(lad2, rng2, 2300, 3)           # 2300 x 4 steps x 4 parameters: past ADAPT's 20000 updates
        assert_match(dev2, lad2, rng2, what="rwm + adapt")


def test_variant_flags_are_validated():
    w = small_workloads()["simplesin"]
    with pytest.raises(capi.ApemostHipError, match="alternatives"):
        HipSampler(w.model, w.n_par, 4, w.data, flags=capi.FLAG_PROPOSAL_LOGISTIC | capi.FLAG_PROPOSAL_UNIFORM)
    with pytest.raises(capi.ApemostHipError, match="1, 2, 4 or 8 waves"):
        HipSampler(w.model, w.n_par, 4, w.data, flags=capi.FLAG_RANDOMSWAP, waves_per_chain=6)
    with pytest.raises(capi.ApemostHipError, match="adapt_target"):
        HipSampler(w.model, w.n_par, 4, w.data, flags=capi.FLAG_ADAPT, adapt_target=-1.0)


def test_samples_packed_on_the_device_equal_the_rows_repacked_on_the_host():
    """apemost_hip_samples_pack_read_async: the record of the C host's binary sink (layout 0: the
    parameter vectors of the first n_param_chains chains, then (prob, prob - prior) of every chain)
    and thinned rows (layout 1), for every skip / thin / n_param_chains combination tried, against
    numpy on the rows read back whole"""
    import ctypes as C
    import torch
    from apemost_amd import capi, workloads as wl
    from apemost_amd.sampler import HipSampler
    from tests.helpers import make_pair
    n_chain, n_steps = 5, 37
    w = wl.pulse(n_data=64, n_chain=n_chain)
    st, _, _ = make_pair(w, n_chain, seed=8, init_prob=True)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=8)
    s.set_state(st)
    npar = w.n_par
    rows = torch.zeros((n_steps, n_chain, npar + 2), dtype=torch.float64, device="cuda")
    s.launch_round(n_steps, False, rows.data_ptr())
    s.synchronize()
    ref = rows.cpu().numpy()
    L = s.L
    host = C.c_void_p()
    capi.check(L.apemost_hip_host_alloc(ref.nbytes, C.byref(host)))
    packed = torch.zeros(ref.size, dtype=torch.float64, device="cuda")
    counters = np.zeros(2 * n_chain + npar, dtype=np.uint64)    # + chain 0's point after the last step, as doubles
    for layout, npc, skip, thin in ((0, 1, 0, 1), (0, n_chain, 2, 5), (0, 0, 36, 40), (0, 3, 0, 7), (1, 0, 0, 1), (1, 0, 4, 9), (1, 0, 37, 3)):
        kept = C.c_uint64(0)
        capi.check(L.apemost_hip_samples_pack_read_async(s._h, rows.data_ptr(), n_steps, skip, thin, npc, layout, packed.data_ptr(),
                                                         host, counters.ctypes.data_as(C.c_void_p), C.byref(kept)))
        capi.check(L.apemost_hip_samples_wait(s._h))
        steps = list(range(skip, n_steps, thin))
        assert kept.value == len(steps)
        if layout == 0:
            want = np.array([np.concatenate([ref[k, :npc, :npar].ravel(), ref[k, :, npar:].ravel()]) for k in steps]).ravel()
        else:
            want = ref[steps].ravel()
        got = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_double)), shape=(ref.size,))[:want.size]
        assert np.array_equal(got, want), (layout, npc, skip, thin)
        dev = s.get_state()
        assert np.array_equal(counters[:n_chain], dev.accept) and np.array_equal(counters[n_chain:2 * n_chain], dev.reject)
        # (what the C host's progress line prints: the kept steps may end earlier, or be none -- skip 37 of 37)
        assert np.array_equal(counters[2 * n_chain:].view(np.float64), ref[-1, 0, :npar])
    assert L.apemost_hip_samples_pack_read_async(s._h, rows.data_ptr(), n_steps, 0, 0, 0, 0, packed.data_ptr(), host, None,
                                                 None) == capi.ERR_INVALID          # thin 0
    capi.check(L.apemost_hip_host_free(host))
    s.close()
