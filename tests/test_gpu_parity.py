"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same seeded
inputs.  Tolerances: integer state (counters, RNG positions, swap counts, pair indices) bit-exact;
fp64 log-likelihoods rel 1e-12; trajectories rel 1e-9 (ulp-level sin/log/summation
differences accumulate as a random walk over the steps)."""
import ctypes as C
import os

import numpy as np
import pytest

from apemost_amd import capi, workloads as wl
from apemost_amd.sampler import HipSampler
from oracle import oracle as orc
from tests.helpers import assert_match, make_pair, small_workloads

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def test_device_is_gfx950():
    name, cus, mem = capi.device_info(0)
    assert name.startswith("gfx950") and cus >= 200 and mem > 200e9


@pytest.mark.parametrize("seed,sub,off", [(0, 0, 0), (4357, 3 * 256 + 2, 5), (2 ** 63 + 11, 2 ** 63, 4 * 1000 + 1),
                                          (0xdeadbeefdeadbeef, 2 ** 40 + 7, 2 ** 34 + 3)])
def test_rocrand_stream_equals_oracle_philox(seed, sub, off):
    dev = capi.rng_raw(seed, sub, off, 41)
    ref = orc.philox_stream(seed, sub, 41, start=off)
    assert np.array_equal(dev, ref)


def test_gaussian_attempts_match_oracle():
    seed, chain, slot, tick = 77, 5, 1, 123456789
    y, s, valid, _ = capi.rng_attempts(seed, chain, slot, tick, 3, 400)
    ref = [orc.gaussian_attempt(seed, chain, slot, tick, 3 + i) for i in range(400)]
    assert list(valid) == [r[0] for r in ref]
    assert 0.7 < valid.mean() < 0.87                       # polar acceptance pi/4
    ok = valid
    np.testing.assert_allclose(y[ok], [r[1] for r in ref if r[0]], rtol=0)   # exact: integer -> fp64
    np.testing.assert_allclose(s[ok], [r[2] for r in ref if r[0]], rtol=1e-14)
    for t in (0, 1, 2 ** 30 + 17):
        _, _, _, lu = capi.rng_attempts(seed, chain, 4, t, 0, 1)
        assert abs(lu - orc.accept_log_uniform(seed, chain, 4, t)) <= 1e-14 * abs(lu)


def test_swap_pair_host_matches_oracle():
    for n in (2, 8, 128, 2048, 16384):
        for r in range(50):
            u = orc.philox_stream(5, 2 ** 63, 1, start=4 * r)[0] / 2 ** 32
            assert capi.swap_pair(5, r, n) == orc.lib().orc_swap_pair_index(u, n)
    assert capi.swap_pair(5, 0, 1) == -1


@pytest.mark.parametrize("waves", [1, 2, 4, 6, 8])
def test_loglike_golden_vectors(waves, golden_dir):
    # doc/manual.rst:190-213
    data = np.array([[101, 0.67], [102, 1.01], [103, 7.9e-1], [104, 1.34]])
    s = HipSampler(wl.MODEL_SIMPLESIN, 4, 1, data, waves_per_chain=waves)
    prob, prior = s.loglike([[1, 0.2, 1, 0]], 1.0)
    assert abs(prob[0] - (-1.480898044165363e+01)) < 1e-12 * 14.8 and prior[0] == 0
    s.close()
    # SURVEY 8(c) G1 on the reference's own light curve (tests/testlc.dat)
    d = np.loadtxt(os.path.join(golden_dir, "testlc.dat"))
    s = HipSampler(wl.MODEL_SIMPLESIN, 4, 1, d, waves_per_chain=waves)
    prob, _ = s.loglike([[0.7, 5.2, 5.4, 0.0], [1.0, 10.0, 0.25, 0.1]], 1.0)
    np.testing.assert_allclose(prob, [-3.187159885063915e+03, -3.271718790429034e+03], rtol=1e-12)
    s.close()


@pytest.mark.parametrize("name", ["simplesin", "sine3", "pulse", "pulse_vrot"])
@pytest.mark.parametrize("waves", [1, 4, 6])
def test_loglike_matches_oracle(name, waves):
    w = small_workloads()[name]
    rs = np.random.RandomState(3)
    n = 37
    params = w.pmin + (w.pmax - w.pmin) * rs.uniform(0.05, 0.95, size=(n, w.n_par))
    beta = rs.uniform(0.01, 1.0, n)
    s = HipSampler(w.model, w.n_par, 2, w.data, waves_per_chain=waves)
    prob, prior = s.loglike(params, beta)
    ref = [orc.loglike(w.model, params[i], w.data, beta=beta[i]) for i in range(n)]
    np.testing.assert_allclose(prob, [r[0] for r in ref], rtol=1e-12)
    np.testing.assert_allclose(prior, [r[1] for r in ref], rtol=1e-13, atol=1e-300)
    s.close()


@pytest.mark.parametrize("name,n_data", [("sine3", 8192), ("pulse", 1024), ("pulse_vrot", 65536)])
@pytest.mark.parametrize("waves", [1, 4])
def test_loglike_matches_oracle_at_full_size(name, n_data, waves):
    """test_loglike_matches_oracle on the data vectors of BASELINE configs 3, 4 and 5 (8192 / 1024 / 65 536
    points): the stated 1e-12 where rounding has the most terms to grow over -- ln of a running product, one
    reciprocal per two or four points, FMAs."""
    w = wl.by_name(name, n_data=n_data, n_chain=2)
    rs = np.random.RandomState(31)
    n = 37
    params = w.pmin + (w.pmax - w.pmin) * rs.uniform(0.05, 0.95, size=(n, w.n_par))
    beta = rs.uniform(0.01, 1.0, n)
    s = HipSampler(w.model, w.n_par, 2, w.data, waves_per_chain=waves)
    prob, prior = s.loglike(params, beta)
    ref = [orc.loglike(w.model, params[i], w.data, beta=beta[i]) for i in range(n)]
    np.testing.assert_allclose(prob, [r[0] for r in ref], rtol=1e-12)
    np.testing.assert_allclose(prior, [r[1] for r in ref], rtol=1e-13, atol=1e-300)
    s.close()


@pytest.mark.parametrize("name", ["pulse", "pulse_vrot"])
def test_pulse_loglike_over_a_wide_range_of_units(name):
    """The pulse likelihoods take sum ln y as the logarithm of a running product split into mantissa and
    exponent, and the quotients d / y two points at a time over a common denominator (pt_device.h
    LogProdT, terms_lp): products of pairs of numerators and denominators must stay inside fp64's range --
    and where they do not, the lane keeps the smallest exponent they reach, a non-finite sum tells the
    rest, and the wave takes its share again in the reference's own operation order (Model::term_ref).
    The same spectrum in other units -- frequencies in units 1e6 and 1e30 times smaller or larger (the
    lifetime scaled with them: the reference's 2 pi (f - nu) tau has no unit, the kernel's c + (f - nu)^2
    has), heights and data 1e60 and 1e200 times smaller or larger -- against the oracle, which adds a
    gsl_sf_log per point: 1e-12 as everywhere, ragged length, one and four waves per chain.  (Heights
    1e-200 with frequencies 1e30: pulse_vrot's denominator product overflows while the numerators stay
    finite -- ADVICE r3's case.)"""
    # (1100 points: four passes of the one-wave kernel's long loop -- where pulse_vrot shares a reciprocal
    # among FOUR points and takes the sum again when their products leave the range, as they do at the
    # largest heights here -- and a ragged rest)
    w = wl.pulse(n_data=1100, n_chain=2) if name == "pulse" else wl.pulse_vrot(n_data=1100, n_chain=2)
    rs = np.random.RandomState(5)
    base = w.pmin + (w.pmax - w.pmin) * rs.uniform(0.2, 0.8, size=(5, w.n_par))
    beta = rs.uniform(0.05, 1.0, len(base))
    heights = [3, 5] if name == "pulse" else [4, 6]
    freqs = [2, 4] if name == "pulse" else [2, 3, 5]                  # (pulse_vrot: vrot, fa, fb)
    for sf in (1e-30, 1e-6, 1.0, 1e6, 1e30):
        for sh in (1e-200, 1e-60, 1.0, 1e60, 1e200):
            data = w.data.copy()
            data[:, 0] *= sf
            data[:, 1] *= sh
            params = base.copy()
            params[:, 0] /= sf                                         # lifetime: 2 pi (f - nu) tau keeps its value
            params[:, freqs] *= sf
            params[:, heights] *= sh
            for waves in (1, 4):
                s = HipSampler(w.model, w.n_par, 2, data, waves_per_chain=waves)
                prob, prior = s.loglike(params, beta)
                ref = [orc.loglike(w.model, params[i], data, beta=beta[i]) for i in range(len(params))]
                assert np.all(np.isfinite(prob)), (sf, sh, waves)
                np.testing.assert_allclose(prob, [r[0] for r in ref], rtol=1e-12, err_msg="%g %g %d" % (sf, sh, waves))
                np.testing.assert_allclose(prior, [r[1] for r in ref], rtol=1e-13, atol=1e-300)
                s.close()


def test_loglike_ragged_and_tiny_inputs():
    # n_data not a multiple of the wavefront, and smaller than one wavefront
    for n_data in (1, 3, 63, 65, 1000):
        w = wl.simplesin(n_data=n_data, n_chain=2)
        for waves in (1, 8):
            s = HipSampler(w.model, 4, 2, w.data, waves_per_chain=waves)
            prob, _ = s.loglike([w.start], 0.5)
            ref, _ = orc.loglike(w.model, w.start, w.data, beta=0.5)
            assert abs(prob[0] - ref) <= 1e-12 * abs(ref)
            s.close()


def _run_both(w, n_chain, n_rounds, n_swap, waves, seed=42, init_prob=False):
    torch = _torch()
    st, lad, rng = make_pair(w, n_chain, seed=seed, init_prob=init_prob)
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=seed, waves_per_chain=waves)
    s.set_state(st)
    d_samples = torch.zeros((n_rounds * n_swap, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
    s.run_sampler(n_rounds, n_swap, d_samples.data_ptr())
    s.synchronize()
    dev = s.get_state()
    ref_samples = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    return s, dev, d_samples.cpu().numpy(), lad, rng, ref_samples


@pytest.mark.parametrize("name", ["simplesin", "sine3", "pulse", "pulse_vrot"])
@pytest.mark.parametrize("waves", [1, 2, 4])
def test_trajectory_matches_oracle(name, waves):
    """every recorded step of every chain: same accept decisions, same swaps, same RNG positions"""
    w = small_workloads()[name]
    n_rounds, n_swap = 40, 25
    s, dev, samples, lad, rng, ref = _run_both(w, 8, n_rounds, n_swap, waves)
    assert_match(dev, lad, rng, what=name)
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    assert dev.swapcount.sum() > 0                      # swaps were exercised
    assert 0 < dev.accept.sum() < dev.n_iter.sum()      # both accept and reject were exercised
    r, pending = s.round
    assert r == n_rounds and not pending
    s.close()


def test_first_step_always_accepted_quirk_q2():
    # prob = -1e10 after read_calibration_file => first proposal accepted in every chain
    w = small_workloads()["simplesin"]
    s, dev, samples, lad, rng, ref = _run_both(w, 8, 1, 1, 1)
    assert list(dev.accept) == [1] * 8 and list(dev.reject) == [0] * 8
    assert_match(dev, lad, rng)
    s.close()


def test_run_in_pieces_equals_one_run():
    """launch boundaries do not change the trajectory: 10 rounds == 4 + 6 rounds"""
    torch = _torch()
    w = small_workloads()["simplesin"]
    st, _, _ = make_pair(w, 8)
    outs = []
    for split in ((10,), (4, 6)):
        s = HipSampler(w.model, 4, 8, w.data, seed=9)
        s.set_state(st)
        for k in split:
            s.run_sampler(k, 7)
        s.synchronize()
        outs.append(s.get_state())
        s.close()
    for n in ("params", "prob", "prob_best", "params_best", "accept", "swapcount", "ticks"):
        assert np.array_equal(getattr(outs[0], n), getattr(outs[1], n)), n


def test_full_size_config2_properties():
    """BASELINE config 2 (128 chains x 1024 points): size-independent properties"""
    torch = _torch()
    w = wl.simplesin(n_data=1024, n_chain=128)
    st, lad, rng = make_pair(w, 128, seed=7)
    n_rounds, n_swap = 20, w.n_swap
    outs = []
    for rep in range(2):
        s = HipSampler(w.model, 4, 128, w.data, seed=7)
        s.set_state(st)
        d = torch.zeros((n_rounds * n_swap, 128, 6), dtype=torch.float64, device="cuda")
        s.run_sampler(n_rounds, n_swap, d.data_ptr())
        s.synchronize()
        outs.append((s.get_state(), d.cpu().numpy()))
        s.close()
    (a, sa), (b, sb) = outs
    assert np.array_equal(sa, sb) and np.array_equal(a.params, b.params)      # run-to-run identical
    assert np.all(a.accept + a.reject == n_rounds * n_swap) and np.all(a.n_iter == n_rounds * n_swap)
    assert np.all(sa[..., :4] >= w.pmin) and np.all(sa[..., :4] <= w.pmax)   # proposals stay in bounds
    assert np.all(a.prob_best >= a.prob)
    assert a.swapcount.sum() <= n_rounds
    # hot chains accept more than cold ones
    assert a.accept[-8:].mean() > a.accept[:8].mean()
    # and the oracle agrees on the whole thing
    orc.run_sampler(lad, rng, n_rounds, n_swap, n_threads=8)
    assert_match(a, lad, rng, what="config2")


@pytest.mark.parametrize("split", [4, 3])
def test_two_device_shards_equal_whole_ladder(split):
    """edge_export / edge_import / halo swap-in on the device: a ladder cut into two shards (both on
    this GPU, driven in lock step like two ranks would) must reproduce the whole-ladder run bit for bit"""
    torch = _torch()
    from apemost_amd.distributed import HipShardEngine
    w = small_workloads()["pulse"]
    n_global, n_rounds, n_swap, seed = 8, 80, 3, 17
    st, _, _ = make_pair(w, n_global, seed=seed)
    whole = HipSampler(w.model, w.n_par, n_global, w.data, seed=seed)
    whole.set_state(st)
    whole.run_sampler(n_rounds, n_swap)
    whole.synchronize()
    ref = whole.get_state()
    whole.close()

    bounds = [(0, split), (split, n_global)]
    shards = []
    for lo, hi in bounds:
        s = HipSampler(w.model, w.n_par, hi - lo, w.data, seed=seed, chain_offset=lo, n_chains_global=n_global)
        s.set_state(st.slice(lo, hi))
        shards.append(HipShardEngine(s, torch))
    exchanges, pending, rnd = 0, False, 0
    for r in range(n_rounds + 1):
        n_steps = n_swap if r < n_rounds else 0
        if pending:
            a = shards[0].swap_pair(rnd)
            assert a == shards[1].swap_pair(rnd)
            if a == split - 1:
                up, down = shards[0].edge_export(1), shards[1].edge_export(0)
                for e in shards:
                    e.s.synchronize()
                shards[0].edge_import(1, down)
                shards[1].edge_import(0, up)
                exchanges += 1
            rnd += 1
        for e in shards:
            e.launch_rounds(1, n_steps, pending, None)
        pending = n_steps > 0
    got = [e.s.get_state() for e in shards]
    for e in shards:
        e.s.close()
    assert exchanges > 0
    for f in ("params", "params_best", "prob", "prob_best", "prior", "accept", "reject", "swapcount", "ticks",
              "n_iter"):
        assert np.array_equal(np.concatenate([getattr(g, f) for g in got]), getattr(ref, f)), f
    assert ref.swapcount.sum() > 0


def test_sharded_driver_on_one_gpu_batches_rounds_and_matches_oracle():
    """ShardedLadder (world 1) batches rounds into multi-round launches; the in-launch swap hand-off
    between workgroups must reproduce the oracle's swaps exactly"""
    torch = _torch()
    from apemost_amd.distributed import HipShardEngine, ShardedLadder
    w = wl.simplesin(n_data=128, n_chain=32)
    n_rounds, n_swap, seed = 150, 4, 77
    st, lad, rng = make_pair(w, 32, seed=seed)
    s = HipSampler(w.model, 4, 32, w.data, seed=seed)
    s.set_state(st)
    assert s.max_rounds_per_launch > 1
    d = torch.zeros((n_rounds, n_swap, 32, 6), dtype=torch.float64, device="cuda")
    ladder = ShardedLadder(HipShardEngine(s, torch), 32, 0, 32, 0, 1, None)
    ladder.run_sampler(n_rounds, n_swap, d)
    s.synchronize()
    dev = s.get_state()
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(dev, lad, rng, what="batched rounds")
    np.testing.assert_allclose(d.cpu().numpy().reshape(ref.shape), ref, rtol=1e-9)
    assert dev.swapcount.sum() > 10
    s.close()


@pytest.mark.parametrize("n_chain,n_data,flags", [(32, 128, capi.FLAG_SINGLE_ROUND_LAUNCHES), (4096, 64, 0),
                                                  (32, 128, capi.FLAG_COOPERATIVE_LAUNCH), (512, 1024, 0),
                                                  (32, 128, capi.FLAG_COOPERATIVE_LAUNCH | capi.FLAG_TEST_REFUSE_COOPERATIVE),
                                                  (512, 1024, capi.FLAG_TEST_REFUSE_COOPERATIVE)])
def test_launch_policies_match_oracle(n_chain, n_data, flags):
    """The ways a run is cut into launches give the oracle's chain: (i) one round per launch
    forced by flag -- every swap is the fused swap-in at launch start; (ii) the same fallback taken
    by the engine itself because 4096 workgroups cannot be co-resident; (iii) multi-round launches
    placed by hipLaunchCooperativeKernel on request; (iv) the same chosen by the engine where the
    grid fits the occupancy figure but not its cautious estimate (512 workgroups of 8 waves, two per
    CU)."""
    torch = _torch()
    from apemost_amd.distributed import HipShardEngine, ShardedLadder
    w = wl.simplesin(n_data=n_data, n_chain=n_chain)
    n_rounds, n_swap, seed = (150, 4, 77) if n_chain < 100 else (12, 2, 78)
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=seed, flags=flags)
    s.set_state(st)
    refuse = bool(flags & capi.FLAG_TEST_REFUSE_COOPERATIVE)   # (v) a refused cooperative launch: re-issued round by round
    if (flags & capi.FLAG_COOPERATIVE_LAUNCH) or n_chain == 512:
        assert s.max_rounds_per_launch > 1
    else:
        assert s.max_rounds_per_launch == 1
    d = torch.zeros((n_rounds, n_swap, n_chain, 6), dtype=torch.float64, device="cuda")
    if n_chain < 100:
        ladder = ShardedLadder(HipShardEngine(s, torch), n_chain, 0, n_chain, 0, 1, None)
        ladder.run_sampler(n_rounds, n_swap, d)
    else:
        s.run_sampler(n_rounds, n_swap, d.data_ptr())
    s.synchronize()
    dev = s.get_state()
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True, n_threads=8)
    assert_match(dev, lad, rng, what="launch policy")
    np.testing.assert_allclose(d.cpu().numpy().reshape(ref.shape), ref, rtol=1e-9)
    assert dev.swapcount.sum() > (10 if n_chain < 100 else 0)
    if n_chain == 512 and not refuse:
        assert s.max_rounds_per_launch > 1     # the runtime did place the grid
    if refuse:
        assert s.max_rounds_per_launch == 1    # ... and after a refusal every launch holds one round
    s.close()


@pytest.mark.parametrize("name", ["pulse", "pulse_vrot"])
def test_all_heights_zero_is_nan_as_in_the_reference(name):
    """y = 0 at every data point: the reference's term is ln 0 + d / 0 = -inf + inf = NaN.  The device
    logarithm has no zero case of its own in these likelihoods (log_tab_nz): the quotient's
    reciprocal makes the NaN, in the one-wave kernels (ocml logarithm for pulse) and the others"""
    w = wl.by_name(name, n_data=300, n_chain=4)
    p = np.array(w.start, float)
    if name == "pulse":
        p[3::2] = 0.0
    else:
        p[4] = p[6] = 0.0
    assert np.isnan(orc.loglike(w.model, p, w.data, 0.7)[0])
    for waves in (1, 4, 8):
        s = HipSampler(w.model, w.n_par, 1, w.data, waves_per_chain=waves)
        prob, _ = s.loglike(p[None, :], np.array([0.7]))
        assert np.isnan(prob[0]), (name, waves)
        s.close()


@pytest.mark.parametrize("n_modes", [1, 2, 3, 4])
def test_pulse_few_modes_paths_match_oracle(n_modes):
    """Model<PULSE>::term evaluates one to three modes side by side (few_modes<M>) and more in a loop:
    log-likelihood and a short run against the oracle for each path, one-wave and one-barrier kernels"""
    rs = np.random.RandomState(3 + n_modes)
    nu = np.linspace(10, 12, 300)
    modes = [(10.3 + 0.4 * k, 3.0 - 0.5 * k) for k in range(n_modes)]
    y = sum(h / (1 + (2 * np.pi * (f - nu) * 5.0) ** 2) for f, h in modes) + 0.05
    data = np.stack([nu, y * rs.exponential(1.0, len(nu))], 1)

    class W:
        pass
    w = W()
    w.model, w.n_par, w.data = wl.MODEL_PULSE, 2 + 2 * n_modes, data
    w.start = np.array([5.0, 0.05] + [v for f, h in modes for v in (f, h)])
    w.pmin = np.array([0.1, 0] + [v for _ in modes for v in (10, 0)], float)
    w.pmax = np.array([50, 1] + [v for _ in modes for v in (12, 20)], float)
    w.step = (w.pmax - w.pmin) * 0.02
    for waves in (1, 4):
        s = HipSampler(w.model, w.n_par, 1, w.data, waves_per_chain=waves)
        pts = w.pmin + (w.pmax - w.pmin) * rs.uniform(0.1, 0.9, (5, w.n_par))
        prob, prior = s.loglike(pts, np.full(5, 0.8))
        for k in range(5):
            ref, ref_prior = orc.loglike(w.model, pts[k], w.data, 0.8)
            assert abs(prob[k] - ref) <= 1e-12 * abs(ref) and abs(prior[k] - ref_prior) <= 1e-12 * abs(ref_prior)
        s.close()
        st, lad, rng = make_pair(w, 4, seed=21)
        s = HipSampler(w.model, w.n_par, 4, w.data, seed=21, waves_per_chain=waves)
        s.set_state(st)
        s.run_sampler(20, 5)
        s.synchronize()
        dev = s.get_state()
        s.close()
        orc.run_sampler(lad, rng, 20, 5)
        assert_match(dev, lad, rng, what="pulse %d modes waves %d" % (n_modes, waves))


def test_rounds_within_shard_stops_at_the_first_straddling_pair():
    """apemost_hip_rounds_within_shard (what a sharded ladder may put into one launch) against the
    pairs apemost_hip_sampler_swap_pair predicts one by one"""
    w = small_workloads()["simplesin"]
    s = HipSampler(w.model, 4, 4, w.data, seed=5, chain_offset=4, n_chains_global=12)
    seen = set()
    for first in (0, 7, 100, 1000):
        k = s.rounds_within_shard(first, 50)
        straddle = [s.swap_pair(first + i) in (3, 7) for i in range(50)]
        assert k == (straddle.index(True) if True in straddle else 50)
        seen.add(k)
    assert len(seen) > 1 and s.rounds_within_shard(0, 0) == 0
    s.close()
    whole = HipSampler(w.model, 4, 12, w.data, seed=5)
    assert whole.rounds_within_shard(3, 77) == 77          # no edge to straddle
    whole.close()


def test_impossible_prior_box_is_refused_not_spun_on():
    """min > max can never be hit by the redraw loop (src/markov_chain.c:235-240 would spin on the
    host, a kernel on the GPU): set_state refuses it"""
    w = small_workloads()["simplesin"]
    st, _, _ = make_pair(w, 4)
    s = HipSampler(w.model, 4, 4, w.data)
    s.set_state(st)
    bad = st.copy()
    bad.pmin[2, 1], bad.pmax[2, 1] = 1.0, 0.5
    with pytest.raises(capi.ApemostHipError, match="chain 2 parameter 1"):
        s.set_state(bad)
    with pytest.raises(capi.ApemostHipError, match="min"):
        s.set_state(bad, ("pmin",))          # one-sided view is checked against what the device holds
    bad.pmax[0, 0] = np.nan
    with pytest.raises(capi.ApemostHipError):
        s.set_state(bad, ("pmax",))
    s.run_sampler(3, 2)                      # the sampler still holds the good state
    s.synchronize()
    assert np.all(s.get_state().ticks == 6)
    s.close()


def test_single_chain_ladder_has_no_swaps():
    w = small_workloads()["simplesin"]
    s, dev, samples, lad, rng, ref = _run_both(w, 1, 10, 9, 2)
    assert_match(dev, lad, rng, what="single chain")
    assert dev.swapcount[0] == 0 and s.round == (10, False)
    s.close()


def test_pulse_with_five_modes_twelve_parameters():
    """generic mode count (apps/pulse.c loops over n_par): 63/12 = 5 attempt lanes per parameter"""
    torch = _torch()
    rs = np.random.RandomState(5)
    nu = np.linspace(10, 12, 150)
    modes = [(10.2 + 0.35 * k, 1.0 + k) for k in range(5)]
    y = sum(h / (1 + (2 * np.pi * (f - nu) * 4.0) ** 2) for f, h in modes) + 0.05
    data = np.stack([nu, y * rs.exponential(1.0, 150)], 1)
    start = [4.0, 0.05] + [v for f, h in modes for v in (f, h)]
    pmin = [0.1, 0] + [v for _ in modes for v in (10, 0)]
    pmax = [50, 1] + [v for _ in modes for v in (12, 20)]

    class W:
        pass
    w = W()
    w.model, w.n_par, w.data = wl.MODEL_PULSE, 12, data
    w.start, w.pmin, w.pmax = np.array(start), np.array(pmin, float), np.array(pmax, float)
    w.step = (w.pmax - w.pmin) * 0.1
    s, dev, samples, lad, rng, ref = _run_both(w, 6, 30, 5, 4)
    assert_match(dev, lad, rng, what="pulse 5 modes")
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    s.close()


def test_proposals_far_wider_than_the_prior_box_use_the_wave_parallel_redraw():
    """step = 6 x range: nearly every prepared attempt leaves [min,max], so the 64-at-a-time redraw
    path of the proposal runs constantly; attempt indices must still match the oracle's serial loop"""
    torch = _torch()
    w = small_workloads()["simplesin"]
    st, lad, rng = make_pair(w, 4, seed=3)
    st.step[:] = (w.pmax - w.pmin) * 6.0
    lad.step[:] = st.step
    for waves in (1, 8):
        s = HipSampler(w.model, 4, 4, w.data, seed=3, waves_per_chain=waves)
        s.set_state(st)
        d = torch.zeros((60, 4, 6), dtype=torch.float64, device="cuda")
        s.run_sampler(12, 5, d.data_ptr())
        s.synchronize()
        dev = s.get_state()
        lad2 = orc.Ladder(w.model, 4, 4, w.data)
        from tests.helpers import to_oracle
        to_oracle(st, lad2)
        rng2 = orc.Rng(orc.RNG_STREAMS, 3, lad2)
        ref = orc.run_sampler(lad2, rng2, 12, 5, record=True)
        assert_match(dev, lad2, rng2, what="wide steps waves=%d" % waves)
        np.testing.assert_allclose(d.cpu().numpy(), ref, rtol=1e-9)
        s.close()


def test_sine_argument_range_guard():
    """|2 pi (f x + phi)| >= 2^45: the phase has no significant bits; the engine returns NaN (and the
    accept test then rejects) instead of a meaningless sine.  Large but representable arguments
    (Julian-date sized x) still agree with libm to the usual tolerance."""
    w = wl.simplesin(n_data=50, n_chain=2)
    big = w.data.copy()
    big[:, 0] += 2.45e6                       # x ~ JD
    s = HipSampler(w.model, 4, 2, big)
    p = [0.9, 0.27, 0.4, 0.5]
    prob, _ = s.loglike([p], 1.0)
    ref, _ = orc.loglike(w.model, p, big)
    assert abs(prob[0] - ref) <= 1e-12 * abs(ref)
    s.close()
    huge = w.data.copy()
    huge[3, 0] = 1e14
    s = HipSampler(w.model, 4, 2, huge)
    prob, _ = s.loglike([p], 1.0)
    assert np.isnan(prob[0])
    s.close()


def test_maximum_parameter_count():
    """n_par = 62 (APEMOST_HIP_MAX_PAR): one attempt lane per parameter, pulse with 30 modes"""
    torch = _torch()
    rs = np.random.RandomState(9)
    n_modes = 30
    nu = np.linspace(10, 12, 90)
    modes = [(10.03 + 0.065 * k, 0.5 + 0.1 * k) for k in range(n_modes)]
    y = sum(h / (1 + (2 * np.pi * (f - nu) * 4.0) ** 2) for f, h in modes) + 0.05
    data = np.stack([nu, y * rs.exponential(1.0, len(nu))], 1)

    class W:
        pass
    w = W()
    w.model, w.n_par, w.data = wl.MODEL_PULSE, 2 + 2 * n_modes, data
    w.start = np.array([4.0, 0.05] + [v for f, h in modes for v in (f, h)])
    w.pmin = np.array([0.1, 0] + [v for _ in modes for v in (10, 0)], float)
    w.pmax = np.array([50, 1] + [v for _ in modes for v in (12, 20)], float)
    w.step = (w.pmax - w.pmin) * 0.02
    s, dev, samples, lad, rng, ref = _run_both(w, 3, 6, 4, 2)
    assert_match(dev, lad, rng, what="62 parameters")
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    s.close()


def test_posterior_and_acceptance_agree_with_reference_stream_mode():
    """Statistical gate: the engine's tick-addressed Philox draws and the
    reference's single sequential mt19937 stream are different random numbers, so trajectories
    differ, but the sampled law must be the same: per-chain acceptance rates and the cold chain's
    posterior means / standard deviations agree within Monte-Carlo error."""
    torch = _torch()
    w = wl.simplesin(n_data=256, n_chain=8)
    # a prior box around the main mode: the full box has alias modes in frequency that a chain
    # visits or not depending on a handful of swaps, which no 40 000-step statistic can pin down
    w.pmin = np.array([0.5, 0.199, 0.3, 0.2])
    w.pmax = np.array([1.5, 0.201, 0.5, 0.8])
    w.step = (w.pmax - w.pmin) * 0.1
    n_rounds, n_swap, burn = 160, 250, 5000          # 40 000 steps per chain
    st, lad, _ = make_pair(w, 8, seed=1)
    s = HipSampler(w.model, 4, 8, w.data, seed=1)
    s.set_state(st)
    d = torch.zeros((n_rounds * n_swap, 8, 6), dtype=torch.float64, device="cuda")
    s.run_sampler(n_rounds, n_swap, d.data_ptr())
    s.synchronize()
    dev = s.get_state()
    gpu = d.cpu().numpy()
    s.close()
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)              # the reference's RNG mode
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    n = n_rounds * n_swap
    acc_gpu, acc_ref = dev.accept / n, lad.accept / n
    assert np.all(np.abs(acc_gpu - acc_ref) < 0.03), (acc_gpu, acc_ref)

    def batch_stats(x, nb=25):
        b = x[: len(x) // nb * nb].reshape(nb, -1).mean(1)
        return x.mean(), b.std(ddof=1) / np.sqrt(nb)

    for p in range(4):
        g, r = gpu[burn:, 0, p], ref[burn:, 0, p]
        (mg, eg), (mr, er) = batch_stats(g), batch_stats(r)
        assert abs(mg - mr) < 5 * np.hypot(eg, er) + 1e-12, (p, mg, mr, eg, er)
        assert abs(g.std() - r.std()) < 0.25 * r.std(), (p, g.std(), r.std())


@pytest.mark.parametrize("name,n_chain,n_data,n_swap,n_rounds", [
    ("sine3", 1024, 8192, 1, 6),          # BASELINE config 3
    ("pulse", 2048, 1024, 1, 40),         # config 4, one GPU's worth of chains at full ladder size
    ("pulse_vrot", 256, 65536, 1, 3),     # config 5 data vector (1 MiB, read through L2), fewer chains
])
def test_large_configs_size_independent_properties(name, n_chain, n_data, n_swap, n_rounds):
    """full-size BASELINE shapes: run-to-run bit identity, counter conservation, bounds, and agreement
    with the oracle on a subset of chains (the oracle is too slow for all of them)"""
    torch = _torch()
    w = wl.by_name(name, n_data=n_data, n_chain=n_chain)
    st, _, _ = make_pair(w, n_chain, seed=13)
    outs = []
    for rep in range(2):
        s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=13)
        s.set_state(st)
        d = torch.zeros((n_rounds * n_swap, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
        s.run_sampler(n_rounds, n_swap, d.data_ptr())
        s.synchronize()
        outs.append((s.get_state(), d.cpu().numpy()))
        s.close()
    (a, sa), (b, sb) = outs
    assert np.array_equal(sa, sb) and np.array_equal(a.params, b.params) and np.array_equal(a.prob, b.prob)
    n = n_rounds * n_swap
    assert np.all(a.accept + a.reject == n) and np.all(a.ticks == n) and np.all(a.n_iter == n)
    assert np.all(sa[..., :w.n_par] >= w.pmin) and np.all(sa[..., :w.n_par] <= w.pmax)
    assert np.all(np.isfinite(sa)) and a.swapcount.sum() <= n_rounds
    # oracle check without swaps influence: recompute the final log-posterior of a few chains
    for c in (0, n_chain // 2, n_chain - 1):
        if a.accept[c] == 0:
            continue
        prob, prior = orc.loglike(w.model, a.params[c], w.data, beta=a.beta[c])
        # prob belongs to the last accepted point = current params unless a swap moved params since
        if a.swapcount[max(c - 1, 0):c + 1].sum() == 0:
            assert abs(a.prob[c] - prob) <= 1e-12 * abs(prob), (c, a.prob[c], prob)


def test_circular_parameters_wrap_like_the_reference():
    """-DCIRCULAR_PARAMS: a circular parameter (the phase) wraps around its range instead of being
    redrawn (src/markov_chain.c:241-265); same attempt indices and wrapped values as the oracle"""
    torch = _torch()
    w = small_workloads()["simplesin"]
    st, lad, rng = make_pair(w, 6, seed=8)
    st.step[:, 2] = 0.8          # phase jumps of the order of its [0,1] range: wraps all the time
    lad.step[:] = st.step
    lad.circular = 1 << 2
    s = HipSampler(w.model, 4, 6, w.data, seed=8, circular_params=1 << 2)
    s.set_state(st)
    d = torch.zeros((200, 6, 6), dtype=torch.float64, device="cuda")
    s.run_sampler(20, 10, d.data_ptr())
    s.synchronize()
    dev = s.get_state()
    ref = orc.run_sampler(lad, rng, 20, 10, record=True)
    assert_match(dev, lad, rng, what="circular")
    np.testing.assert_allclose(d.cpu().numpy(), ref, rtol=1e-9)
    # and it really is a different chain from the redraw rule
    lad2 = orc.Ladder(w.model, 6, 4, w.data)
    from tests.helpers import to_oracle
    to_oracle(st, lad2)
    orc.run_sampler(lad2, orc.Rng(orc.RNG_STREAMS, 8, lad2), 20, 10)
    assert not np.array_equal(lad2.params, lad.params)
    s.close()


@pytest.mark.parametrize("waves,helper", [(0, None), (8, None), (0, 0), (8, 0)])
@pytest.mark.parametrize("flags", [0, capi.FLAG_RANDOMSWAP])
def test_config4_shard_kernel_as_the_bench_launches_it_matches_oracle(flags, waves, helper, monkeypatch):
    """BASELINE config 4 as one GPU of eight sees it and as bench.py --config 4 launches it: pulse,
    256 chains x 1024 points, the geometry the engine chooses by itself (waves 0: four likelihood waves
    + owner + three producers + the helper wavefront per chain -- one workgroup per CU --, the data vector in LDS;
    eight likelihood waves, the engine's choice for more than three modes, asked for by name), n_swap 1 -- every
    step is a round, so the pipeline of prepared proposals runs through round boundaries that are none and
    restarts only for the two chains of a swap attempt --, 256 rounds in ONE launch, the pulse prior computed
    one logarithm per lane (by the helper).  helper 0 (APEMOST_OB_HELPER=0): the kernels without the helper, the
    owner takes the prior -- what ladders of 257-512 chains run; their twelve-wave form at 256 chains is placed by
    hipLaunchCooperativeKernel (256 workgroups fit the occupancy figure but not the engine's cautious estimate).
    Against the oracle: counters, ticks and swap counts bit-exact, every recorded row to 1e-9; the same under
    -DRANDOMSWAP."""
    if helper is not None:
        monkeypatch.setenv("APEMOST_OB_HELPER", str(helper))
    torch = _torch()
    n_chain, n_rounds = 256, 256
    w = wl.pulse(n_data=1024, n_chain=n_chain)
    st, lad, rng = make_pair(w, n_chain, seed=404, init_prob=True)
    lad.randomswap = 1 if flags & capi.FLAG_RANDOMSWAP else 0
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=404, flags=flags, waves_per_chain=waves)
    assert s.geometry == (waves or 4, True)
    one_barrier, cooperative, max_rounds = s.launch_policy
    # (with a helper: one workgroup per CU, resident as launched.  Without: 256 eight-wave workgroups are resident
    # by the engine's own estimate, two per CU; the twelve-wave ones go through the runtime's placement)
    assert s.ob_helper == (helper is None)
    assert one_barrier and cooperative == (waves == 8 and helper == 0) and max_rounds >= n_rounds
    s.set_state(st)
    d = torch.zeros((n_rounds, 1, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
    before = s.round[0]
    s.launch_rounds(n_rounds, 1, False, d.data_ptr())          # one launch, as one bench step
    s.launch_round(0, True)                                     # the swap attempt that closes the last round
    s.synchronize()
    assert s.launch_policy[1] == cooperative and s.round[0] - before == n_rounds   # nothing was re-issued
    dev = s.get_state()
    ref = orc.run_sampler(lad, rng, n_rounds, 1, record=True, n_threads=8)
    assert_match(dev, lad, rng, what="config-4 shard kernel flags=%d" % flags)
    np.testing.assert_allclose(d.cpu().numpy().reshape(ref.shape), ref, rtol=1e-9, atol=1e-300)
    assert dev.swapcount.sum() > 20 and np.all(dev.accept + dev.reject == n_rounds)
    s.close()


def test_a_hand_off_that_times_out_is_reported_and_the_sampler_falls_back():
    """The safety net under multi-round launches: a workgroup that waits for its swap partner's record
    gives up after a bounded number of polls (8 million; a few thousand under the test hook), raises
    the launch's error word and goes on, so the grid always drains.  FLAG_TEST_WITHHOLD_PUBLISH makes
    the lower chain of swap attempt 3 keep its publish to itself: synchronize reports code 1, the
    word is cleared, the sampler holds one round per launch from then on, and -- reloaded -- gives
    the oracle's chain."""
    torch = _torch()
    n_chain, n_rounds, n_swap = 32, 40, 3
    w = wl.simplesin(n_data=128, n_chain=n_chain)
    st, lad, rng = make_pair(w, n_chain, seed=91)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=91, flags=capi.FLAG_TEST_WITHHOLD_PUBLISH)
    s.set_state(st)
    assert s.max_rounds_per_launch > 1
    assert s.swap_pair(3) >= 0
    s.launch_rounds(n_rounds, n_swap, False)
    with pytest.raises(capi.ApemostHipError) as err:
        s.synchronize()
    assert "code 1" in str(err.value) and "one round per launch" in str(err.value)
    s.synchronize()                                    # the word is cleared: no second report
    assert s.max_rounds_per_launch == 1 and s.launch_policy == (s.launch_policy[0], False, 1)
    with pytest.raises(capi.ApemostHipError):
        s.launch_rounds(2, n_swap, True)               # refused on the host, nothing launched
    # the results of the void launch are dropped: reload, rewind the swap stream, run again
    s.set_state(st)
    s.set_round(0, False)
    d = torch.zeros((n_rounds, n_swap, n_chain, 6), dtype=torch.float64, device="cuda")
    s.run_sampler(n_rounds, n_swap, d.data_ptr())
    s.synchronize()
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(s.get_state(), lad, rng, what="after a hand-off timeout")
    np.testing.assert_allclose(d.cpu().numpy().reshape(ref.shape), ref, rtol=1e-9)
    s.close()


def test_a_refused_cooperative_launch_in_the_middle_of_a_sharded_run():
    """ShardedLadder.run_sampler asks for the launch limit before every launch (it was read
    once, so after a refusal -- the engine falls to one round per launch -- the next launch of the
    same call asked for more and was rejected).  1100 rounds is more than one launch holds: the first
    launch is refused and re-issued round by round, all later ones are single rounds."""
    torch = _torch()
    from apemost_amd.distributed import HipShardEngine, ShardedLadder
    n_chain, n_rounds, n_swap = 16, 1100, 2
    w = wl.simplesin(n_data=64, n_chain=n_chain)
    st, lad, rng = make_pair(w, n_chain, seed=5150)
    s = HipSampler(w.model, 4, n_chain, w.data, seed=5150,
                   flags=capi.FLAG_COOPERATIVE_LAUNCH | capi.FLAG_TEST_REFUSE_COOPERATIVE)
    s.set_state(st)
    assert s.max_rounds_per_launch == 1024
    ladder = ShardedLadder(HipShardEngine(s, torch), n_chain, 0, n_chain, 0, 1, None)
    ladder.run_sampler(n_rounds, n_swap, None)
    s.synchronize()
    assert s.max_rounds_per_launch == 1
    orc.run_sampler(lad, rng, n_rounds, n_swap, n_threads=8)
    assert_match(s.get_state(), lad, rng, what="refusal mid-run")
    s.close()
