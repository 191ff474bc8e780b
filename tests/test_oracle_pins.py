"""Pins the CPU oracle to every known-answer value available for this path.

Sources of the expected values:
  * GSL's published mt19937 self-test (rng/test.c: seed 4357, 1000th output) and
    the reference's default seed rule (src/mcmc.c:27-35 -> gsl_rng_env_setup, seed 0 => 4357);
  * Random123's published Philox4x32-10 known-answer vectors;
  * the reference manual's eval example (doc/manual.rst:190-213);
  * the reference's own unit-test fixtures (tests/tests.c:89-160, tests/testinput1,
    tests/testlc.dat -- copied as data into tests/golden/);
  * SURVEY.md 8(c) probe values (marked as such: not from real GSL).
"""
import math
import os

import numpy as np
import pytest

from oracle import oracle as orc


def test_mt19937_default_seed_stream():
    s = orc.mt_stream(0, 1000)           # seed 0 => 4357
    assert list(s[:3]) == [4293858116, 699692587, 1213834231]
    assert int(s[999]) == 1186927261     # GSL rng/test.c KAT
    assert np.array_equal(s, orc.mt_stream(4357, 1000))
    # numpy's legacy generator uses the same init_genrand(4357)
    ref = np.random.RandomState(4357).randint(0, 2 ** 32, size=1000, dtype=np.uint64)
    assert np.array_equal(s.astype(np.uint64), ref)


def test_uniform_and_gaussian_kats():
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    import ctypes as C
    L = orc.lib()
    u = [L.orc_uniform(C.byref(rng.c)) for _ in range(3)]
    assert u == [4293858116 / 2 ** 32, 699692587 / 2 ** 32, 1213834231 / 2 ** 32]
    assert abs(u[0] - 0.999741748906672) < 1e-15
    # SURVEY 8(c) probe values (polar method, second variate discarded)
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    g = [L.orc_gaussian(C.byref(rng.c), 1.0) for _ in range(3)]
    np.testing.assert_allclose(g, [0.1339186081186759, -0.088100991831438394, 1.6744084062537739],
                               rtol=1e-14)
    assert rng.c.draws == 10
    nxt = math.log(L.orc_uniform(C.byref(rng.c)))
    assert abs(nxt - (-0.27451079819355362)) < 1e-15


def test_philox_random123_kats():
    assert orc.philox_block([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox_block([0xffffffff] * 4, [0xffffffff] * 2) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox_block([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                            [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # stream addressing: n-th output = word n%4 of block n/4, subsequence in the upper counter
    s = orc.philox_stream(0, 0, 8)
    assert list(s[:4]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(s[4:]) == orc.philox_block([1, 0, 0, 0], [0, 0])
    seed, sub, blk = 0xa4093822 | (0x299f31d0 << 32), 0x13198a2e | (0x03707344 << 32), 0x1243f6a88
    assert list(orc.philox_stream(seed, sub, 4, start=4 * blk)) == \
        orc.philox_block([blk & 0xffffffff, blk >> 32, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])


def test_tick_addressed_attempts_follow_the_polar_method():
    """STREAMS mode: attempt q at tick t is Philox block (t<<24)|q; same transform as gsl_ran_gaussian"""
    import math
    seed, chain, slot, tick = 9, 3, 2, 77
    n_ok = 0
    for q in range(200):
        ok, y, s = orc.gaussian_attempt(seed, chain, slot, tick, q)
        w = orc.philox_stream(seed, chain * 256 + slot, 2, start=4 * ((tick << 24) | q))
        x, yy = -1 + 2 * (int(w[0]) / 2 ** 32), -1 + 2 * (int(w[1]) / 2 ** 32)
        r2 = x * x + yy * yy
        assert ok == (w[0] != 0 and w[1] != 0 and 0 < r2 <= 1.0)
        if ok:
            n_ok += 1
            assert y == yy and abs(s - math.sqrt(-2.0 * math.log(r2) / r2)) <= 1e-15 * s
    assert 130 < n_ok < 185
    w0 = orc.philox_stream(seed, chain * 256 + 4, 1, start=4 * (tick << 24))[0]
    assert orc.accept_log_uniform(seed, chain, 4, tick) == math.log(int(w0) / 2 ** 32)


def test_simplesin_manual_eval_kat():
    # doc/manual.rst:190-213: data rows and "1 0.2 1 0" -> -1.480898044165363e+01, prior 0
    data = np.array([[101, 0.67], [102, 1.01], [103, 7.9e-1], [104, 1.34]])
    prob, prior = orc.loglike(orc.MODEL_SIMPLESIN, [1, 0.2, 1, 0], data, beta=1.0)
    assert prior == 0.0
    assert abs(prob - (-1.480898044165363e+01)) < 5e-14


def _load_params(path):
    rows = []
    for line in open(path):
        t = line.split()
        rows.append((float(t[0]), float(t[1]), float(t[2]), t[3], float(t[4])))
    return rows


def test_reference_parser_fixtures(golden_dir):
    # values asserted by tests/tests.c:95-117
    p = _load_params(os.path.join(golden_dir, "testinput1"))
    assert len(p) == 3
    assert p[0] == (0.7, 0.4, 3.0, "Amplitude", 0.3)
    assert p[1] == (5.2, 4.0, 24.0, "Frequenz", 0.01)
    assert p[2] == (5.4, 0.0, 6.1, "Phase", 0.01)
    d = np.loadtxt(os.path.join(golden_dir, "testlc.dat"))
    assert d.shape == (1522, 2)
    assert d[0, 0] == 1.7355217099999998 and d[1, 0] == 1.7356002600000000
    assert d[1521, 0] == 46.8043750000000003
    assert d[0, 1] == 0.4731745866773314 and d[1, 1] == -0.9900871130450772
    assert d[1521, 1] == -0.3527955490681067


def test_simplesin_on_reference_lightcurve(golden_dir):
    # SURVEY 8(c) G1 (probe, libm sin, reproduced there independently in numpy)
    d = np.loadtxt(os.path.join(golden_dir, "testlc.dat"))
    for params, want in (((0.7, 5.2, 5.4, 0.0), -3.187159885063915e+03),
                         ((1.0, 10.0, 0.25, 0.1), -3.271718790429034e+03)):
        prob, _ = orc.loglike(orc.MODEL_SIMPLESIN, params, d)
        assert abs(prob - want) < 1e-12 * abs(want)
        # independent numpy restatement with serial summation order
        a, f, ph, o = params
        m = a * np.sin(2.0 * np.pi * (f * d[:, 0] + ph)) + o
        ss = 0.0
        for v in (m - d[:, 1]):
            ss += v * v
        assert abs(prob - ss / -0.5) < 1e-12 * abs(want)


def test_log_and_mod_kats():
    # tests/tests.c:145-158 (relative 1e-3 there)
    assert abs(math.log(1e-200) - (-460.5170)) < 1e-3 * 460
    L = orc.lib()
    for x, y, want in ((3.14, 3.00, 0.14), (3.14, 1.30, 0.54), (-3.14, 1.30, 0.76), (0, 1.30, 0.0),
                       (6000.3214, 1.1324, 0.8662), (-6000.3214, 1.1324, 0.2662)):
        assert abs(L.orc_mod_double(x, y) - want) < 1e-3


def test_pulse_models_against_numpy():
    from apemost_amd import workloads as wl
    w = wl.pulse(n_data=64, n_chain=2)
    p = w.start.copy()
    p[3] = 3.3
    beta = 0.37
    prob, prior = orc.loglike(orc.MODEL_PULSE, p, w.data, beta=beta)
    nu, d = w.data[:, 0], w.data[:, 1]
    y = sum(p[j + 1] / (1 + (2 * np.pi * (p[j] - nu) * p[0]) ** 2) for j in (2, 4))
    want_prior = -(np.log(p[3] + 1e-6) + np.log(p[5] + 1e-6)) / 2
    want = want_prior - beta * (p[1] + np.sum(np.log(y) + d / y))
    assert abs(prior - want_prior) < 1e-15
    assert abs(prob - want) < 1e-12 * abs(want)

    w = wl.pulse_vrot(n_data=64, n_chain=2)
    p = w.start.copy()
    prob, prior = orc.loglike(orc.MODEL_PULSE_VROT, p, w.data, beta=beta)
    nu, d = w.data[:, 0], w.data[:, 1]
    lor = lambda f, h: h / (1 + (2 * np.pi * (f - nu) * p[0]) ** 2)
    y = lor(p[3], p[4]) + lor(p[5] - p[2], p[6]) + lor(p[5], p[6]) + lor(p[5] + p[2], p[6])
    want_prior = -(np.log(p[4] + 1e-6) + np.log(p[6] + 1e-6)) / 2
    want = want_prior - beta * (p[1] + np.sum(np.log(y) + d / y))
    assert abs(prob - want) < 1e-12 * abs(want)


def test_beta_ladder_properties():
    n = 8
    b = [orc.get_chain_beta(orc.LADDER_CHEBYSHEV_BETA, i, n, 0.01) for i in range(n)]
    assert b[0] == 1.0 and abs(b[-1] - 0.01) < 1e-15
    assert all(b[i] > b[i + 1] for i in range(n - 1))
    want = [0.01 + 0.99 / 2 * (1 - math.cos((n - 1 - i) * math.pi / (n - 1))) for i in range(n)]
    np.testing.assert_allclose(b, want, rtol=1e-15)
    assert orc.get_chain_beta(orc.LADDER_CHEBYSHEV_BETA, 0, 1, 0.3) == 1.0
    for kind in range(6):
        assert abs(orc.get_chain_beta(kind, 0, n, 0.05) - 1.0) < 1e-15
        assert abs(orc.get_chain_beta(kind, n - 1, n, 0.05) - 0.05) < 1e-15


def test_circular_params_reference_mode():
    """GLOBAL_MT mode with a circular parameter: the first jump is kept and wrapped with mod_double
    (src/markov_chain.c:241-265, src/mcmc_internal.h:46-48)"""
    from apemost_amd import workloads as wl
    w = wl.simplesin(n_data=32, n_chain=1)
    lad = orc.Ladder.from_params(w.model, 1, w.start, w.pmin, w.pmax, w.step, w.data)
    lad.step[0, 2] = 3.0            # phase step of three ranges: almost always outside [0,1]
    lad.circular = 1 << 2
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    before = int(rng.c.draws)
    seen = []
    for _ in range(50):
        orc.step(lad, rng, 0)
        seen.append(lad.params[0, 2])
    assert all(0.0 <= v <= 1.0 for v in seen)
    lad2 = orc.Ladder.from_params(w.model, 1, w.start, w.pmin, w.pmax, w.step, w.data)
    lad2.step[0, 2] = 3.0
    rng2 = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    for _ in range(50):
        orc.step(lad2, rng2, 0)
    # the redraw rule consumes many more uniforms for the same number of steps
    assert int(rng2.c.draws) > int(rng.c.draws) - before + 50


# ---- compile-time variants of the reference (proposal law, swap schedule, -DADAPT) ----------------
# GSL is not vendored in the reference; the laws are pinned to its published algorithms
# (randist/logistic.c, randist/flat.c) restated here independently on the raw generator words,
# and to the distributions they claim to sample.

def test_proposal_laws_reference_mode():
    import ctypes as C
    L = orc.lib()
    words = [int(v) for v in orc.mt_stream(0, 4000)]
    sigma = 0.37
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    got = [L.orc_jump(C.byref(rng.c), sigma, orc.PROPOSAL_LOGISTIC) for _ in range(2000)]
    assert int(rng.c.draws) == 2000                      # no zero word among them: one uniform each
    for k in (0, 1, 2, 999, 1999):
        x = words[k] / 2.0 ** 32
        assert got[k] == sigma * math.log(x / (1 - x))   # a * log(x / (1 - x)), x = gsl_rng_uniform_pos
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    flat = [L.orc_jump(C.byref(rng.c), sigma, orc.PROPOSAL_UNIFORM) for _ in range(2000)]
    for k in (0, 1, 2, 999, 1999):
        u = words[k] / 2.0 ** 32
        assert flat[k] == (-sigma) * (1 - u) + sigma * u  # a * (1 - u) + b * u with a = -sigma, b = sigma
    assert all(-sigma <= v <= sigma for v in flat)
    # the laws they claim to be (scipy's parametrisation: logistic scale = a; uniform on [-a, a])
    from scipy import stats
    assert stats.kstest(got, stats.logistic(scale=sigma).cdf).pvalue > 1e-3
    assert stats.kstest(flat, stats.uniform(-sigma, 2 * sigma).cdf).pvalue > 1e-3
    # and the default is untouched
    rng, rng2 = orc.Rng(orc.RNG_GLOBAL_MT, 0), orc.Rng(orc.RNG_GLOBAL_MT, 0)
    assert L.orc_jump(C.byref(rng.c), sigma, orc.PROPOSAL_GAUSSIAN) == L.orc_gaussian(C.byref(rng2.c), sigma)


def test_proposal_laws_on_tick_addressed_streams():
    """attempt q of (chain, slot) at tick t reads word 0 of block (t<<24)|q; the logistic law fails an
    attempt whose word is zero (GSL redraws), the flat law never fails"""
    seed, chain, slot, tick, sigma = 11, 5, 2, 7, 0.2
    for q in range(5):
        w0 = int(orc.philox_stream(seed, chain * 256 + slot, 1, start=4 * ((tick << 24) | q))[0])
        x = w0 / 2.0 ** 32
        ok, j = orc.jump_attempt(seed, chain, slot, tick, q, orc.PROPOSAL_LOGISTIC, sigma)
        assert ok and j == sigma * math.log(x / (1 - x))
        ok, j = orc.jump_attempt(seed, chain, slot, tick, q, orc.PROPOSAL_UNIFORM, sigma)
        assert ok and j == (-sigma) * (1 - x) + sigma * x
        okg, y, sq = orc.gaussian_attempt(seed, chain, slot, tick, q)
        ok, j = orc.jump_attempt(seed, chain, slot, tick, q, orc.PROPOSAL_GAUSSIAN, sigma)
        assert ok == okg and (not ok or j == sigma * y * sq)


def test_proposal_law_changes_the_chain_and_stays_in_the_box():
    from apemost_amd import workloads as wl
    w = wl.simplesin(n_data=32, n_chain=1)
    ends = {}
    for law in (orc.PROPOSAL_GAUSSIAN, orc.PROPOSAL_LOGISTIC, orc.PROPOSAL_UNIFORM):
        for kind in (orc.RNG_GLOBAL_MT, orc.RNG_STREAMS):
            lad = orc.Ladder.from_params(w.model, 1, w.start, w.pmin, w.pmax, w.step * 0.05, w.data)
            lad.proposal = law
            rng = orc.Rng(kind, 3, lad)
            orc.calc_model(lad, 0)
            for _ in range(300):
                orc.step(lad, rng, 0)
                assert ((lad.params[0] >= w.pmin) & (lad.params[0] <= w.pmax)).all()
            assert 0 < lad.accept[0] < 300
            ends[law, kind] = lad.params[0].copy()
    assert len({tuple(v) for v in ends.values()}) == 6


def test_randomswap_draws_one_more_uniform():
    """parallel_tempering_decide_swap_random(chains, n_beta, 1): swap_probability first (always below
    1.0 / 1), then the pair from the NEXT uniform, then the acceptance uniform"""
    from apemost_amd import workloads as wl
    n = 6
    w = wl.simplesin(n_data=32, n_chain=n)
    words = [int(v) for v in orc.mt_stream(0, 30)]
    lad = orc.Ladder.from_params(w.model, n, w.start, w.pmin, w.pmax, w.step, w.data)
    lad.beta[:] = np.linspace(1.0, 0.1, n)
    lad.prob[:] = -np.arange(n) * 3.0
    lad.randomswap = 1
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    for r in range(5):
        a, trace = orc.tempering_interaction(lad, rng)
        assert int(rng.c.draws) == 3 * (r + 1)
        assert int(trace[0]) == int(n * 1000 * (words[3 * r + 1] / 2.0 ** 32)) % (n - 1)
        assert trace[2] == math.log(words[3 * r + 2] / 2.0 ** 32)
    # tick-addressed: words 1 and 2 of the round's block instead of 0 and 1
    lad.randomswap = 1
    rng = orc.Rng(orc.RNG_STREAMS, 9, lad)
    pw = [int(v) for v in orc.philox_stream(9, 1 << 63, 8)]
    a, trace = orc.tempering_interaction(lad, rng)
    assert int(trace[0]) == int(n * 1000 * (pw[1] / 2.0 ** 32)) % (n - 1) and trace[2] == math.log(pw[2] / 2.0 ** 32)
    lad.randomswap = 0
    a, trace = orc.tempering_interaction(lad, rng)
    assert int(trace[0]) == int(n * 1000 * (pw[4] / 2.0 ** 32)) % (n - 1) and trace[2] == math.log(pw[5] / 2.0 ** 32)


def test_adapt_thresholds_and_scaling():
    """adapt() with -DADAPT (src/parallel_tempering.c:282-301): nothing below 20000 counted updates;
    accepts/REJECTS below target-0.05 -> steps * 0.99, above target+0.05 -> steps * (1/0.99); past
    100000 the counters restart"""
    import ctypes as C
    from apemost_amd import workloads as wl
    w = wl.simplesin(n_data=16, n_chain=4)
    lad = orc.Ladder.from_params(w.model, 4, w.start, w.pmin, w.pmax, w.step, w.data)
    lad.adapt, lad.adapt_target = 1, 0.5
    step0 = lad.step.copy()
    #            accepts per parameter, rejects per parameter
    cases = [(2000, 2999),      # 19996 < 20000: untouched
             (1500, 3500),      # ratio 0.4286 < 0.45: scaled down
             (2000, 3100),      # ratio 0.645 > 0.55: scaled up
             (8500, 16600)]     # 100400 > 100000, ratio 0.512 in the band: reset only
    for c, (a, r) in enumerate(cases):
        lad.params_accepts[c, :], lad.params_rejects[c, :] = a, r
        lad.accept[c], lad.reject[c] = a, r
    st = lad.c_state()
    for c in range(4):
        orc.lib().orc_adapt(C.byref(st), c)
    assert np.array_equal(lad.step[0], step0[0]) and lad.accept[0] == 2000
    assert np.array_equal(lad.step[1], step0[1] * 0.99)
    assert np.array_equal(lad.step[2], step0[2] * (1 / 0.99))
    assert np.array_equal(lad.step[3], step0[3])
    assert lad.params_accepts[3].sum() == 0 and lad.params_rejects[3].sum() == 0 and lad.accept[3] == lad.reject[3] == 0
    assert lad.params_accepts[1].sum() == 6000      # below 100000: kept


def test_rwm_block_against_a_numpy_restatement():
    """adapt() with -DRWM (src/parallel_tempering.c:268-281) + rmw_adapt_stepwidth (src/markov_chain.c:342-367):
    prob_old kept, ONE markov_chain_step (no mcmc_check behind it), then per parameter
    step += U / sqrt(n_iter) * (min(1, exp(prob - prob_old)) - TARGET) * (max - min), clamped to [1e-7, 1e6] x
    range.  The oracle's orc_rwm against the same expression in numpy on the oracle's own step, in both RNG modes
    (GLOBAL_MT: the n_par uniforms are the next n_par of the one stream, as get_next_uniform_random draws them;
    STREAMS: words p % 4 of blocks (tick << 24) | (1 + p / 4) of the accept slot), the clamps included."""
    from apemost_amd import workloads as wl
    from tests.helpers import make_pair
    w = wl.simplesin(n_data=32, n_chain=3)
    for kind in (orc.RNG_STREAMS, orc.RNG_GLOBAL_MT):
        st, lad, _ = make_pair(w, 3, seed=9, init_prob=True)
        _, twin, _ = make_pair(w, 3, seed=9, init_prob=True)
        rng, rng2 = orc.Rng(kind, 9, lad), orc.Rng(kind, 9, twin)
        lad.adapt_target = twin.adapt_target = 0.4
        lad.n_iter[:] = twin.n_iter[:] = (7, 1000, 123456)
        lad.step[2, 1] = twin.step[2, 1] = 1.2e-7 * (w.pmax[1] - w.pmin[1])       # next to the lower clamp
        lad.step[1, 0] = twin.step[1, 0] = 1e6 * (w.pmax[0] - w.pmin[0])          # on the upper one
        for c in range(3):
            prob_old, tick = twin.prob[c], (int(rng2.ticks[c]) if kind == orc.RNG_STREAMS else 0)
            orc.rwm(lad, rng, c)
            orc.step(twin, rng2, c)
            alpha = min(1.0, float(np.exp(twin.prob[c] - prob_old)))
            for p in range(4):
                u = (orc.lib().orc_rwm_uniform(9, c, 4, tick, p) if kind == orc.RNG_STREAMS
                     else orc.lib().orc_uniform(ctypes_byref(rng2)))
                scale = w.pmax[p] - w.pmin[p]
                v = twin.step[c, p] + u / np.sqrt(float(twin.n_iter[c])) * (alpha - 0.4) * scale
                twin.step[c, p] = min(max(v, 0.0000001 * scale), 1000000 * scale)
            assert np.array_equal(lad.step[c], twin.step[c]), (kind, c)
            assert lad.prob[c] == twin.prob[c] and np.array_equal(lad.params[c], twin.params[c])
            assert lad.n_iter[c] == twin.n_iter[c] and lad.accept[c] + lad.reject[c] == 1
        assert lad.step[2, 1] >= 1e-7 * (w.pmax[1] - w.pmin[1]) and lad.step[1, 0] <= 1e6 * (w.pmax[0] - w.pmin[0])
        if kind == orc.RNG_STREAMS:
            assert np.array_equal(rng.ticks, rng2.ticks)
            # the four uniforms of a block are its four words, the fifth parameter's is word 0 of the next block
            words = orc.philox_stream(9, 2 * 256 + 4, 8, start=4 * ((5 << 24) | 1))
            assert [orc.lib().orc_rwm_uniform(9, 2, 4, 5, p) for p in range(4)] == [x / 2.0 ** 32 for x in words[:4]]


def ctypes_byref(rng):
    import ctypes as C
    return C.byref(rng.c)


# ---- the reference's other example likelihoods (checkers of the user-supplied device models) -------------
def test_oracle_restatements_of_the_other_example_apps_against_numpy():
    """ll_sine2 / ll_normal / ll_bernoulli of the oracle (apps/simplesin2.c:12-34, apps/normal.c:8-34,
    apps/bernoulli_example.c:10-49) against the same formulas written independently in numpy, with the
    serial left-to-right sums of the C loops"""
    rs = np.random.RandomState(11)
    # simplesin2: beta * sum (A sin(2 pi (f x + 0.3312)) - y)^2 / (-2 sigma^2)
    d = np.stack([100 + 0.5 * np.arange(40), rs.normal(0, 1, 40)], 1)
    for A, f, beta in ((0.8, 0.21, 1.0), (1.7, 0.05, 0.3)):
        want = 0.0
        for x, y in d:
            r = A * np.sin(2.0 * np.pi * (f * x + 0.3312)) - y
            want += r * r
        want = beta * want / (-2 * 0.5 * 0.5)
        got, prior = orc.loglike(orc.MODEL_SINE2, np.array([A, f]), d, beta=beta)
        assert abs(got - want) <= 1e-13 * abs(want) and prior == 0.0
    # normal: beta * max over ten peaks (even: Gaussian-like, odd: triangular), floor 0
    for x, beta in ((3.0, 1.0), (150.0, 0.5), (8000.0, 0.2), (0.4, 1.0)):
        b = 0.0
        for i in range(10):
            pos, height, sigma = np.exp(i), 10 * 1.0 ** i, float(i)
            with np.errstate(divide="ignore", invalid="ignore"):
                if i % 2 == 0:
                    a = -sigma * ((x - pos) / sigma) ** 2 / 2 + height if sigma else np.nan
                else:
                    a = -height * abs(x - pos) / sigma + height
            if a > b:
                b = a
        got, _ = orc.loglike(orc.MODEL_NORMAL, np.array([x]), d, beta=beta)
        assert abs(got - beta * b) <= 1e-13 * max(abs(beta * b), 1.0), (x, got, beta * b)
    # bernoulli: prior = sum_j>=1 -(p_j / 2)^2 / 2; prob = prior + beta * sum_i log(p_i or 1 - p_i)
    X = rs.normal(0, 1, (30, 2))
    out = (rs.uniform(size=30) < 0.5).astype(float)
    data = np.column_stack([out, X])
    for p, beta in ((np.array([0.3, 1.1, -0.7]), 1.0), (np.array([-1.0, 0.2, 2.5]), 0.4)):
        prior = sum(-(pj / 2) ** 2 / 2 for pj in p[1:])
        s = 0.0
        for row in data:
            eta = p[0] + row[1] * p[1] + row[2] * p[2]
            pi = 1 / (1 + np.exp(-eta)) if eta > 0 else np.exp(eta) / (1 + np.exp(eta))
            s += np.log(1 - pi) if row[0] == 0 else np.log(pi)
        got, got_prior = orc.loglike(orc.MODEL_BERNOULLI, p, data, beta=beta)
        assert abs(got_prior - prior) <= 1e-15 and abs(got - (prior + beta * s)) <= 1e-13 * abs(prior + beta * s)
