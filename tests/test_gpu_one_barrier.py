"""The one-barrier round kernel (apemost_amd/csrc/pt_onebarrier.h: 4 or 8 likelihood waves + owner
+ candidate producer, accept test as a threshold on the data sum, both next proposals prepared
ahead) against the classic two-phase kernel and against the oracle.  The two kernels make the same
draws and add the data sum in the same order, so their chains are bit-identical (the accept
comparison differs in form only: a different decision needs prob_new - prob and ln U to agree to an
ulp)."""
import numpy as np
import pytest

from apemost_amd import capi, workloads as wl
from apemost_amd.sampler import HipSampler
from oracle import oracle as orc
from tests.helpers import assert_match, make_pair, small_workloads, to_oracle

pytestmark = pytest.mark.gpu
FIELDS = ("params", "params_best", "prob", "prob_best", "prior", "accept", "reject", "swapcount", "ticks", "n_iter",
          "params_accepts", "params_rejects")


def _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=0, pieces=None, **kw):
    import torch
    s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=seed, waves_per_chain=waves, flags=flags, **kw)
    s.set_state(st)
    d = torch.zeros((n_rounds * n_swap, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
    done = 0
    for k in (pieces or (n_rounds,)):
        s.run_sampler(k, n_swap, d[done * n_swap:].data_ptr())
        done += k
    s.synchronize()
    out = s.get_state(), d.cpu().numpy()
    s.close()
    return out


@pytest.mark.parametrize("name", ["simplesin", "sine3", "pulse", "pulse_vrot"])
@pytest.mark.parametrize("waves", [4, 8])
def test_one_barrier_equals_two_phase_kernel_and_oracle(name, waves):
    w = small_workloads()[name]
    n_chain, n_rounds, n_swap, seed = 8, 60, 11, 97
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    a, sa = _run(w, st, n_chain, n_rounds, n_swap, waves, seed)
    b, sb = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=capi.FLAG_TWO_BARRIER_STEP)
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(sa, sb)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(a, lad, rng, what="one barrier " + name)
    np.testing.assert_allclose(sa, ref, rtol=1e-9, atol=1e-300)
    assert a.swapcount.sum() > 0 and 0 < a.accept.sum() < a.n_iter.sum()


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("n_swap", [1, 2, 15])
def test_one_barrier_round_shapes(n_swap, waves):
    """rounds of one step (a barrier at the round start, one per step), of two, of many; launches cut
    at arbitrary rounds; single-round launches (every swap fused into the next launch's start)"""
    w = wl.simplesin(n_data=1024, n_chain=16)        # one pass of the data per lane: rows in registers
    n_chain, n_rounds, seed = 16, 90, 5
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    a, sa = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, lds_policy=1)
    b, sb = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, pieces=(1, 7, 40, 42))
    c, sc = _run(w, st, n_chain, n_rounds, n_swap, waves, seed, flags=capi.FLAG_SINGLE_ROUND_LAUNCHES)
    for other, so in ((b, sb), (c, sc)):
        for f in FIELDS:
            assert np.array_equal(getattr(a, f), getattr(other, f)), f
        assert np.array_equal(sa, so)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True, n_threads=8)
    assert_match(a, lad, rng, what="round shape %d waves %d" % (n_swap, waves))
    np.testing.assert_allclose(sa, ref, rtol=1e-9, atol=1e-300)


def test_one_barrier_redraw_path_and_circular_parameters():
    """step widths of six times the prior box: almost every prepared attempt leaves the box, so the
    prepared proposals fail and the workgroup takes the redraw path (extra barrier) constantly; with
    the phase circular (-DCIRCULAR_PARAMS) its first usable attempt wraps instead"""
    w = small_workloads()["simplesin"]
    for circular in (0, 1 << 2):
        st, lad, rng = make_pair(w, 4, seed=3)
        st.step[:] = (w.pmax - w.pmin) * 6.0
        lad.step[:] = st.step
        lad.circular = circular
        for waves in (4, 8):
            dev, samples = _run(w, st, 4, 12, 5, waves, 3, circular_params=circular)
            lad2 = orc.Ladder(w.model, 4, 4, w.data)
            to_oracle(st, lad2)
            lad2.circular = circular
            rng2 = orc.Rng(orc.RNG_STREAMS, 3, lad2)
            ref = orc.run_sampler(lad2, rng2, 12, 5, record=True)
            assert_match(dev, lad2, rng2, what="wide steps waves=%d circular=%d" % (waves, circular))
            np.testing.assert_allclose(samples, ref, rtol=1e-9)


def test_one_barrier_maximum_parameter_count():
    """n_par = 62: one attempt lane per parameter in the owner and the producer; pulse with 30 modes
    reads its parameters from the chosen LDS row while it evaluates"""
    rs = np.random.RandomState(9)
    n_modes = 30
    nu = np.linspace(10, 12, 300)
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
    st, lad, rng = make_pair(w, 3, seed=42)
    for waves in (4, 8):
        st, lad, rng = make_pair(w, 3, seed=42)
        dev, samples = _run(w, st, 3, 6, 4, waves, 42)
        ref = orc.run_sampler(lad, rng, 6, 4, record=True)
        assert_match(dev, lad, rng, what="62 parameters waves %d" % waves)
        np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)


def test_one_barrier_config2_bench_shape_matches_oracle():
    """BASELINE config 2 as bench.py runs it: 128 chains x 1024 points, the engine's own choice of
    workgroup (4 likelihood waves, rows in registers), 32 rounds x 15 steps per launch with the swaps
    handed over inside the launch"""
    w = wl.simplesin(n_data=1024, n_chain=128)
    st, lad, rng = make_pair(w, 128, seed=7)
    probe = HipSampler(w.model, w.n_par, 128, w.data)
    assert probe.geometry == (4, True)
    probe.close()
    dev, samples = _run(w, st, 128, 64, 15, 0, 7)
    ref = orc.run_sampler(lad, rng, 64, 15, record=True, n_threads=8)
    assert_match(dev, lad, rng, what="config 2")
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    assert dev.swapcount.sum() > 5


@pytest.mark.parametrize("name", ["simplesin", "pulse"])
def test_injected_ties_of_the_accept_comparison(name, capsys):
    """check_accept (src/markov_chain.c:282-311) accepts iff prob_new == prob or prob_new > prob or
    ln U < prob_new - prob; the one-barrier kernel decides the same question as ONE comparison on the raw
    data sum, S < S_max(prob, ln U, prior) (pt_onebarrier.h ObThreshold).  Here the tie is injected: every
    chain's `prob` is set to prob_new - ln U of its coming step (the kernel's own prob_new of that proposal,
    taken from a first run in which everything is accepted; ln U of the tick from the oracle's stream) moved
    by k = -12..12 units u, one step is run through apemost_hip_launch_round, and the decisions are compared
    with the reference's rule evaluated on the same numbers.  u = one ulp of the largest number the
    log-posterior is put together from -- max(|prob_new|, |prior|, |beta (p1 + S)|, |ln U|) 2^-52: that, not
    the ulp of a `prob` that may be a small difference of large terms, is the rounding the reference's own
    prob_new carries.
      * the two-phase kernel uses the reference's comparison itself: identical for every k;
      * the one-barrier kernel: identical outside a window of +-kWindow u around the tie (S_max is three
        roundings away from the reference's difference), monotone in k -- never an accept above a reject --,
        and inside the window it is the threshold form that decides."""
    import torch
    kWindow, kRange = 4, 12
    w = small_workloads()[name]
    n_chain, seed = 8, 61
    st, _, _ = make_pair(w, n_chain, seed=seed)
    st.prob[:] = -1e10                           # quirk Q2: the first proposal is accepted whatever it is

    def one_step(state, flags):
        s = HipSampler(w.model, w.n_par, n_chain, w.data, seed=seed, waves_per_chain=4, flags=flags)
        s.set_state(state)
        d = torch.zeros((1, n_chain, w.n_par + 2), dtype=torch.float64, device="cuda")
        s.launch_round(1, False, d.data_ptr())
        s.synchronize()
        out = s.get_state(), d.cpu().numpy()[0]
        s.close()
        return out

    first, rows = one_step(st, 0)
    assert np.all(first.accept == 1)
    prob_new = rows[:, w.n_par].copy()           # the kernel's own log-posterior of each chain's proposal
    prior = prob_new - rows[:, w.n_par + 1]
    ln_u = np.array([orc.accept_log_uniform(seed, c, w.n_par, 0) for c in range(n_chain)])
    assert np.all(ln_u < 0)
    tie = prob_new - ln_u                        # prob_new - tie ~ ln U
    unit = 2.0 ** -52 * np.maximum.reduce([np.abs(prob_new), np.abs(prior), np.abs(prior - prob_new), np.abs(ln_u), np.abs(tie)])
    ks = list(range(-kRange, kRange + 1))
    table = {}
    for k in ks:
        inj = st.copy()
        p = tie + k * unit
        inj.prob[:] = p
        ref = (prob_new == p) | (prob_new > p) | (ln_u < prob_new - p)
        ob, rows_ob = one_step(inj, 0)
        tp, rows_tp = one_step(inj, capi.FLAG_TWO_BARRIER_STEP)
        assert np.array_equal(rows_ob[:, :w.n_par][ob.accept == 1], rows[:, :w.n_par][ob.accept == 1])
        assert np.array_equal(tp.accept == 1, ref), (k, "two-phase kernel")
        table[k] = (ref, ob.accept == 1)
    flips = []
    for c in range(n_chain):
        ref_c = np.array([table[k][0][c] for k in ks])
        ob_c = np.array([table[k][1][c] for k in ks])
        flips.append(int(ob_c.sum()) - int(ref_c.sum()))   # units by which the threshold form's tie sits above the reference's
    with capsys.disabled():
        print("\n[%s] units u by which the one-barrier tie sits above the reference's, per chain: %s (u / ulp(prob): %s)"
              % (name, flips, np.round(unit / np.spacing(np.abs(tie)), 1).tolist()))
    for c in range(n_chain):
        ref_c = np.array([table[k][0][c] for k in ks])
        ob_c = np.array([table[k][1][c] for k in ks])
        # a larger `prob` is harder to beat: accepts (small k) first, then rejects
        assert np.all(np.diff(ob_c.astype(int)) <= 0) and np.all(np.diff(ref_c.astype(int)) <= 0), c
        assert ob_c[0] and not ob_c[-1], c
        differ = np.array(ks)[ref_c != ob_c]
        assert np.all(np.abs(differ) <= kWindow), (c, differ)


@pytest.mark.parametrize("name,sf,sh", [("pulse", 1e30, 1e200), ("pulse_vrot", 1e30, 1e-200), ("pulse", 1.0, 1e-250)])
def test_one_barrier_sampling_through_the_range_guard_matches_oracle(name, sf, sh):
    """The pulse likelihoods' pair-range guard inside the one-barrier step: a spectrum in units where the products of
    two points' numerators or denominators leave fp64's range (frequencies x sf with the lifetime / sf, heights and data x
    sh), so that every evaluation of every likelihood wave falls back to the reference's operation order
    (ObEngine::lik_partial -> Model::term_ref) -- sampled with swaps against the oracle: counters and ticks exact, rows
    1e-9, and bit-identical to the two-phase kernel (whose Engine::lane_sum takes the same fallback)."""
    w = small_workloads()[name]
    heights = [3, 5] if name == "pulse" else [4, 6]
    freqs = [2, 4] if name == "pulse" else [2, 3, 5]
    w.data = w.data.copy()
    w.data[:, 0] *= sf
    w.data[:, 1] *= sh
    for a in (w.start, w.pmin, w.pmax, w.step):
        a[0] /= sf
        a[freqs] *= sf
        a[heights] *= sh
    n_chain, n_rounds, n_swap, seed = 6, 25, 7, 83
    st, lad, rng = make_pair(w, n_chain, seed=seed)
    a, sa = _run(w, st, n_chain, n_rounds, n_swap, 4, seed)
    b, sb = _run(w, st, n_chain, n_rounds, n_swap, 4, seed, flags=capi.FLAG_TWO_BARRIER_STEP)
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert np.array_equal(sa, sb)
    ref = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(a, lad, rng, what="range guard " + name)
    np.testing.assert_allclose(sa, ref, rtol=1e-9, atol=0)
    assert np.all(np.isfinite(sa)) and 0 < a.accept.sum() < a.n_iter.sum()
