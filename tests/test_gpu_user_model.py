"""A likelihood that is not one of the built-in device models, supplied as device source and compiled
by hiprtc into the engine's two-phase kernels (1, 2, 4 and 8 waves per chain since round 4) (include/apemost_device_model.h, APEMOST_MODEL_USER;
SURVEY 8(f4): the plugin surface for apps/simplesin2.c, apps/normal.c, apps/bernoulli_example.c and
for functions registered with set_function).  Checked against oracle models that restate those apps
(oracle/apemost_oracle.c ll_sine2 / ll_normal / ll_bernoulli)."""
import os
import subprocess

import numpy as np
import pytest

from apemost_amd import capi, workloads as wl
from apemost_amd.sampler import HipSampler
from apemost_amd.state import LadderState
from oracle import oracle as orc
from tests import hostlib
from tests.helpers import assert_match, to_oracle

pytestmark = pytest.mark.gpu

MODELS = os.path.join(hostlib.HOST, "examples", "device_models")


def _sine2(n_data=300):
    rs = np.random.RandomState(4)
    x = 100 + 0.5 * np.arange(n_data)
    y = 0.8 * np.sin(2 * np.pi * (0.21 * x + 0.3312)) + rs.normal(0, 0.5, n_data)
    box = dict(start=np.array([0.9, 0.2]), pmin=np.array([0.0, 0.0]), pmax=np.array([2.0, 0.3]))
    return np.stack([x, y], 1), box


def _bernoulli(n_data=257):
    rs = np.random.RandomState(5)
    X = rs.normal(0, 1, (n_data, 2))
    eta = 0.3 + 1.1 * X[:, 0] - 0.7 * X[:, 1]
    out = (rs.uniform(size=n_data) < 1 / (1 + np.exp(-eta))).astype(float)
    box = dict(start=np.array([0.0, 0.0, 0.0]), pmin=np.array([-5.0] * 3), pmax=np.array([5.0] * 3))
    return np.column_stack([out, X]), box


def _normal():
    box = dict(start=np.array([3.0]), pmin=np.array([0.0]), pmax=np.array([9000.0]))
    return np.zeros((4, 2)), box


CASES = {"simplesin2": (_sine2, orc.MODEL_SINE2), "bernoulli_example": (_bernoulli, orc.MODEL_BERNOULLI),
         "normal": (_normal, orc.MODEL_NORMAL)}


def _pair(name, n_chain, seed, waves=0):
    make, omodel = CASES[name]
    data, box = make()
    n_par = len(box["start"])
    st = LadderState.from_params(n_chain, box["start"], box["pmin"], box["pmax"], (box["pmax"] - box["pmin"]) * 0.03)
    for i in range(n_chain):
        st.beta[i] = orc.get_chain_beta(orc.LADDER_CHEBYSHEV_BETA, i, n_chain, 0.05) if n_chain > 1 else 1.0
        st.step[i] = np.minimum(st.step[i] * st.beta[i] ** -0.5, box["pmax"] - box["pmin"])
    lad = orc.Ladder(omodel, n_chain, n_par, data)
    to_oracle(st, lad)
    for c in range(n_chain):
        orc.calc_model(lad, c)
    st.prob[:], st.prior[:] = lad.prob, lad.prior
    s = HipSampler(wl.MODEL_USER, n_par, n_chain, data, seed=seed, waves_per_chain=waves,
                   device_model_source=os.path.join(MODELS, name + ".hip"))
    s.set_state(st)
    return s, st, lad, orc.Rng(orc.RNG_STREAMS, seed, lad), data, box


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_user_model_likelihood_sampling_and_calibration_match_oracle(name, waves):
    """the device source of each of the reference's three other example apps, in workgroups of 1, 2, 4 and 8
    wavefronts per chain (the two-phase kernels; 4 and 8 with the candidate producers): calc_model at random
    points (rel 1e-12), a run with swaps (rows rel 1e-9, counters exact), and the calibration (status
    and sweep counts exact)"""
    import torch
    n_chain = 6
    s, st, lad, rng, data, box = _pair(name, n_chain, seed=61, waves=waves)
    assert s.geometry == (waves, False) and not s.launch_policy[0]
    rs = np.random.RandomState(8)
    pts = box["pmin"] + (box["pmax"] - box["pmin"]) * rs.uniform(0.05, 0.95, (40, len(box["start"])))
    betas = rs.uniform(0.05, 1, 40)
    prob, prior = s.loglike(pts, betas)
    ref = [orc.loglike(CASES[name][1], p, data, beta=b) for p, b in zip(pts, betas)]
    np.testing.assert_allclose(prob, [r[0] for r in ref], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(prior, [r[1] for r in ref], rtol=1e-12, atol=1e-300)
    n_rounds, n_swap = 30, 7
    d = torch.zeros((n_rounds * n_swap, n_chain, len(box["start"]) + 2), dtype=torch.float64, device="cuda")
    s.run_sampler(n_rounds, n_swap, d.data_ptr())
    s.synchronize()
    rows = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(s.get_state(), lad, rng, what="user model run " + name)
    np.testing.assert_allclose(d.cpu().numpy(), rows, rtol=1e-9, atol=1e-300)
    dcfg = capi.calib_defaults(burn_in_iterations=400, iter_limit=20000)
    ocfg = orc.calib_defaults(burn_in_iterations=400, iter_limit=20000)
    status, iters = s.markov_chain_calibrate(0, n_chain, dcfg)
    for c in range(n_chain):
        assert (int(status[c]), int(iters[c])) == orc.markov_chain_calibrate(lad, rng, c, ocfg), c
    assert_match(s.get_state(), lad, rng, what="user model calibration " + name)
    s.close()


def test_user_model_with_a_variant_flag_and_bad_sources(tmp_path):
    """the run-time compilation follows the sampler's variant (a logistic proposal law here); a source
    that does not compile, or does not exist, fails apemost_hip_create with the compiler's words"""
    s, st, lad, rng, data, box = _pair("simplesin2", 4, seed=62)
    s.close()
    s = HipSampler(wl.MODEL_USER, 2, 4, data, seed=62, flags=capi.FLAG_PROPOSAL_LOGISTIC,
                   device_model_source=os.path.join(MODELS, "simplesin2.hip"))
    s.set_state(st)
    lad.proposal = orc.PROPOSAL_LOGISTIC
    s.run_sampler(20, 5)
    s.synchronize()
    orc.run_sampler(lad, rng, 20, 5)
    assert_match(s.get_state(), lad, rng, what="user model, logistic proposals")
    s.close()
    bad = tmp_path / "bad.hip"
    bad.write_text('#include "apemost_device_model.h"\n__device__ double apemost_user_term(const apemost_model_ctx *c, int i) { return nonsense; }\n')
    for src, words in ((bad, "nonsense"), (tmp_path / "missing.hip", "cannot be read")):
        with pytest.raises(capi.ApemostHipError) as err:
            HipSampler(wl.MODEL_USER, 2, 4, data, seed=1, device_model_source=src)
        assert words in str(err.value)
    with pytest.raises(capi.ApemostHipError):
        HipSampler(wl.MODEL_USER, 2, 4, data, seed=1)                      # no source named
    with pytest.raises(capi.ApemostHipError):
        HipSampler(wl.MODEL_SIMPLESIN, 4, 4, data, seed=1, device_model_source=bad)   # a source for a built-in model


def test_c_application_with_its_own_likelihood_runs_through_a_device_model(tmp_path):
    """A C application whose calc_model() is none of the built-in models (examples/sine2_model.c = the
    model of the reference's apps/simplesin2.c): without a device source the phases stop (there is no
    CPU sampler); with APEMOST_DEVICE_MODEL_SRC naming a source that reproduces the plugin the three
    phases run on the GPU and calibration_results equals the Python mirror's, which equals the
    oracle's sine2 model to 1e-9; a device source that computes something else is refused."""
    n_beta, burn, iters = 5, 400, 2000
    data, box = _sine2(200)
    work = tmp_path / "w"
    work.mkdir()
    (work / "params").write_text("".join("%.15e\t%.15e\t%.15e\t%s\t-1\n" % (s0, lo, hi, nm) for s0, lo, hi, nm in
                                         zip(box["start"], box["pmin"], box["pmax"], ("amplitude", "frequency"))))
    (work / "data").write_text("".join("%.17e\t%.17e\n" % tuple(r) for r in data))
    exe = hostlib.make(str(tmp_path / "sine2.exe"), app=os.path.join(hostlib.HOST, "examples", "sine2_model.c"),
                       ccflags="-DN_BETA=%d -DBURN_IN_ITERATIONS=%d -DMAX_ITERATIONS=%d" % (n_beta, burn, iters))
    env = dict(os.environ, APEMOST_SEED="13")
    env.pop("APEMOST_DEVICE_MODEL_SRC", None)
    none = subprocess.run([exe, "calibrate_first"], cwd=str(work), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert none.returncode == 1 and b"APEMOST_DEVICE_MODEL_SRC" in none.stderr
    wrong = dict(env, APEMOST_DEVICE_MODEL_SRC=os.path.join(MODELS, "normal.hip"))
    bad = subprocess.run([exe, "calibrate_first"], cwd=str(work), env=wrong, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode == 1 and b"not the device model of APEMOST_DEVICE_MODEL_SRC" in bad.stderr
    env["APEMOST_DEVICE_MODEL_SRC"] = os.path.join(MODELS, "simplesin2.hip")
    for phase in ("calibrate_first", "calibrate_rest", "run"):
        subprocess.check_call([exe, phase], cwd=str(work), env=env, stdout=subprocess.DEVNULL)
    c_calib = (work / "calibration_results").read_text()
    assert len(np.loadtxt(str(work / "amplitude-chain-0.prob.dump"))) == iters

    rt = lambda a: np.array([float("%.15e" % v) for v in np.ravel(a)]).reshape(np.shape(a))
    file_data = np.loadtxt(str(work / "data"))
    step = rt((box["pmax"] - box["pmin"]) * 0.1)                    # step < 0 in the params file: 10 % of the range
    dcfg = capi.calib_defaults(burn_in_iterations=burn)
    mk = lambda n: LadderState.from_params(n, rt(box["start"]), rt(box["pmin"]), rt(box["pmax"]), step)
    src = os.path.join(MODELS, "simplesin2.hip")
    s = HipSampler(wl.MODEL_USER, 2, 1, file_data, seed=13, device_model_source=src)
    s.set_state(mk(1))
    assert s.calibrate_first(dcfg) == 0
    first = s.get_state()
    s.close()
    s = HipSampler(wl.MODEL_USER, 2, n_beta, file_data, seed=13, device_model_source=src)
    st = mk(n_beta)
    st.beta[0], st.step[0], st.params[0] = rt(first.beta[0]), rt(first.step[0]), rt(first.params[0])
    st.params_best[0] = st.params[0]
    s.set_state(st)
    status, beta_0, _ = s.calibrate_rest(dcfg)
    assert status == 0
    assert s.get_state().calibration_results_text() == c_calib
    s.close()
    # ... and the oracle's restatement of the app, from the same start
    lad = orc.Ladder(orc.MODEL_SINE2, n_beta, 2, file_data)
    to_oracle(mk(n_beta), lad)
    rng = orc.Rng(orc.RNG_STREAMS, 13, lad)
    ocfg = orc.calib_defaults(burn_in_iterations=burn)
    assert orc.calibrate_first(lad, rng, ocfg) == 0
    lad.beta[0], lad.step[0], lad.params[0] = rt(lad.beta[0]), rt(lad.step[0]), rt(lad.params[0])
    lad.params_best[0] = lad.params[0]
    o_status, o_beta_0, _ = orc.calibrate_rest(lad, rng, ocfg)
    assert o_status == 0 and abs(o_beta_0 - beta_0) <= 1e-9 * beta_0
    got = np.loadtxt(str(work / "calibration_results"))
    np.testing.assert_allclose(got[:, 0], lad.beta, rtol=1e-9)
    np.testing.assert_allclose(got[:, 1:3], lad.step, rtol=1e-9)
    np.testing.assert_allclose(got[:, 3:5], lad.params, rtol=1e-9)


def test_small_user_model_ladder_gets_several_waves_per_chain_and_runs_faster_for_it(capsys):
    """The reference's ladders are <= 99 chains (src/parallel_tempering.c:366): one wave per chain leaves most
    of the chip idle.  A user model gets the workgroup shape a built-in sine model would get from
    choose_waves (minus the one-barrier kernel: a user's finish() is an arbitrary function of the data sum)
    -- 16 chains x 8192 points: eight waves, against one wave per chain at least 4x the steps/s; 16 x 1024:
    four waves.  Prints the rates and hiprtc's compile time for the 16 kernels of the model."""
    import time
    src = os.path.join(MODELS, "simplesin2.hip")
    rates = {}
    for n_data, expect in ((8192, 8), (1024, 4)):
        data, box = _sine2(n_data)
        n_chain = 16
        st = LadderState.from_params(n_chain, box["start"], box["pmin"], box["pmax"], (box["pmax"] - box["pmin"]) * 0.03)
        for waves in (0, 1):
            s = HipSampler(wl.MODEL_USER, 2, n_chain, data, seed=3, waves_per_chain=waves, device_model_source=src)
            if waves == 0:
                assert s.geometry == (expect, False)
                compile_s = s.user_model_compile_seconds
            s.set_state(st)
            s.calc_model(0, n_chain)
            s.run_sampler(4, 50)
            s.synchronize()
            t0 = time.perf_counter()
            s.run_sampler(200, 50)
            s.synchronize()
            rates[(n_data, waves)] = 200 * 50 * n_chain / (time.perf_counter() - t0)
            final = s.get_state()
            s.close()
            if waves == 0:
                keep = final
            else:      # same draws, same decisions unless a comparison lands inside the rounding of the data sum
                assert np.array_equal(final.accept, keep.accept) and np.allclose(final.params, keep.params, rtol=1e-9)
    with capsys.disabled():
        print("\n[user model, 16 chains] steps/s: 8192 points %.3g (8 waves) vs %.3g (1 wave); 1024 points %.3g (4 waves) vs "
              "%.3g (1 wave); hiprtc %.2f s" % (rates[(8192, 0)], rates[(8192, 1)], rates[(1024, 0)], rates[(1024, 1)], compile_s))
    assert rates[(8192, 0)] >= 4 * rates[(8192, 1)]
    assert rates[(1024, 0)] >= 1.5 * rates[(1024, 1)]
