"""GPU tests of the C host layer's single-chain API and of its sample sinks: the reference's
functions markov_chain_step, markov_chain_step_for, burn_in, markov_chain_calibrate,
assess_acceptance_rate (src/markov_chain.c), tempering_interaction
(src/parallel_tempering_interaction.c:125-141), set_function (apps/library.c), called the way a
reference application calls them -- on `mcmc` objects, with a parallel_tempering_mcmc the
application allocated itself -- and compared with the CPU oracle."""
import ctypes as C
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from apemost_amd import build, workloads as wl
from oracle import oracle as orc
from tests import hostlib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SINE = os.path.join(hostlib.HOST, "examples", "sine_model.c")
SEED = 19


def _inputs(work, w):
    work.mkdir(exist_ok=True)
    (work / "params").write_text(w.params_file_text())
    (work / "data").write_text(w.data_file_text())
    return str(work / "params").encode(), str(work / "data").encode()


def _rt(a):
    return np.array([float("%.15e" % v) for v in np.ravel(a)]).reshape(np.shape(a))


def _oracle_chain(w, data, chain_id, beta):
    lad = orc.Ladder.from_params(w.model, 1, _rt(w.start), _rt(w.pmin), _rt(w.pmax), _rt(w.step), data,
                                 chain_offset=chain_id)
    lad.beta[0] = beta
    return lad, orc.Rng(orc.RNG_STREAMS, SEED, lad)


def _assert_chain(m, lad, c=0, what=""):
    mc = m.contents
    n = mc.n_par
    assert (mc.accept, mc.reject, mc.n_iter) == (lad.accept[c], lad.reject[c], lad.n_iter[c]), what
    assert [mc.params_accepts[p] for p in range(n)] == list(lad.params_accepts[c]), what
    assert [mc.params_rejects[p] for p in range(n)] == list(lad.params_rejects[c]), what
    np.testing.assert_allclose(hostlib.vec(mc.params), lad.params[c], rtol=1e-9, err_msg=what)
    np.testing.assert_allclose(hostlib.vec(mc.params_best), lad.params_best[c], rtol=1e-9, err_msg=what)
    np.testing.assert_allclose(hostlib.vec(mc.params_step), lad.step[c], rtol=1e-9, err_msg=what)
    np.testing.assert_allclose([mc.prob, mc.prob_best], [lad.prob[c], lad.prob_best[c]], rtol=1e-9, err_msg=what)


_SCRIPT = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, {root!r})
from tests import hostlib
L = hostlib.load({lib!r})
m = L.mcmc_load({params!r}, {data!r})
keep = hostlib.attach_tempering(L, m, 0.6)
L.calc_model(m, None)
for k in range(25):
    L.markov_chain_step(m)
print(repr((hostlib.vec(m.contents.params), m.contents.prob, m.contents.accept)))
'''


def test_single_chain_api_matches_oracle(tmp_path, monkeypatch):
    """markov_chain_step / _step_for / burn_in / markov_chain_calibrate / assess_acceptance_rate on a
    chain set up the reference way (mcmc_load + a malloc'ed two-field parallel_tempering_mcmc +
    set_beta): same draws, same decisions as the oracle at the chain's ladder position; the RNG
    address lives in the bridge's side table, so nothing is read from uninitialised memory"""
    monkeypatch.setenv("APEMOST_SEED", str(SEED))
    build.build_hip()
    lib = hostlib.make(str(tmp_path / "libsine.so"), app=SINE, ccflags="-DN_BETA=4", shared=True)
    L = hostlib.load(lib)
    w = wl.simplesin(n_data=200, n_chain=4)
    w.step = w.step * 0.05                            # narrow enough for the cold posterior to accept some
    params, data = _inputs(tmp_path / "w", w)
    data_m = np.loadtxt(data.decode())
    m = L.mcmc_load(params, data)
    keep = hostlib.attach_tempering(L, m, 0.6)
    L.apemost_chain_place(m, 3)                       # ladder position 3 selects the RNG streams
    lad, rng = _oracle_chain(w, data_m, 3, 0.6)
    L.calc_model(m, None)                             # host plugin
    orc.calc_model(lad, 0)
    assert abs(m.contents.prob - lad.prob[0]) <= 1e-12 * abs(lad.prob[0])
    # markov_chain_step: counters move, n_iter and the best point do not (src/markov_chain.c:369-386)
    for k in range(40):
        L.markov_chain_step(m)
        orc.step(lad, rng, 0)
    assert m.contents.n_iter == 0 and m.contents.prob_best == -1e10
    _assert_chain(m, lad, what="markov_chain_step")
    assert 0 < m.contents.accept < 40
    # markov_chain_step_for + mcmc_check_best, as the calibration loop calls them
    for k in range(30):
        L.markov_chain_step_for(m, k % 4)
        L.mcmc_check_best(m)
        orc.step_for(lad, rng, 0, k % 4)
        orc.check_best(lad, 0)
    _assert_chain(m, lad, what="markov_chain_step_for")
    assert m.contents.accept + m.contents.reject == 40     # quirk Q5: F5 leaves the global counters alone
    # burn_in (src/markov_chain.c:34-79)
    L.burn_in(m, 600)
    orc.burn_in(lad, rng, 0, 600)
    _assert_chain(m, lad, what="burn_in")
    # markov_chain_calibrate = burn_in + calibrate_orig (src/markov_chain_calibrate.c:1182-1204)
    L.markov_chain_calibrate(m, 400, 0.5, 0.01, 20000, 0.85, 0.5)
    cfg = orc.calib_defaults(burn_in_iterations=400, iter_limit=20000)
    assert orc.markov_chain_calibrate(lad, rng, 0, cfg)[0] == 0
    _assert_chain(m, lad, what="markov_chain_calibrate")
    # assess_acceptance_rate (src/markov_chain.c:117-224) for one parameter and for the whole step
    for param in (1, 4):
        rate, acc = C.c_double(0), C.c_double(0)
        n = L.assess_acceptance_rate(m, param, 0.5, 0.0, 1.0, C.byref(rate), C.byref(acc))
        ref_n, ref_rate, ref_acc = _oracle_assess(lad, rng, param, 0.5, 0.0, 1.0)
        assert (n, rate.value, acc.value) == (ref_n, ref_rate, ref_acc)
        assert n >= 40 and 0 < rate.value < 1
        _assert_chain(m, lad, what="assess_acceptance_rate %d" % param)
    L.mcmc_free(m)
    del keep
    # two fresh processes give the same chain: no uninitialised memory enters the RNG address
    outs = [subprocess.check_output([sys.executable, "-c", _SCRIPT.format(root=ROOT, lib=lib, params=params, data=data)],
                                    env=dict(os.environ, APEMOST_SEED=str(SEED))) for _ in range(2)]
    assert outs[0] == outs[1] and b"nan" not in outs[0]


def _oracle_assess(lad, rng, param, desired, min_acc, max_acc):
    """assess_acceptance_rate restated on oracle steps (src/markov_chain.c:117-224), including the
    rate taken from the counter before the last step and the truncated drift"""
    single = param < lad.n_par
    counter = (lambda: int(lad.params_accepts[0, param])) if single else (lambda: int(lad.accept[0]))
    orc.lib().orc_reset_accept_rejects(C.byref(lad.c_state()), 0)
    log, n, before = [], 40, 0
    while True:
        while len(log) < n:
            before = counter()
            if single:
                orc.step_for(lad, rng, 0, param)
            else:
                orc.step(lad, rng, 0)
            orc.check_best(lad, 0)
            log.append(counter() != before)
        rate = before / float(n)
        running, drift = 0, 1
        for j in range(n):
            running += log[j]
            drift = max(drift, abs(int(running - rate * j)))
        wanted = min(max(abs(rate - desired) * 0.25, 0.005, min_acc), max_acc)
        accuracy = drift / 1. / n
        if accuracy <= wanted:
            return n, rate, accuracy
        n = (int((drift / 1. / wanted) / 8) + 1) * 8


def test_tempering_interaction_on_host_chains_matches_oracle(tmp_path, monkeypatch):
    """tempering_interaction(chains, n_beta, iter): 60 attempts on a ladder of host chain objects;
    pairs, decisions, exchanged positions and swap counts equal the oracle's"""
    monkeypatch.setenv("APEMOST_SEED", str(SEED))
    build.build_hip()
    n_beta = 5
    lib = hostlib.make(str(tmp_path / "libsine.so"), app=SINE, ccflags="-DN_BETA=%d" % n_beta, shared=True)
    L = hostlib.load(lib)
    w = wl.simplesin(n_data=64, n_chain=n_beta)
    params, data = _inputs(tmp_path / "w", w)
    data_m = np.loadtxt(data.decode())
    lad = orc.Ladder.from_params(w.model, n_beta, _rt(w.start), _rt(w.pmin), _rt(w.pmax), _rt(w.step), data_m)
    rng = orc.Rng(orc.RNG_STREAMS, SEED, lad)
    chains = (C.POINTER(hostlib.Mcmc) * n_beta)()
    keep = []
    rs = np.random.RandomState(4)
    for i in range(n_beta):
        chains[i] = L.mcmc_load(params, data) if i == 0 else L.mcmc_load_params(params)
        if i:
            L.mcmc_reuse_data(chains[i], chains[0])
        beta = orc.get_chain_beta(0, i, n_beta, 0.3)
        keep.append(hostlib.attach_tempering(L, chains[i], beta))
        L.apemost_chain_place(chains[i], i)
        p = w.pmin + (w.pmax - w.pmin) * rs.uniform(0.2, 0.8, 4)
        hostlib.set_vec(chains[i].contents.params, p)
        hostlib.set_vec(chains[i].contents.params_best, p)
        L.calc_model(chains[i], None)
        chains[i].contents.prob_best = chains[i].contents.prob + i
        lad.beta[i], lad.params[i], lad.params_best[i] = beta, p, p
        orc.calc_model(lad, i)
        lad.prob_best[i] = lad.prob[i] + i
    for r in range(60):
        L.tempering_interaction(chains, n_beta, r)
        orc.tempering_interaction(lad, rng)
    assert lad.swapcount.sum() > 5
    for i in range(n_beta):
        assert keep[i].swapcount == lad.swapcount[i]
        np.testing.assert_allclose(hostlib.vec(chains[i].contents.params), lad.params[i], rtol=1e-12)
        np.testing.assert_allclose(hostlib.vec(chains[i].contents.params_best), lad.params_best[i], rtol=1e-12)
        assert abs(chains[i].contents.prob_best - lad.prob_best[i]) <= 1e-12 * abs(lad.prob_best[i])
        assert abs(chains[i].contents.prob - lad.prob[i]) <= 1e-12 * abs(lad.prob[i])      # quirk Q1: prob stays
    for i in range(1, n_beta):
        chains[i].contents.data = None            # aliased: chain 0 frees the matrix
    for i in range(n_beta):
        L.mcmc_free(chains[i])


def _read_bin(path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import samples_bin
    hdr, params, probs = samples_bin.read(path)
    return hdr, np.array(params), np.array(probs)


def test_binary_and_thinned_sinks_carry_the_same_samples_as_the_text_dumps(tmp_path):
    """APEMOST_DUMP=binary / thin:N (second half): samples.bin holds the doubles the text
    dumps print; thinning keeps iterations N, 2N, ...; the default text mode is untouched"""
    n_beta, iters = 8, 6000                          # whole rounds: n_swap = 2000 / 8 = 250 divides 6000
    w = wl.simplesin(n_data=128, n_chain=n_beta)
    exe = hostlib.make(str(tmp_path / "sine.exe"), ccflags="-DN_BETA=%d -DBURN_IN_ITERATIONS=600 -DMAX_ITERATIONS=%d" % (n_beta, iters))
    runs = {}
    for mode in ("text", "binary", "thin:7", "binary:all,thin:7"):
        work = tmp_path / mode.replace(":", "_").replace(",", "_")
        _inputs(work, w)
        env = dict(os.environ, APEMOST_SEED="3", APEMOST_DUMP=mode)
        if mode == "text":
            del env["APEMOST_DUMP"]                  # the default
        for phase in ("calibrate_first", "calibrate_rest", "run"):
            subprocess.check_call([exe, phase], cwd=str(work), env=env, stdout=subprocess.DEVNULL)
        runs[mode] = work
    t = runs["text"]
    amp = np.loadtxt(str(t / "amplitude-chain-0.prob.dump"))
    prob2 = np.loadtxt(str(t / "prob-chain2.dump"))
    assert len(amp) == iters and not os.path.exists(str(t / "samples.bin"))
    hdr, params, probs = _read_bin(str(runs["binary"] / "samples.bin"))
    assert hdr == dict(version=2, n_beta=n_beta, n_par=4, n_swap=2000 // n_beta, thin=1, n_param_chains=1)
    assert params.shape == (iters, 1, 4) and probs.shape == (iters, n_beta, 2)
    assert np.array_equal(_rt(params[:, 0, 0]), amp)
    np.testing.assert_allclose(probs[:, 2], prob2, rtol=2e-6)
    assert not os.path.exists(str(runs["binary"] / "prob-chain0.dump"))
    for f in ("acceptance_rate.dump", "calibration_results"):
        assert (runs["binary"] / f).read_text() == (t / f).read_text()
    thin_amp = np.loadtxt(str(runs["thin:7"] / "amplitude-chain-0.prob.dump"))
    assert np.array_equal(thin_amp, amp[6::7])                      # iterations 7, 14, ...
    assert np.array_equal(np.loadtxt(str(runs["thin:7"] / "prob-chain2.dump")), prob2[6::7])
    hdr7, params7, probs7 = _read_bin(str(runs["binary:all,thin:7"] / "samples.bin"))
    assert hdr7["thin"] == 7 and hdr7["n_param_chains"] == n_beta
    assert np.array_equal(params7[:, 0], params[6::7, 0]) and np.array_equal(probs7, probs[6::7])
    assert params7.shape == (len(amp[6::7]), n_beta, 4)
    # tools/samples_bin.py expands the binary file into the reference's text files
    out = tmp_path / "expanded"
    out.mkdir()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "samples_bin.py"), str(runs["binary"] / "samples.bin"),
                           "--params", str(runs["binary"] / "params"), "--text", str(out)])
    assert (out / "amplitude-chain-0.prob.dump").read_text() == (t / "amplitude-chain-0.prob.dump").read_text()
    assert (out / "prob-chain2.dump").read_text() == (t / "prob-chain2.dump").read_text()


def test_run_phase_with_more_chains_than_file_descriptors(tmp_path):
    """a ladder of 1100 chains through the drop-in C host: the reference keeps one FILE* per chain open
    (and asserts n_beta < 100); here prob-chain files are appended per batch, so RLIMIT_NOFILE = 1024
    is not a limit (BASELINE configs 3-5 have 1024 and 2048 chains)"""
    n_beta, iters = 1100, 40
    w = wl.simplesin(n_data=64, n_chain=n_beta)
    exe = hostlib.make(str(tmp_path / "sine.exe"), ccflags="-DN_BETA=%d -DN_SWAP=2 -DMAX_ITERATIONS=%d" % (n_beta, iters))
    work = tmp_path / "w"
    _inputs(work, w)
    from apemost_amd.state import LadderState
    from apemost_amd.sampler import get_chain_beta
    st = LadderState.from_params(n_beta, w.start, w.pmin, w.pmax, w.step * 0.3)
    for i in range(n_beta):
        st.beta[i] = get_chain_beta(0, i, n_beta, 0.05)
    (work / "calibration_results").write_text(st.calibration_results_text())
    import resource
    soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)

    def limit():
        resource.setrlimit(resource.RLIMIT_NOFILE, (min(1024, hard), hard))
    subprocess.check_call([exe, "run"], cwd=str(work), env=dict(os.environ, APEMOST_SEED="8"), stdout=subprocess.DEVNULL,
                          preexec_fn=limit)
    for i in (0, 517, n_beta - 1):
        assert np.loadtxt(str(work / ("prob-chain%d.dump" % i))).shape == (iters, 2)
    assert len(np.loadtxt(str(work / "amplitude-chain-0.prob.dump"))) == iters


_LIB_SCRIPT = r'''
import ctypes as C, math, os, sys
sys.path.insert(0, {root!r})
from tests import hostlib
L = hostlib.load({lib!r})
SIGMA = 0.5
def loglike(mp, old):
    m = mp.contents
    a, f, ph, o = (m.params.contents.data[i] for i in range(4))
    d = m.data.contents
    s = 0.0
    for i in range(d.size1):
        x, y = d.data[i * d.tda], d.data[i * d.tda + 1]
        r = a * math.sin(2.0 * math.pi * (f * x + ph)) + o - y
        s += r * r
    return s / (-2 * SIGMA * SIGMA)
def prior(mp, old):
    return {prior}
cbs = (hostlib.CALLBACK(loglike), hostlib.CALLBACK(prior))
L.set_function.argtypes = [hostlib.CALLBACK, hostlib.CALLBACK]
L.set_function(*cbs)
L.calibrate_first()
print("CALIBRATED")
'''


def test_library_flavour_samples_on_the_device_only_when_the_callbacks_match_a_device_model(tmp_path):
    """libapemost.so + set_function: a registered (LogLike, Prior) pair that equals the device's
    simplesin runs calibrate_first on the GPU and writes the same calibration_results as the
    executable built from the C plugin; a pair with a non-zero prior (tempered with beta by the
    library, quirk Q6) matches no device model and stops -- there is no CPU sampler"""
    n_beta = 4
    flags = "-DN_BETA=%d -DBURN_IN_ITERATIONS=400" % n_beta
    lib = hostlib.make(str(tmp_path / "libapemost.so"), ccflags=flags, shared=True)
    exe = hostlib.make(str(tmp_path / "sine.exe"), ccflags=flags)
    w = wl.simplesin(n_data=96, n_chain=n_beta)
    env = dict(os.environ, APEMOST_SEED="6")
    for name in ("lib", "exe", "bad"):
        _inputs(tmp_path / name, w)
    subprocess.check_call([exe, "calibrate_first"], cwd=str(tmp_path / "exe"), env=env, stdout=subprocess.DEVNULL)
    out = subprocess.check_output([sys.executable, "-c", _LIB_SCRIPT.format(root=ROOT, lib=lib, prior="0.0")],
                                  cwd=str(tmp_path / "lib"), env=env)
    assert b"CALIBRATED" in out
    assert (tmp_path / "lib" / "calibration_results").read_text() == (tmp_path / "exe" / "calibration_results").read_text()
    bad = subprocess.run([sys.executable, "-c", _LIB_SCRIPT.format(root=ROOT, lib=lib, prior="-1.5 * mp.contents.params.contents.data[0]")],
                         cwd=str(tmp_path / "bad"), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode == 1 and b"matches none of the device" in bad.stderr and b"CALIBRATED" not in bad.stdout
    assert not os.path.exists(str(tmp_path / "bad" / "calibration_results"))


def test_c_host_with_two_shards_equals_one_device(tmp_path):
    """APEMOST_DEVICES=0,0: the C host splits the ladder into two shards (both on this box's only GPU),
    each with its own sampler and stream; edge records cross with hipMemcpyPeerAsync inside
    apemost_hip_run_shards.  Every output file equals the single-shard run byte for byte."""
    n_beta, iters = 10, 4000
    w = wl.simplesin(n_data=128, n_chain=n_beta)
    exe = hostlib.make(str(tmp_path / "sine.exe"),
                       ccflags="-DN_BETA=%d -DBURN_IN_ITERATIONS=600 -DMAX_ITERATIONS=%d -DN_SWAP=5" % (n_beta, iters))
    outs = {}
    for name, devices in (("one", None), ("two", "0,0"), ("three", "0,0,0")):
        work = tmp_path / name
        _inputs(work, w)
        env = dict(os.environ, APEMOST_SEED="12")
        if devices:
            env["APEMOST_DEVICES"] = devices
        for phase in ("calibrate_first", "calibrate_rest", "run"):
            subprocess.check_call([exe, phase], cwd=str(work), env=env, stdout=subprocess.DEVNULL)
        outs[name] = work
    files = ["calibration_results", "acceptance_rate.dump", "amplitude-chain-0.prob.dump", "phase-chain-0.prob.dump"] + \
            ["prob-chain%d.dump" % i for i in range(n_beta)]
    for name in ("two", "three"):
        for f in files:
            assert (outs[name] / f).read_bytes() == (outs["one"] / f).read_bytes(), (name, f)
    assert len((outs["one"] / "prob-chain9.dump").read_text().splitlines()) == iters


def test_run_shards_abi_equals_whole_ladder_and_oracle():
    """apemost_hip_run_shards / apemost_hip_edge_exchange driven through ctypes: three shards on one
    device against the whole ladder on one sampler and against the oracle"""
    import torch
    from apemost_amd import capi
    from apemost_amd.sampler import HipSampler
    from tests.helpers import assert_match, make_pair
    w = wl.pulse(n_data=96, n_chain=11)
    n_global, n_rounds, n_swap, seed = 11, 200, 3, 29
    st, lad, rng = make_pair(w, n_global, seed=seed)
    whole = HipSampler(w.model, w.n_par, n_global, w.data, seed=seed)
    whole.set_state(st)
    d_whole = torch.zeros((n_rounds * n_swap, n_global, w.n_par + 2), dtype=torch.float64, device="cuda")
    whole.run_sampler(n_rounds, n_swap, d_whole.data_ptr())
    whole.synchronize()
    ref = whole.get_state()
    whole.close()
    bounds = [(0, 4), (4, 7), (7, 11)]
    shards = []
    for lo, hi in bounds:
        s = HipSampler(w.model, w.n_par, hi - lo, w.data, seed=seed, chain_offset=lo, n_chains_global=n_global)
        s.set_state(st.slice(lo, hi))
        shards.append(s)
    bufs = [torch.zeros((n_rounds * n_swap, hi - lo, w.n_par + 2), dtype=torch.float64, device="cuda") for lo, hi in bounds]
    torch.cuda.synchronize()
    handles = (C.c_void_p * 3)(*[s._h for s in shards])
    ptrs = (C.c_void_p * 3)(*[b.data_ptr() for b in bufs])
    L = capi.lib()
    for part in (70, 130):                      # two calls: the swap position carries over
        off = 0 if part == 70 else 70
        p2 = (C.c_void_p * 3)(*[b[off * n_swap:].data_ptr() for b in bufs])
        capi.check(L.apemost_hip_run_shards(handles, 3, part, n_swap, p2))
    for s in shards:
        s.synchronize()
    got = [s.get_state() for s in shards]
    for f in ("params", "params_best", "prob", "prob_best", "prior", "accept", "reject", "swapcount", "ticks", "n_iter"):
        assert np.array_equal(np.concatenate([getattr(g, f) for g in got]), getattr(ref, f)), f
    assert np.array_equal(torch.cat(bufs, dim=1).cpu().numpy(), d_whole.cpu().numpy())
    assert ref.swapcount[3] + ref.swapcount[6] > 0          # the shard edges were crossed
    orc_samples = orc.run_sampler(lad, rng, n_rounds, n_swap, record=True)
    assert_match(ref, lad, rng, what="run_shards")
    np.testing.assert_allclose(d_whole.cpu().numpy(), orc_samples, rtol=1e-9, atol=1e-300)
    # mismatched shards are refused
    rc = L.apemost_hip_run_shards((C.c_void_p * 2)(shards[0]._h, shards[2]._h), 2, 1, 1, None)
    assert rc == capi.ERR_INVALID
    for s in shards:
        s.close()


def test_c_application_built_with_the_variant_macros_equals_python_mirror_and_oracle(tmp_path):
    """-DPROPOSAL_LOGISTIC -DRANDOMSWAP -DADAPT on the application's compile line (the reference's own
    knobs, src/mcmc_gettersetter.c:290-305, parallel_tempering_interaction.c:130-131,
    parallel_tempering.c:282-301) reach the engine as apemost_hip_config.flags: the three phases of
    the C application equal the Python mirror created with the same flags, and the run phase equals
    the oracle's"""
    from apemost_amd import capi
    from apemost_amd.sampler import HipSampler
    from apemost_amd.state import LadderState
    import torch
    n_beta, burn, iters = 8, 800, 8000                # 32 rounds of 250; ADAPT starts after 5000 steps
    w = wl.simplesin(n_data=128, n_chain=n_beta)
    work = tmp_path / "w"
    _inputs(work, w)
    exe = hostlib.make(str(tmp_path / "sine.exe"),
                       ccflags="-DN_BETA=%d -DBURN_IN_ITERATIONS=%d -DMAX_ITERATIONS=%d -DPROPOSAL_LOGISTIC -DRANDOMSWAP "
                               "-DADAPT -DTARGET_ACCEPTANCE_RATE=0.4" % (n_beta, burn, iters))
    env = dict(os.environ, APEMOST_SEED="7")
    c_progress = {}
    for phase in ("calibrate_first", "calibrate_rest", "run"):
        subprocess.check_call([exe, phase], cwd=str(work), env=env, stdout=subprocess.DEVNULL)
        c_progress[phase] = (work / "calibration_progress.data").read_text()
    assert c_progress["run"] == c_progress["calibrate_rest"]      # the run phase leaves the file alone
    c_calib = (work / "calibration_results").read_text()
    c_amp = np.loadtxt(str(work / "amplitude-chain-0.prob.dump"))
    c_accept = (work / "acceptance_rate.dump").read_text().strip().splitlines()[-1].split()

    flags = capi.FLAG_PROPOSAL_LOGISTIC | capi.FLAG_RANDOMSWAP | capi.FLAG_ADAPT
    data = np.loadtxt(str(work / "data"))
    dcfg = capi.calib_defaults(burn_in_iterations=burn, rat_limit=0.4, target_global=0.4)
    mk = lambda: LadderState.from_params(n_beta, _rt(w.start), _rt(w.pmin), _rt(w.pmax), _rt(w.step))
    s = HipSampler(w.model, 4, 1, data, seed=7, flags=flags, adapt_target=0.4)
    s.set_state(mk().slice(0, 1))
    assert s.calibrate_first(dcfg) == 0
    # calibration_progress.data (src/markov_chain_calibrate.c:1143-1146): chain 0's readjustments
    assert s.calibration_progress_text() == c_progress["calibrate_first"] != ""
    first = s.get_state()
    s.close()
    s = HipSampler(w.model, 4, n_beta, data, seed=7, flags=flags, adapt_target=0.4)
    st = mk()
    st.beta[0], st.step[0], st.params[0] = _rt(first.beta[0]), _rt(first.step[0]), _rt(first.params[0])
    st.params_best[0] = st.params[0]
    s.set_state(st)
    status, beta_0, _ = s.calibrate_rest(dcfg)
    assert status == 0
    assert s.get_state().calibration_results_text() == c_calib
    # ... and after calibrate_rest the lines of the ladder's last chain (the last one to reopen the file)
    assert s.calibration_progress_text() == c_progress["calibrate_rest"] != c_progress["calibrate_first"]
    assert c_progress["calibrate_rest"].startswith("0\t200\t")
    s.close()

    s = HipSampler(w.model, 4, n_beta, data, seed=7, flags=flags, adapt_target=0.4)
    st = mk()
    st.read_calibration_results(c_calib)
    s.set_state(st)
    n_swap = 2000 // n_beta
    d = torch.zeros((iters, n_beta, 6), dtype=torch.float64, device="cuda")
    s.run_sampler(iters // n_swap, n_swap, d.data_ptr())
    s.synchronize()
    run = s.get_state()
    s.close()
    samples = d.cpu().numpy()
    assert np.array_equal(c_amp, _rt(samples[:, 0, 0]))
    assert [int(t) for t in c_accept] == [iters] + [int(a) for a in run.accept]
    assert not np.allclose(run.step, st.step)          # adapt() moved the widths during the run

    lad = orc.Ladder(w.model, n_beta, 4, data)
    for n in ("params", "params_best", "step", "pmin", "pmax", "beta"):
        getattr(lad, n)[...] = getattr(st, n)
    lad.proposal, lad.randomswap, lad.adapt, lad.adapt_target = orc.PROPOSAL_LOGISTIC, 1, 1, 0.4
    rng = orc.Rng(orc.RNG_STREAMS, 7, lad)
    ref = orc.run_sampler(lad, rng, iters // n_swap, n_swap, record=True, n_threads=8)
    np.testing.assert_allclose(samples, ref, rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(run.step, lad.step, rtol=1e-12)
    assert np.array_equal(run.accept, lad.accept) and np.array_equal(run.swapcount, lad.swapcount)
