"""The C host layer (apemost_amd/host): APEMoST's own API on top of libapemost_hip.so.

CPU part: it builds under the reference's strict flags, the reference's applications link against it
unchanged (only where /root/reference exists -- never on the GPU box), the host plugin path
(eval) reproduces the manual's known answer, the ABI library exports every declared symbol.
GPU part: the three phases of a C application (calibrate_first, calibrate_rest, run) produce
byte-identical files to the Python mirror driving the same engine, and match the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from apemost_amd import build, capi, workloads as wl
from tests import hostlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "apemost_amd", "host")
REF = "/root/reference"
STRICT = "-std=c99 -fopenmp -Wall -Werror -Wextra -ansi -pedantic"   # the reference Makefile's CFLAGS


def _make(out, app=None, main=None, ccflags="", strict=STRICT, shared=False):
    build.build_hip()
    return hostlib.make(out, app, main, ccflags, strict, shared)


def test_abi_library_exports_every_declared_symbol():
    build.build_hip()
    hdr = open(os.path.join(ROOT, "include", "apemost_hip.h")).read()
    declared = set(re.findall(r"\b(apemost_hip_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    nm = subprocess.check_output(["nm", "-D", "--defined-only", build.HIP_LIB]).decode()
    exported = set(re.findall(r" T (apemost_hip_[a-z_0-9]+)", nm))
    assert declared <= exported, declared - exported


def test_example_app_builds_strict(tmp_path):
    build.build_hip()
    exe = _make(str(tmp_path / "sine.exe"), ccflags="-DN_BETA=4")
    assert os.path.exists(exe)


def test_variant_macros_build_strict_and_rwm_is_refused(tmp_path):
    """the reference's compile-time variants (-DPROPOSAL_LOGISTIC, -DPROPOSAL_UNIFORM, -DRANDOMSWAP,
    -DADAPT, and since round 4 -DRWM) are accepted on the application's compile line (the bridge turns
    them into engine flags); what the engine does not carry -- the two proposal laws at once, -DRWM with
    its clamps redefined -- stops the build with a message"""
    build.build_hip()
    for flags in ("-DPROPOSAL_LOGISTIC -DRANDOMSWAP -DADAPT -DTARGET_ACCEPTANCE_RATE=0.4", "-DPROPOSAL_UNIFORM", "-DRWM -DADAPT"):
        assert os.path.exists(_make(str(tmp_path / "v.exe"), ccflags="-DN_BETA=4 " + flags))
        os.remove(str(tmp_path / "v.exe"))
    for flags in ("-DRWM -DMINIMAL_STEPWIDTH=0.001", "-DPROPOSAL_LOGISTIC -DPROPOSAL_UNIFORM"):
        with pytest.raises(subprocess.CalledProcessError):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "apemost_amd", "host"),
                                   "OUT=" + str(tmp_path / "no.exe"), "CCFLAGS=-DN_BETA=4 " + flags],
                                  stderr=subprocess.DEVNULL)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_apps_link_unchanged_and_eval_matches_manual(tmp_path):
    build.build_hip()
    for app in ("simplesin", "pulse", "pulse_vrot"):
        for main in ("generic_main", "eval_main", "benchmark_main"):
            _make(str(tmp_path / ("%s_%s.exe" % (main, app))), app="%s/apps/%s.c" % (REF, app),
                  main="%s/apps/%s.c" % (REF, main), ccflags="-DN_BETA=8")
    # the reference's other example likelihoods: they sample through a device model
    # given as source (APEMOST_DEVICE_MODEL_SRC, examples/device_models/), and link unchanged
    for app in ("simplesin2", "normal", "bernoulli_example"):
        _make(str(tmp_path / ("generic_main_%s.exe" % app)), app="%s/apps/%s.c" % (REF, app),
              main="%s/apps/generic_main.c" % REF, ccflags="-DN_BETA=8")
        assert os.path.exists(os.path.join(HOST, "examples", "device_models", app + ".hip"))
    # doc/manual.rst:190-213: the eval example, host plugin path only (no GPU involved)
    work = tmp_path / "w"
    work.mkdir()
    (work / "data").write_text("101\t0.67\n102\t1.01\n103\t7.9e-1\n104\t1.34\n")
    (work / "params").write_text("0\t0\t2\tamplitude\t-1\n0\t0\t0.3\tfrequency\t-1\n0\t0\t1.0\tphase\t-1\n0\t0\t2\toffset\t-1\n")
    out = subprocess.check_output([str(tmp_path / "eval_main_simplesin.exe")], input=b"1 0.2 1 0\n", cwd=str(work))
    prob, prior = (float(t) for t in out.split())
    assert abs(prob - (-1.480898044165363e+01)) < 5e-14 and prior == 0.0
    chk = subprocess.check_output([str(tmp_path / "generic_main_simplesin.exe"), "check"], cwd=str(work)).decode()
    assert "N_BETA: 8" in chk and "params\tfound" in chk


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_unit_tests_link_unchanged_and_pass(tmp_path, golden_dir):
    """the reference's own TAP suite (tests/tests.c + run-tests.c: histogram, chain construction,
    parser on its fixtures, append, RNG wrappers, mod_double, dump files) compiled unchanged against
    the host layer under the reference's flags: 8 of 8 ok.  Only the binary runs; on the GPU box the
    reference tree does not exist and this test is skipped."""
    exe = _make(str(tmp_path / "tests.exe"), app=REF + "/tests/tests.c", main=REF + "/tests/run-tests.c",
                ccflags="-DN_BETA=4")
    work = tmp_path / "w"
    (work / "tests").mkdir(parents=True)
    for f in ("testinput1", "testlc.dat"):           # the fixtures the suite reads (tests/tests.c:90,135)
        (work / "tests" / f).write_bytes(open(os.path.join(golden_dir, f), "rb").read())
    out = subprocess.check_output([exe], cwd=str(work)).decode()
    tap = [l for l in out.splitlines() if re.match(r"(not )?ok \d+", l)]
    assert tap == ["ok %d" % i for i in range(1, 9)], out[-2000:]
    assert "all 8 tests successful" in out


def test_tempering_struct_keeps_the_reference_layout(tmp_path):
    """applications allocate parallel_tempering_mcmc themselves (apps/eval_main.c:50): two fields,
    16 bytes, as src/parallel_tempering_beta.h:65-76 declares it"""
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mcmc.h"\n#include "parallel_tempering_beta.h"\n'
                   'int main(void) { printf("%lu %lu %lu\\n", (unsigned long)sizeof(parallel_tempering_mcmc),\n'
                   '(unsigned long)offsetof(parallel_tempering_mcmc, beta), (unsigned long)offsetof(parallel_tempering_mcmc, swapcount)); return 0; }\n')
    exe = str(tmp_path / "probe.exe")
    subprocess.check_call(["gcc", "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(HOST, "gsl"),
                           "-I" + os.path.join(ROOT, "include"), "-DWITHOUT_GARBAGE_COLLECTOR", str(src), "-o", exe])
    assert subprocess.check_output([exe]).split() == [b"16", b"0", b"8"]


def _write_inputs(work, w):
    work.mkdir(exist_ok=True)
    (work / "params").write_text(w.params_file_text())
    (work / "data").write_text(w.data_file_text())


def test_library_flavour_set_function_and_quirk_q6(tmp_path):
    """libapemost.so (reference Makefile:48-49, apps/library.c): a client registers LogLike and Prior;
    calc_model() then sets prior and prob = beta * (prior + loglike) -- the prior is tempered too
    (SURVEY quirk Q6).  Host path only, no GPU."""
    import ctypes as C
    lib = _make(str(tmp_path / "libapemost.so"), ccflags="-DN_BETA=4", shared=True)
    L = hostlib.load(lib)
    w = wl.simplesin(n_data=32, n_chain=4)
    _write_inputs(tmp_path / "w", w)
    m = L.mcmc_load(str(tmp_path / "w" / "params").encode(), str(tmp_path / "w" / "data").encode())
    keep = hostlib.attach_tempering(L, m, 0.25)
    calls = []

    def loglike(mp, old):
        calls.append("L")
        return -3.5 * mp.contents.params.contents.data[0] - mp.contents.data.contents.size1

    def prior(mp, old):
        calls.append("P")
        return -2.0

    cb_l, cb_p = hostlib.CALLBACK(loglike), hostlib.CALLBACK(prior)
    L.set_function.argtypes = [hostlib.CALLBACK, hostlib.CALLBACK]
    L.set_function(cb_l, cb_p)
    L.calc_model(m, None)
    assert calls == ["P", "L"]
    assert m.contents.prior == -2.0
    assert m.contents.prob == 0.25 * (-2.0 + (-3.5 * w.start[0] - 32))
    L.mcmc_free(m)
    del keep


BETA_LADDERS = ["chebyshev_beta", "equidistant_beta", "equidistant_temperature", "chebyshev_temperature",
                "equidistant_stepwidth", "chebyshev_stepwidth", "hot_chains"]


@pytest.mark.parametrize("kind,name", list(enumerate(BETA_LADDERS)))
def test_beta_ladder_and_beta_0_match_oracle(kind, name, tmp_path):
    """get_chain_beta for every BETA_ALIGNMENT (src/parallel_tempering_beta.c:53-90) and calc_beta_0
    (:92-102): the C host layer and the Python mirror against the oracle's restatement"""
    import ctypes as C
    from apemost_amd import sampler
    from apemost_amd.state import LadderState
    from oracle import oracle as orc
    order = {"chebyshev_beta": 0, "equidistant_beta": 1, "equidistant_temperature": 2, "chebyshev_temperature": 3,
             "equidistant_stepwidth": 4, "chebyshev_stepwidth": 5, "hot_chains": 6}
    lib = _make(str(tmp_path / "lib.so"), ccflags="-DN_BETA=4 -DBETA_ALIGNMENT=" + name, shared=True)
    L = hostlib.load(lib)
    for n_beta in (1, 2, 8, 20, 128):
        for beta_0 in (0.02, 0.4):
            for i in range(n_beta):
                ref = orc.get_chain_beta(order[name], i, n_beta, beta_0)
                assert L.get_chain_beta(i, n_beta, beta_0) == ref, (name, n_beta, i)
                assert abs(sampler.get_chain_beta(kind, i, n_beta, beta_0) - ref) <= 4e-16 * abs(ref)
            cold = L.get_chain_beta(0, n_beta, beta_0)      # chain 0 is the cold chain
            assert abs(cold - (beta_0 if name == "hot_chains" and n_beta > 1 else 1.0)) < 4e-16
    if kind == 0:
        w = wl.pulse(n_data=16, n_chain=3)
        _write_inputs(tmp_path / "w", w)
        m = L.mcmc_load(str(tmp_path / "w" / "params").encode(), str(tmp_path / "w" / "data").encode())
        factors = np.array([1.3, 0.7, 1.0, 2.2, 0.9, 1.1])
        fv = L.gsl_vector_alloc(6)
        hostlib.set_vec(fv, factors)
        lad = orc.Ladder.from_params(w.model, 3, w.start, w.pmin, w.pmax, w.step, w.data)
        ref = orc.lib().orc_calc_beta_0(C.byref(lad.c_state()), 0, factors.ctypes.data_as(C.POINTER(C.c_double)))
        got = L.calc_beta_0(m, fv)
        assert abs(got - ref) <= 1e-15 * ref and 0 < got < 1
        st = LadderState.from_params(3, w.start, w.pmin, w.pmax, w.step)
        assert abs(sampler.calc_beta_0(st, 0, factors) - ref) <= 1e-15 * ref
        L.gsl_vector_free(fv)
        L.mcmc_free(m)


def _rt(a):
    return np.array([float("%.15e" % v) for v in np.ravel(a)]).reshape(np.shape(a))


@pytest.mark.gpu
def test_c_application_workflow_equals_python_mirror_and_oracle(tmp_path):
    from apemost_amd.sampler import HipSampler
    from apemost_amd.state import LadderState
    from oracle import oracle as orc
    n_beta, burn, iters = 8, 1000, 3000
    w = wl.simplesin(n_data=256, n_chain=n_beta)
    work = tmp_path / "w"
    work.mkdir()
    (work / "params").write_text(w.params_file_text())
    (work / "data").write_text(w.data_file_text())
    exe = _make(str(tmp_path / "sine.exe"),
                ccflags="-DN_BETA=%d -DBURN_IN_ITERATIONS=%d -DMAX_ITERATIONS=%d" % (n_beta, burn, iters))
    env = dict(os.environ, APEMOST_SEED="5")
    for phase in ("calibrate_first", "calibrate_rest", "run"):
        subprocess.check_call([exe, phase], cwd=str(work), env=env, stdout=subprocess.DEVNULL)
    out = subprocess.check_output([exe, "analyse"], cwd=str(work), env=env).decode()
    c_calib = (work / "calibration_results").read_text()
    c_accept = (work / "acceptance_rate.dump").read_text().strip().splitlines()[-1].split()
    c_amp = np.loadtxt(str(work / "amplitude-chain-0.prob.dump"))
    c_prob = np.loadtxt(str(work / "prob-chain3.dump"))
    assert len(c_amp) == iters and c_prob.shape == (iters, 2)

    # the same three phases through the Python mirror (each phase = fresh process: ticks restart)
    data = np.loadtxt(str(work / "data"))
    start, pmin, pmax, step = w.start, w.pmin, w.pmax, w.step
    dcfg = capi.calib_defaults(burn_in_iterations=burn)
    mk = lambda: LadderState.from_params(n_beta, _rt(start), _rt(pmin), _rt(pmax), _rt(step))
    s = HipSampler(w.model, 4, 1, data, seed=5)
    st = mk().slice(0, 1)
    s.set_state(st)
    assert s.calibrate_first(dcfg) == 0
    first = s.get_state()
    s.close()
    s = HipSampler(w.model, 4, n_beta, data, seed=5)
    st = mk()
    st.beta[0], st.step[0], st.params[0] = _rt(first.beta[0]), _rt(first.step[0]), _rt(first.params[0])
    st.params_best[0] = st.params[0]
    s.set_state(st)
    status, beta_0, _ = s.calibrate_rest(dcfg)
    assert status == 0
    rest = s.get_state()
    s.close()
    assert rest.calibration_results_text() == c_calib          # byte-identical text

    import torch
    s = HipSampler(w.model, 4, n_beta, data, seed=5)
    st = mk()
    st.read_calibration_results(c_calib)
    s.set_state(st)
    n_swap = 2000 // n_beta
    d = torch.zeros((iters, n_beta, 6), dtype=torch.float64, device="cuda")
    s.run_sampler(iters // n_swap, n_swap, d.data_ptr())
    s.synchronize()
    run = s.get_state()
    s.close()
    assert [int(t) for t in c_accept] == [iters] + [int(a) for a in run.accept]
    samples = d.cpu().numpy()
    assert np.array_equal(c_amp, _rt(samples[:, 0, 0]))         # "%.15e" text of the same doubles
    np.testing.assert_allclose(c_prob[:, 0], samples[:, 3, 4], rtol=2e-6)   # "%6e" keeps 7 digits

    # analyse phase (host post-processing of the dump files): histogram of chain 0 integrates to the
    # bin width (the reference's scaling), evidence = rectangle rule over beta of <prob-prior>/beta
    h = np.loadtxt(str(work / "amplitude.histogram"))
    assert h.shape == (200, 3) and abs(h[:, 2].sum() - (w.pmax[0] - w.pmin[0]) / 200) < 1e-12
    hist, _ = np.histogram(samples[:, 0, 0], bins=200, range=(w.pmin[0], w.pmax[0]))
    np.testing.assert_allclose(h[:, 2], hist * ((w.pmax[0] - w.pmin[0]) / 200 / iters), atol=1e-12)
    betas = np.array([float(l.split()[0]) for l in c_calib.strip().splitlines()])
    means = np.array([np.loadtxt(str(work / ("prob-chain%d.dump" % i)))[:, 1].mean() / betas[i] for i in range(n_beta)])
    evidence, prev = 0.0, 0.0
    for j in range(n_beta - 1, -1, -1):
        evidence += means[j] * (betas[j] - prev)
        prev = betas[j]
    got = float(re.search(r"\] (-?[0-9.]+)\n", out).group(1))
    assert abs(got - evidence) < 1e-4 * abs(evidence) + 1e-5
    assert os.path.exists(str(work / "marginal_distributions.gnuplot")) and "mcmc error estimate of amplitude" in out

    # and the oracle agrees with the run phase (tolerance: DESIGN.md 7)
    lad = orc.Ladder(w.model, n_beta, 4, data)
    for n in ("params", "params_best", "step", "pmin", "pmax", "beta"):
        getattr(lad, n)[...] = getattr(st, n)
    rng = orc.Rng(orc.RNG_STREAMS, 5, lad)
    ref = orc.run_sampler(lad, rng, iters // n_swap, n_swap, record=True)
    np.testing.assert_allclose(samples, ref, rtol=1e-9)
    assert np.array_equal(run.accept, lad.accept)
