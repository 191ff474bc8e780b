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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "apemost_amd", "host")
REF = "/root/reference"
STRICT = "-std=c99 -fopenmp -Wall -Werror -Wextra -ansi -pedantic"   # the reference Makefile's CFLAGS


def _make(out, app=None, main=None, ccflags="", strict=STRICT):
    cmd = ["make", "-s", "-C", HOST, "OUT=" + out, "STRICT=" + strict, "CCFLAGS=" + ccflags]
    if app:
        cmd.append("APP=" + app)
    if main:
        cmd.append("MAIN=" + main)
    subprocess.check_call(cmd)
    return out


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


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_apps_link_unchanged_and_eval_matches_manual(tmp_path):
    build.build_hip()
    for app in ("simplesin", "pulse", "pulse_vrot"):
        for main in ("generic_main", "eval_main", "benchmark_main"):
            _make(str(tmp_path / ("%s_%s.exe" % (main, app))), app="%s/apps/%s.c" % (REF, app),
                  main="%s/apps/%s.c" % (REF, main), ccflags="-DN_BETA=8")
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
