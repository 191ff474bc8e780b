"""Whole-workflow check of the CPU oracle in the reference's own RNG mode.

SURVEY.md 8(c) G5 records the last `acceptance_rate.dump` row that the compiled
reference printed for BASELINE config 1 (N_BETA=8, 256 points, BURN_IN_ITERATIONS=2000,
MAX_ITERATIONS=100000, GSL_RNG_SEED=0, OMP_NUM_THREADS=1; libm-sin GSL stand-in):
    100000 53535 53609 53084 50156 51946 55040 49129 69024
Reproducing all eight counters after 800 000 Metropolis steps exercises calibrate_first,
calibrate_rest, the "%.15e" calibration_results round trip, run_sampler and the swap
in exactly the reference's draw order.
"""
import numpy as np

from apemost_amd import workloads as wl
from oracle import oracle as orc


def _rt(a):
    """the "%.15e" text round trip of calibration_results (parallel_tempering_config.c:176-202)"""
    return np.array([float("%.15e" % v) for v in np.ravel(a)]).reshape(np.shape(a))


def test_config1_three_phase_workflow_matches_survey_probe():
    w = wl.simplesin(n_data=256, n_chain=8)
    mk = lambda: orc.Ladder.from_params(w.model, 8, w.start, w.pmin, w.pmax, [-1] * 4, w.data)
    cfg = orc.calib_defaults(burn_in_iterations=2000)

    first = mk()
    assert orc.calibrate_first(first, orc.Rng(orc.RNG_GLOBAL_MT, 0), cfg) == orc.CALIB_OK

    rest = mk()   # new process: setup_chains + read_calibration_file(chains, 1)
    rest.beta[0], rest.step[0], rest.params[0] = _rt(first.beta[0]), _rt(first.step[0]), _rt(first.params[0])
    rest.params_best[0] = rest.params[0]
    status, beta_0, factors = orc.calibrate_rest(rest, orc.Rng(orc.RNG_GLOBAL_MT, 0), cfg)
    assert status == orc.CALIB_OK
    assert rest.beta[0] == 1.0 and abs(rest.beta[-1] - beta_0) < 1e-15

    run = mk()    # new process: read_calibration_file(chains, n_beta); prob stays -1e10 (quirk Q2)
    run.beta[:], run.step[:], run.params[:] = _rt(rest.beta), _rt(rest.step), _rt(rest.params)
    run.params_best[:] = run.params
    rng = orc.Rng(orc.RNG_GLOBAL_MT, 0)
    orc.run_sampler(run, rng, 400, 250)       # n_swap = 2000/8
    assert list(run.n_iter) == [100000] * 8
    assert list(run.accept) == [53535, 53609, 53084, 50156, 51946, 55040, 49129, 69024]
    assert np.all(run.accept + run.reject == 100000)


def test_streams_mode_is_thread_count_invariant():
    w = wl.simplesin(n_data=64, n_chain=6)
    out = []
    for threads in (1, 4):
        lad = orc.Ladder.from_params(w.model, 6, w.start, w.pmin, w.pmax, w.step, w.data)
        lad.beta[:] = [orc.get_chain_beta(0, i, 6, 0.05) for i in range(6)]
        rng = orc.Rng(orc.RNG_STREAMS, 99, lad)
        s = orc.run_sampler(lad, rng, 20, 7, record=True, n_threads=threads)
        out.append((s, lad.params.copy(), lad.accept.copy(), lad.swapcount.copy(), rng.ticks.copy()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert list(out[0][4]) == [140] * 6     # one tick per Metropolis update
