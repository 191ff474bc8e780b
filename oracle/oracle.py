"""ctypes front-end of the CPU oracle (oracle/apemost_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (apemost_amd) never imports
this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libapemost_oracle.so")

MODEL_SIMPLESIN, MODEL_PULSE, MODEL_PULSE_VROT, MODEL_SINE3 = 0, 1, 2, 3
MODEL_SINE2, MODEL_NORMAL, MODEL_BERNOULLI = 4, 5, 6   # apps/simplesin2.c, normal.c, bernoulli_example.c
RNG_GLOBAL_MT, RNG_STREAMS = 0, 1
PROPOSAL_GAUSSIAN, PROPOSAL_LOGISTIC, PROPOSAL_UNIFORM = 0, 1, 2
(LADDER_CHEBYSHEV_BETA, LADDER_EQUIDISTANT_BETA, LADDER_EQUIDISTANT_TEMPERATURE,
 LADDER_CHEBYSHEV_TEMPERATURE, LADDER_EQUIDISTANT_STEPWIDTH, LADDER_CHEBYSHEV_STEPWIDTH,
 LADDER_HOT_CHAINS) = range(7)
CALIB_OK, CALIB_STEP_TOO_LARGE, CALIB_ITER_LIMIT = 0, 1, 2

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint64)


class _MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int)]


class _Rng(C.Structure):
    _fields_ = [("kind", C.c_int), ("mt", _MT), ("seed", C.c_uint64), ("ticks", _up),
                ("round", C.c_uint64), ("draws", C.c_uint64)]


class _State(C.Structure):
    _fields_ = [("n_chain", C.c_int), ("n_par", C.c_int), ("model", C.c_int),
                ("n_data", C.c_int), ("n_cols", C.c_int), ("chain_offset", C.c_int64),
                ("data", _dp), ("sigma", C.c_double), ("hmin", C.c_double), ("circular", C.c_uint64),
                ("params", _dp), ("params_best", _dp), ("step", _dp), ("pmin", _dp),
                ("pmax", _dp), ("params_accepts", _up), ("params_rejects", _up),
                ("beta", _dp), ("prob", _dp), ("prior", _dp), ("prob_best", _dp),
                ("accept", _up), ("reject", _up), ("n_iter", _up), ("swapcount", _up),
                ("proposal", C.c_int), ("randomswap", C.c_int), ("adapt", C.c_int),
                ("adapt_target", C.c_double), ("rwm", C.c_int)]


class CalibCfg(C.Structure):
    _fields_ = [("burn_in_iterations", C.c_uint), ("rat_limit", C.c_double),
                ("target_global", C.c_double), ("max_ar_deviation", C.c_double),
                ("iter_limit", C.c_uint), ("mul", C.c_double), ("adjust_step", C.c_double),
                ("iter_readjust", C.c_uint), ("no_rescaling_limit", C.c_int)]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "apemost_oracle.c")
    hdr = os.path.join(_HERE, "apemost_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libapemost_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_mt_seed.argtypes = [C.POINTER(_MT), C.c_ulong]
        L.orc_mt_next.argtypes = [C.POINTER(_MT)]
        L.orc_mt_next.restype = C.c_uint32
        L.orc_philox_at.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_philox_at.restype = C.c_uint32
        L.orc_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32)]
        L.orc_uniform.argtypes = [C.POINTER(_Rng)]
        L.orc_uniform.restype = C.c_double
        L.orc_gaussian.argtypes = [C.POINTER(_Rng), C.c_double]
        L.orc_gaussian.restype = C.c_double
        L.orc_gaussian_attempt.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, _dp, _dp]
        L.orc_gaussian_attempt.restype = C.c_int
        L.orc_jump.argtypes = [C.POINTER(_Rng), C.c_double, C.c_int]
        L.orc_jump.restype = C.c_double
        L.orc_jump_attempt.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                       C.c_double, _dp]
        L.orc_jump_attempt.restype = C.c_int
        L.orc_adapt.argtypes = [C.POINTER(_State), C.c_int]
        L.orc_rwm.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_int]
        L.orc_rwm_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_int]
        L.orc_rwm_uniform.restype = C.c_double
        L.orc_accept_log_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint64]
        L.orc_accept_log_uniform.restype = C.c_double
        L.orc_loglike.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double,
                                  C.c_double, C.c_double, _dp]
        L.orc_loglike.restype = C.c_double
        L.orc_calc_model.argtypes = [C.POINTER(_State), C.c_int]
        L.orc_check_accept.argtypes = [C.c_double, C.c_double, C.POINTER(_Rng), C.POINTER(_State),
                                       C.c_int, C.POINTER(C.c_int)]
        L.orc_check_accept.restype = C.c_int
        for name in ("orc_markov_chain_step",):
            getattr(L, name).argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_int]
        L.orc_markov_chain_step_for.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_int, C.c_int]
        L.orc_check_best.argtypes = [C.POINTER(_State), C.c_int]
        L.orc_restart_from_best.argtypes = [C.POINTER(_State), C.c_int]
        L.orc_reset_accept_rejects.argtypes = [C.POINTER(_State), C.c_int]
        L.orc_ladder_beta.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_double]
        L.orc_ladder_beta.restype = C.c_double
        L.orc_get_chain_beta.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_double]
        L.orc_get_chain_beta.restype = C.c_double
        L.orc_calc_beta_0.argtypes = [C.POINTER(_State), C.c_int, _dp]
        L.orc_calc_beta_0.restype = C.c_double
        L.orc_tempering_interaction.argtypes = [C.POINTER(_State), C.POINTER(_Rng), _dp]
        L.orc_tempering_interaction.restype = C.c_int
        L.orc_swap_decision.argtypes = [C.c_double] * 5 + [_dp]
        L.orc_swap_decision.restype = C.c_int
        L.orc_swap_pair_index.argtypes = [C.c_double, C.c_int]
        L.orc_swap_pair_index.restype = C.c_int
        L.orc_run_sampler.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_uint64, C.c_uint,
                                      _dp, C.c_int]
        L.orc_run_steps.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_uint, _dp, C.c_int]
        L.orc_tempering_interaction_shard.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_int64, _dp, _dp,
                                                      C.POINTER(C.c_int)]
        L.orc_tempering_interaction_shard.restype = C.c_int
        L.orc_burn_in.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_int, C.c_uint]
        L.orc_calib_defaults.argtypes = [C.POINTER(CalibCfg)]
        for name in ("orc_calibrate_orig", "orc_markov_chain_calibrate"):
            f = getattr(L, name)
            f.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.c_int, C.POINTER(CalibCfg), _up]
            f.restype = C.c_int
        L.orc_set_progress_path.argtypes = [C.c_char_p]
        L.orc_set_progress_path.restype = None
        L.orc_calibrate_first.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.POINTER(CalibCfg)]
        L.orc_calibrate_first.restype = C.c_int
        L.orc_calibrate_rest.argtypes = [C.POINTER(_State), C.POINTER(_Rng), C.POINTER(CalibCfg),
                                         C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp]
        L.orc_calibrate_rest.restype = C.c_int
        L.orc_mod_double.argtypes = [C.c_double, C.c_double]
        L.orc_mod_double.restype = C.c_double
        _lib = L
    return _lib


def calib_defaults(**overrides):
    cfg = CalibCfg()
    lib().orc_calib_defaults(C.byref(cfg))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


_F64 = ("params", "params_best", "step", "pmin", "pmax", "beta", "prob", "prior", "prob_best")
_U64 = ("params_accepts", "params_rejects", "accept", "reject", "n_iter", "swapcount")


class Ladder:
    """A ladder of chains in the oracle's structure-of-arrays layout (numpy owned).

    Field names and shapes are identical to the device engine's host mirror, so
    tests compare array by array.
    """

    def __init__(self, model, n_chain, n_par, data, chain_offset=0, sigma=0.5, hmin=1e-6):
        self.model, self.n_chain, self.n_par = int(model), int(n_chain), int(n_par)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        assert self.data.ndim == 2
        self.chain_offset = int(chain_offset)
        self.sigma, self.hmin = float(sigma), float(hmin)
        self.circular = 0   # bit p: parameter p wraps (CIRCULAR_PARAMS)
        # the reference's compile-time variants (0 = default build)
        self.proposal = PROPOSAL_GAUSSIAN   # -DPROPOSAL_LOGISTIC / -DPROPOSAL_UNIFORM
        self.randomswap = 0                 # -DRANDOMSWAP
        self.adapt = 0                      # -DADAPT
        self.adapt_target = 0.5             # TARGET_ACCEPTANCE_RATE
        self.rwm = 0                        # -DRWM
        z2 = lambda dt: np.zeros((n_chain, n_par), dtype=dt)
        z1 = lambda dt: np.zeros((n_chain,), dtype=dt)
        self.params, self.params_best, self.step = z2(np.float64), z2(np.float64), z2(np.float64)
        self.pmin, self.pmax = z2(np.float64), z2(np.float64)
        self.params_accepts, self.params_rejects = z2(np.uint64), z2(np.uint64)
        self.beta = np.ones((n_chain,), dtype=np.float64)      # setup_chains: beta := 1
        self.prob = np.full((n_chain,), -1e10)                  # mcmc_init, src/mcmc.c:47
        self.prior = z1(np.float64)
        self.prob_best = np.full((n_chain,), -1e10)             # src/mcmc.c:49
        self.accept, self.reject = z1(np.uint64), z1(np.uint64)
        self.n_iter, self.swapcount = z1(np.uint64), z1(np.uint64)

    @classmethod
    def from_params(cls, model, n_chain, start, pmin, pmax, step, data, **kw):
        """What setup_chains() leaves (src/parallel_tempering_config.c:95-123): every chain a
        copy of the params file; step<0 means 10 % of the range (src/mcmc_parser.c:84-87)."""
        start, pmin, pmax, step = (np.asarray(a, dtype=np.float64) for a in (start, pmin, pmax, step))
        lad = cls(model, n_chain, len(start), data, **kw)
        step = np.where(step < 0, (pmax - pmin) * 0.1, step)
        lad.params[:] = start
        lad.params_best[:] = start
        lad.pmin[:], lad.pmax[:], lad.step[:] = pmin, pmax, step
        return lad

    def copy_from(self, other):
        for n in _F64 + _U64:
            getattr(self, n)[...] = getattr(other, n)
        return self

    def c_state(self):
        st = _State()
        st.n_chain, st.n_par, st.model = self.n_chain, self.n_par, self.model
        st.n_data, st.n_cols = self.data.shape
        st.chain_offset = self.chain_offset
        st.data = self.data.ctypes.data_as(_dp)
        st.sigma, st.hmin = self.sigma, self.hmin
        st.circular = self.circular
        st.proposal, st.randomswap, st.adapt = int(self.proposal), int(self.randomswap), int(self.adapt)
        st.adapt_target = float(self.adapt_target)
        st.rwm = int(self.rwm)
        for n in _F64:
            a = getattr(self, n)
            assert a.flags.c_contiguous and a.dtype == np.float64, n
            setattr(st, n, a.ctypes.data_as(_dp))
        for n in _U64:
            a = getattr(self, n)
            assert a.flags.c_contiguous and a.dtype == np.uint64, n
            setattr(st, n, a.ctypes.data_as(_up))
        return st


class Rng:
    def __init__(self, kind, seed=0, ladder=None):
        self.c = _Rng()
        self.c.kind = kind
        self.ticks = None
        if kind == RNG_GLOBAL_MT:
            lib().orc_mt_seed(C.byref(self.c.mt), seed)
        else:
            assert ladder is not None
            self.c.seed = seed
            self.ticks = np.zeros(ladder.n_chain, dtype=np.uint64)
            self.c.ticks = self.ticks.ctypes.data_as(_up)

    @property
    def round(self):
        return int(self.c.round)

    @round.setter
    def round(self, v):
        self.c.round = v


def mt_stream(seed, n):
    g = _MT()
    L = lib()
    L.orc_mt_seed(C.byref(g), seed)
    return np.array([L.orc_mt_next(C.byref(g)) for _ in range(n)], dtype=np.uint32)


def philox_block(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def philox_stream(seed, subsequence, n, start=0):
    L = lib()
    return np.array([L.orc_philox_at(seed, subsequence, start + i) for i in range(n)], dtype=np.uint32)


def gaussian_attempt(seed, chain_global, slot, tick, q):
    y, sq = C.c_double(0), C.c_double(0)
    ok = lib().orc_gaussian_attempt(seed, chain_global, slot, tick, q, C.byref(y), C.byref(sq))
    return bool(ok), y.value, sq.value


def jump_attempt(seed, chain_global, slot, tick, q, proposal, sigma):
    j = C.c_double(0)
    ok = lib().orc_jump_attempt(seed, chain_global, slot, tick, q, proposal, sigma, C.byref(j))
    return bool(ok), j.value


def accept_log_uniform(seed, chain_global, n_par, tick):
    return lib().orc_accept_log_uniform(seed, chain_global, n_par, tick)


def loglike(model, params, data, beta=1.0, sigma=0.5, hmin=1e-6):
    params = np.ascontiguousarray(params, dtype=np.float64)
    data = np.ascontiguousarray(data, dtype=np.float64)
    prior = C.c_double(0)
    prob = lib().orc_loglike(model, len(params), params.ctypes.data_as(_dp), data.ctypes.data_as(_dp),
                             data.shape[0], data.shape[1], beta, sigma, hmin, C.byref(prior))
    return prob, prior.value


def run_sampler(ladder, rng, n_rounds, n_swap, record=False, n_threads=1):
    st = ladder.c_state()
    samples = None
    ptr = None
    if record:
        samples = np.zeros((n_rounds * n_swap, ladder.n_chain, ladder.n_par + 2))
        ptr = samples.ctypes.data_as(_dp)
    lib().orc_run_sampler(C.byref(st), C.byref(rng.c), n_rounds, n_swap, ptr, n_threads)
    return samples


def run_steps(ladder, rng, n_steps, record=False, n_threads=1):
    st = ladder.c_state()
    samples, ptr = None, None
    if record:
        samples = np.zeros((n_steps, ladder.n_chain, ladder.n_par + 2))
        ptr = samples.ctypes.data_as(_dp)
    lib().orc_run_steps(C.byref(st), C.byref(rng.c), n_steps, ptr, n_threads)
    return samples


def tempering_interaction_shard(ladder, rng, n_global, halo_lo=None, halo_hi=None):
    st = ladder.c_state()
    sw = C.c_int(0)
    as_p = lambda h: None if h is None else np.ascontiguousarray(h, dtype=np.float64).ctypes.data_as(_dp)
    keep = [None if h is None else np.ascontiguousarray(h, dtype=np.float64) for h in (halo_lo, halo_hi)]
    ptrs = [None if k is None else k.ctypes.data_as(_dp) for k in keep]
    a = lib().orc_tempering_interaction_shard(C.byref(st), C.byref(rng.c), n_global, ptrs[0], ptrs[1], C.byref(sw))
    return a, bool(sw.value)


def step(ladder, rng, chain):
    st = ladder.c_state()
    lib().orc_markov_chain_step(C.byref(st), C.byref(rng.c), chain)


def step_for(ladder, rng, chain, p):
    st = ladder.c_state()
    lib().orc_markov_chain_step_for(C.byref(st), C.byref(rng.c), chain, p)


def rwm(ladder, rng, chain):
    st = ladder.c_state()
    lib().orc_rwm(C.byref(st), C.byref(rng.c), chain)


def check_best(ladder, chain):
    st = ladder.c_state()
    lib().orc_check_best(C.byref(st), chain)


def calc_model(ladder, chain):
    st = ladder.c_state()
    lib().orc_calc_model(C.byref(st), chain)


def tempering_interaction(ladder, rng):
    st = ladder.c_state()
    trace = np.zeros(3)
    a = lib().orc_tempering_interaction(C.byref(st), C.byref(rng.c), trace.ctypes.data_as(_dp))
    return a, trace


def burn_in(ladder, rng, chain, iterations):
    st = ladder.c_state()
    lib().orc_burn_in(C.byref(st), C.byref(rng.c), chain, iterations)


def markov_chain_calibrate(ladder, rng, chain, cfg):
    st = ladder.c_state()
    iters = C.c_uint64(0)
    status = lib().orc_markov_chain_calibrate(C.byref(st), C.byref(rng.c), chain, C.byref(cfg),
                                              C.byref(iters))
    return status, iters.value


_progress_path = None


def set_progress_path(path):
    """calibration_progress.data of the calibrations that follow (None: none); every chain's
    calibration truncates it, as in the reference"""
    global _progress_path
    _progress_path = None if path is None else os.fsencode(str(path))   # kept alive: the C side holds the pointer
    lib().orc_set_progress_path(_progress_path)


def calibrate_first(ladder, rng, cfg):
    st = ladder.c_state()
    return lib().orc_calibrate_first(C.byref(st), C.byref(rng.c), C.byref(cfg))


def calibrate_rest(ladder, rng, cfg, ladder_kind=LADDER_CHEBYSHEV_BETA, beta_0=-0.001,
                   skip_calibrate_allchains=False, n_threads=1):
    st = ladder.c_state()
    b0 = C.c_double(0)
    factors = np.zeros(ladder.n_par)
    status = lib().orc_calibrate_rest(C.byref(st), C.byref(rng.c), C.byref(cfg), ladder_kind, beta_0,
                                      int(skip_calibrate_allchains), n_threads, C.byref(b0),
                                      factors.ctypes.data_as(_dp))
    return status, b0.value, factors


def get_chain_beta(kind, i, n_beta, beta_0):
    return lib().orc_get_chain_beta(kind, i, n_beta, beta_0)
