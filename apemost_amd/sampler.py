"""HipSampler: host-side mirror of the reference's parallel-tempering operators for one
device shard, implemented entirely by calls into libapemost_hip.so (include/apemost_hip.h).

Method names follow the reference functions they replace:
  run_sampler            src/parallel_tempering.c:347-419
  tempering_interaction  src/parallel_tempering_interaction.c:125-141 (fused into the next round)
  calc_model             apps/<model>.c (device re-implementation)
  markov_chain_calibrate src/markov_chain_calibrate.c:1182-1204
  calibrate_first/rest   src/parallel_tempering.c:78-207
"""
import ctypes as C

import numpy as np

from . import capi
from .state import LadderState

LADDER_CHEBYSHEV_BETA = 0


def ladder_beta(kind, i, n_beta, beta_0):
    """BETA_ALIGNMENT functions, src/parallel_tempering_beta.c:53-83 (host scalar math)."""
    import math
    d = n_beta - 1
    if kind == 0:
        return beta_0 + (1 - beta_0) / 2 * (1 - math.cos(i * math.pi / d))
    if kind == 1:
        return beta_0 + i * (1 - beta_0) / d
    if kind == 2:
        return 1 / (1 / beta_0 + i * (1 - 1 / beta_0) / d)
    if kind == 3:
        return 1 / (1 / beta_0 + (1 - 1 / beta_0) / 2 * (1 - math.cos(i * math.pi / d)))
    if kind == 4:
        return beta_0 + math.pow(i * 1.0 / d, 2) * (1 - beta_0)
    if kind == 5:
        return beta_0 + (1 - beta_0) * math.pow((1 - math.cos(i * math.pi / d)) / 2, 2)
    if kind == 6:
        return beta_0
    raise ValueError("unknown ladder kind %r" % kind)


def get_chain_beta(kind, i, n_beta, beta_0):
    """get_chain_beta, src/parallel_tempering_beta.c:85-90: chain 0 is beta = 1."""
    if n_beta == 1:
        return 1.0
    return ladder_beta(kind, n_beta - i - 1, n_beta, beta_0)


def calc_beta_0(state, chain, stepwidth_factors):
    """calc_beta_0, src/parallel_tempering_beta.c:92-102 (BETA_0_STEPWIDTH = 1.0)."""
    r = (state.pmax[chain] - state.pmin[chain]) * 1.0
    r = r / state.step[chain]
    r = r / np.asarray(stepwidth_factors)
    return float(np.max(r)) ** -0.5


class HipSampler:
    def __init__(self, model, n_par, n_chains, data, seed=0, device=0, chain_offset=0,
                 n_chains_global=None, waves_per_chain=0, sigma=0.5, hmin=1e-6, lds_policy=0, circular_params=0,
                 flags=0, adapt_target=0.0, device_model_source=None):
        data = np.ascontiguousarray(data, dtype=np.float64)
        assert data.ndim == 2
        self.cfg = capi.Config(abi_version=capi.ABI_VERSION, device=device, model=model, n_par=n_par,
                               n_chains=n_chains, n_data=data.shape[0], n_cols=data.shape[1],
                               waves_per_chain=waves_per_chain, lds_policy=lds_policy, chain_offset=chain_offset,
                               n_chains_global=n_chains if n_chains_global is None else n_chains_global,
                               seed=seed, sigma=sigma, hmin=hmin, circular_params=circular_params, flags=flags,
                               adapt_target=adapt_target,
                               device_model_source=None if device_model_source is None else str(device_model_source).encode())
        self._h = C.c_void_p()
        self.L = capi.lib()
        capi.check(self.L.apemost_hip_create(C.byref(self.cfg), C.byref(self._h)))
        capi.check(self.L.apemost_hip_set_data(self._h, data.ctypes.data_as(C.POINTER(C.c_double))))
        self.n_par, self.n_chains = n_par, n_chains
        self.n_chains_global = self.cfg.n_chains_global
        self.chain_offset = chain_offset
        self.seed = seed

    def close(self):
        if self._h:
            self.L.apemost_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -----------------------------------------------------------------
    def synchronize(self):
        capi.check(self.L.apemost_hip_synchronize(self._h))

    @property
    def stream(self):
        p = C.c_void_p()
        capi.check(self.L.apemost_hip_stream(self._h, C.byref(p)))
        return p.value or 0

    @property
    def geometry(self):
        w, l = C.c_int(0), C.c_int(0)
        capi.check(self.L.apemost_hip_waves_per_chain(self._h, C.byref(w), C.byref(l)))
        return w.value, bool(l.value)

    @property
    def user_model_compile_seconds(self):
        """hiprtc's time for this sampler's device model (0: the process had compiled the same source before)"""
        t = C.c_double(0)
        capi.check(self.L.apemost_hip_user_model_compile_seconds(self._h, C.byref(t)))
        return t.value

    @property
    def launch_policy(self):
        """(one-barrier kernel, cooperative multi-round launches, rounds one launch may hold)"""
        ob, co, mr = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        capi.check(self.L.apemost_hip_launch_policy(self._h, C.byref(ob), C.byref(co), C.byref(mr)))
        return bool(ob.value), bool(co.value), mr.value

    @property
    def ob_helper(self):
        """the one-barrier kernel runs with a helper wavefront (the prior of the proposal in flight off the owner)"""
        ob = C.c_int32(0)
        capi.check(self.L.apemost_hip_launch_policy(self._h, C.byref(ob), None, None))
        return ob.value == 2

    def set_state(self, state, fields=None):
        v = state.view(fields) if fields else state.view()
        capi.check(self.L.apemost_hip_set_state(self._h, C.byref(v)))

    def get_state(self, state=None, fields=None):
        state = state or LadderState(self.n_chains, self.n_par)
        v = state.view(fields) if fields else state.view()
        capi.check(self.L.apemost_hip_get_state(self._h, C.byref(v)))
        return state

    @property
    def round(self):
        r, p = C.c_uint64(0), C.c_int(0)
        capi.check(self.L.apemost_hip_get_round(self._h, C.byref(r), C.byref(p)))
        return r.value, bool(p.value)

    def set_round(self, round_, swap_pending=False):
        capi.check(self.L.apemost_hip_set_round(self._h, round_, int(swap_pending)))

    # -- hot path -----------------------------------------------------------------
    def calc_model(self, first=0, count=-1):
        capi.check(self.L.apemost_hip_calc_model(self._h, first, count))

    def loglike(self, params, beta):
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(-1, self.n_par)
        n = params.shape[0]
        beta = np.ascontiguousarray(np.broadcast_to(np.asarray(beta, dtype=np.float64), (n,)))
        prob, prior = np.zeros(n), np.zeros(n)
        dp = C.POINTER(C.c_double)
        capi.check(self.L.apemost_hip_loglike(self._h, n, params.ctypes.data_as(dp), beta.ctypes.data_as(dp),
                                              prob.ctypes.data_as(dp), prior.ctypes.data_as(dp)))
        return prob, prior

    def _adopt(self, d_samples):
        """The sampler launches on its own stream.  A sample buffer that torch has just created
        (torch.zeros fills it on torch's current stream) must be complete before the first kernel
        writes rows into it, or the fill may land on top of them."""
        if d_samples and d_samples != getattr(self, "_last_samples", 0):
            import sys
            torch = sys.modules.get("torch")    # (a process without torch has no torch buffer to wait for)
            if torch is not None and torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()
        self._last_samples = d_samples

    def launch_round(self, n_steps, apply_swap, d_samples=0):
        self._adopt(d_samples)
        capi.check(self.L.apemost_hip_launch_round(self._h, n_steps, int(apply_swap), d_samples))

    def launch_rounds(self, n_rounds, n_steps, apply_swap, d_samples=0):
        self._adopt(d_samples)
        capi.check(self.L.apemost_hip_launch_rounds(self._h, n_rounds, n_steps, int(apply_swap), d_samples))

    def swap_pair(self, round_):
        """lower chain of the pair swap attempt `round_` picks under this sampler's swap schedule"""
        return int(self.L.apemost_hip_sampler_swap_pair(self._h, round_))

    def rounds_within_shard(self, first_round, max_rounds):
        """how many swap attempts from `first_round` on (at most max_rounds) do not straddle an edge of this shard"""
        return int(self.L.apemost_hip_rounds_within_shard(self._h, first_round, max_rounds))

    @property
    def max_rounds_per_launch(self):
        v = C.c_int32(0)
        capi.check(self.L.apemost_hip_max_rounds_per_launch(self._h, C.byref(v)))
        return v.value

    def markov_chain_step_for(self, param, n_steps=1, d_samples=0):
        self._adopt(d_samples)
        capi.check(self.L.apemost_hip_launch_round_for(self._h, n_steps, param, d_samples))

    def run_sampler(self, n_rounds, n_swap, d_samples=0):
        """n_rounds x {n_swap steps per chain, one swap attempt}; asynchronous."""
        self._adopt(d_samples)
        capi.check(self.L.apemost_hip_run(self._h, n_rounds, n_swap, d_samples))

    def edge_export(self, side, d_buf):
        capi.check(self.L.apemost_hip_edge_export(self._h, side, d_buf))

    def edge_import(self, side, d_buf):
        capi.check(self.L.apemost_hip_edge_import(self._h, side, d_buf))

    # -- calibration ----------------------------------------------------------------
    def markov_chain_calibrate(self, first, count, cfg=None, burn_in_only=False):
        """markov_chain_calibrate() for chains [first, first+count).  Unless cfg names another chain
        (progress_chain >= 0), the readjustments of the LAST chain of the range are logged: the
        reference reopens calibration_progress.data "w" for every chain it calibrates
        (src/markov_chain_calibrate.c:1052), so that chain's lines are the ones a single-threaded
        run leaves behind (calibration_progress_text)."""
        src = cfg or capi.calib_defaults()
        cfg = capi.CalibConfig()
        C.pointer(cfg)[0] = src
        if cfg.progress_chain < 0 and not burn_in_only:
            cfg.progress_chain = first + count - 1
        status = np.zeros(count, dtype=np.int32)
        iters = np.zeros(count, dtype=np.uint64)
        rc = self.L.apemost_hip_calibrate_chains(self._h, first, count, C.byref(cfg), int(burn_in_only),
                                                 status.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 iters.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc not in (capi.OK, capi.ERR_CALIBRATION):
            capi.check(rc)
        if not burn_in_only:
            self._progress_text = self._format_progress(self.calibrate_progress())
        return status, iters

    def _format_progress(self, rows):
        """the line format of src/markov_chain_calibrate.c:1143-1146: "%d\t%lu\t%f\t%f\t%f\n" of
        (parameter, iter, normalised step, accept rate, -1.)"""
        return "".join("%d\t%d\t%f\t%f\t%f\n" % (i, int(r[0]), r[1 + 2 * i], r[2 + 2 * i], -1.0)
                       for r in rows for i in range(self.n_par))

    def calibration_progress_text(self):
        """contents of calibration_progress.data after the calibrations made so far"""
        return getattr(self, "_progress_text", "")

    def calibrate_progress(self):
        """rows (iter, then (normalised step, accept rate) per parameter) of the chain named by
        cfg.progress_chain in the latest calibration: what the reference writes to
        calibration_progress.data (src/markov_chain_calibrate.c:1143-1146)"""
        n = C.c_int32(0)
        capi.check(self.L.apemost_hip_calibrate_progress(self._h, None, 0, C.byref(n)))
        rows = np.zeros((n.value, 1 + 2 * self.n_par))
        if n.value:
            capi.check(self.L.apemost_hip_calibrate_progress(self._h, rows.ctypes.data_as(C.POINTER(C.c_double)),
                                                             n.value, C.byref(n)))
        return rows

    def calibrate_stats(self):
        """(segments, likelihood evaluations, launches per waves-per-chain) of the latest calibration;
        self.calibrate_seconds_by_waves holds the wall seconds per waves-per-chain"""
        seg, ev = C.c_uint64(0), C.c_uint64(0)
        by = (C.c_uint64 * 9)()
        sec = (C.c_double * 9)()
        capi.check(self.L.apemost_hip_calibrate_stats(self._h, C.byref(seg), C.byref(ev), by, sec))
        self.calibrate_seconds_by_waves = {w: float(sec[w]) for w in range(9) if by[w]}
        return seg.value, ev.value, {w: int(by[w]) for w in range(9) if by[w]}

    def calibrate_first(self, cfg=None):
        """calibrate_first(): calc_model(chain 0) then markov_chain_calibrate(chain 0)."""
        self.calc_model(0, 1)
        status, _ = self.markov_chain_calibrate(0, 1, cfg)
        return int(status[0])

    def calibrate_rest(self, cfg=None, ladder_kind=LADDER_CHEBYSHEV_BETA, beta_0=-0.001,
                       skip_calibrate_allchains=False):
        """calibrate_rest() for a whole ladder on this device.  Entry state: chain 0 carries the
        calibrated steps/params (read_calibration_file(chains, 1)), every beta = 1."""
        assert self.n_chains == self.n_chains_global and self.chain_offset == 0
        from .distributed import calibrate_rest_sharded
        return calibrate_rest_sharded(self, self.n_chains, 0, cfg, ladder_kind, beta_0, skip_calibrate_allchains)
