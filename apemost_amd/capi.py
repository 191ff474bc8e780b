"""ctypes binding of the C ABI in include/apemost_hip.h (libapemost_hip.so).

This is plumbing only: every call goes straight to the HIP library.  There is no
Python or CPU implementation behind it -- if the library is missing or no gfx950
device is present, calls raise ApemostHipError.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

ABI_VERSION = 3
FLAG_SINGLE_ROUND_LAUNCHES, FLAG_COOPERATIVE_LAUNCH, FLAG_TWO_BARRIER_STEP = 1, 2, 4
# the reference's compile-time variants (-DPROPOSAL_LOGISTIC, -DPROPOSAL_UNIFORM, -DRANDOMSWAP, -DADAPT)
FLAG_PROPOSAL_LOGISTIC, FLAG_PROPOSAL_UNIFORM, FLAG_RANDOMSWAP, FLAG_ADAPT = 8, 16, 32, 64
FLAG_TEST_REFUSE_COOPERATIVE, FLAG_TEST_WITHHOLD_PUBLISH = 128, 256   # test hooks (include/apemost_hip.h)
FLAG_RWM = 512
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_RUNTIME, ERR_UNSUPPORTED, ERR_CALIBRATION = 0, -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint64)


class ApemostHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("apemost_hip error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("model", C.c_int32),
                ("n_par", C.c_int32), ("n_chains", C.c_int32), ("n_data", C.c_int32),
                ("n_cols", C.c_int32), ("waves_per_chain", C.c_int32), ("lds_policy", C.c_int32),
                ("flags", C.c_int32), ("chain_offset", C.c_int64),
                ("n_chains_global", C.c_int64), ("seed", C.c_uint64), ("sigma", C.c_double),
                ("hmin", C.c_double), ("circular_params", C.c_uint64), ("adapt_target", C.c_double),
                ("device_model_source", C.c_char_p)]


class StateView(C.Structure):
    _fields_ = [("params", _dp), ("params_best", _dp), ("step", _dp), ("pmin", _dp), ("pmax", _dp),
                ("params_accepts", _up), ("params_rejects", _up), ("beta", _dp), ("prob", _dp),
                ("prior", _dp), ("prob_best", _dp), ("accept", _up), ("reject", _up), ("n_iter", _up),
                ("swapcount", _up), ("ticks", _up)]


class CalibConfig(C.Structure):
    _fields_ = [("burn_in_iterations", C.c_uint32), ("iter_limit", C.c_uint32),
                ("iter_readjust", C.c_uint32), ("no_rescaling_limit", C.c_int32),
                ("rat_limit", C.c_double), ("target_global", C.c_double),
                ("max_ar_deviation", C.c_double), ("mul", C.c_double), ("adjust_step", C.c_double),
                ("progress_chain", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/apemost_hip.h declares
EXPORTS = [
    "apemost_hip_last_error", "apemost_hip_abi_version", "apemost_hip_device_count",
    "apemost_hip_device_info", "apemost_hip_create", "apemost_hip_destroy", "apemost_hip_synchronize",
    "apemost_hip_stream", "apemost_hip_waves_per_chain", "apemost_hip_launch_policy", "apemost_hip_user_model_compile_seconds", "apemost_hip_set_chain_offset", "apemost_hip_set_data", "apemost_hip_set_state",
    "apemost_hip_get_state", "apemost_hip_set_round", "apemost_hip_get_round", "apemost_hip_calc_model",
    "apemost_hip_loglike", "apemost_hip_launch_round", "apemost_hip_launch_rounds", "apemost_hip_max_rounds_per_launch",
    "apemost_hip_launch_round_for", "apemost_hip_run", "apemost_hip_samples_alloc",
    "apemost_hip_samples_read", "apemost_hip_samples_free", "apemost_hip_samples_read_async",
    "apemost_hip_samples_wait", "apemost_hip_samples_pack_read_async", "apemost_hip_host_alloc", "apemost_hip_host_free", "apemost_hip_swap_pair",
    "apemost_hip_sampler_swap_pair", "apemost_hip_rounds_within_shard",
    "apemost_hip_edge_doubles", "apemost_hip_edge_export", "apemost_hip_edge_import",
    "apemost_hip_edge_exchange", "apemost_hip_run_shards",
    "apemost_hip_calib_defaults", "apemost_hip_calibrate_chains", "apemost_hip_calibrate_begin",
    "apemost_hip_calibrate_end", "apemost_hip_calibrate_poll", "apemost_hip_calibrate_cancel", "apemost_hip_calibrate_wait_any",
    "apemost_hip_calibrate_progress", "apemost_hip_calibrate_stats", "apemost_hip_rng_raw",
    "apemost_hip_rng_attempts", "apemost_hip_timer_begin", "apemost_hip_timer_end",
]

_lib = None


def library_path():
    return _build.HIP_LIB


def lib():
    """Load libapemost_hip.so (must have been built by apemost_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME as the system
    # one this library links).  If torch is going to be used for device memory / RCCL plumbing it
    # must be loaded FIRST so that the dynamic linker resolves our DT_NEEDED to the copy already
    # in the process; two runtimes in one process cannot both open the GPU.
    # APEMOST_NO_TORCH=1: the caller will not use torch in this process (bench.py --no-torch, the profile
    # scripts): nothing to order, and the process then holds ONE HIP / HSA runtime -- under rocprofv3,
    # torch's bundled runtime next to the tool's was what aborted at exit (profiles/README.md).
    if not os.environ.get("APEMOST_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    path = library_path()
    if not os.path.exists(path):
        raise ApemostHipError(ERR_NO_DEVICE, "%s not built; run `python -m apemost_amd.build`" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.apemost_hip_last_error.restype = C.c_char_p
    L.apemost_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    L.apemost_hip_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), _up]
    L.apemost_hip_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.apemost_hip_destroy.argtypes = [vp]
    L.apemost_hip_synchronize.argtypes = [vp]
    L.apemost_hip_stream.argtypes = [vp, C.POINTER(vp)]
    L.apemost_hip_waves_per_chain.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.apemost_hip_launch_policy.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.apemost_hip_set_data.argtypes = [vp, _dp]
    L.apemost_hip_set_state.argtypes = [vp, C.POINTER(StateView)]
    L.apemost_hip_get_state.argtypes = [vp, C.POINTER(StateView)]
    L.apemost_hip_set_round.argtypes = [vp, C.c_uint64, C.c_int]
    L.apemost_hip_get_round.argtypes = [vp, _up, C.POINTER(C.c_int)]
    L.apemost_hip_calc_model.argtypes = [vp, C.c_int32, C.c_int32]
    L.apemost_hip_loglike.argtypes = [vp, C.c_int32, _dp, _dp, _dp, _dp]
    L.apemost_hip_launch_round.argtypes = [vp, C.c_uint32, C.c_int, vp]
    L.apemost_hip_launch_round_for.argtypes = [vp, C.c_uint32, C.c_int32, vp]
    L.apemost_hip_launch_rounds.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, vp]
    L.apemost_hip_max_rounds_per_launch.argtypes = [vp, C.POINTER(C.c_int32)]
    L.apemost_hip_run.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
    L.apemost_hip_samples_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.apemost_hip_samples_read.argtypes = [vp, vp, C.c_uint64, _dp]
    L.apemost_hip_samples_free.argtypes = [vp, vp]
    L.apemost_hip_samples_read_async.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.apemost_hip_samples_pack_read_async.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32,
                                                      vp, vp, vp, _up]
    L.apemost_hip_samples_wait.argtypes = [vp]
    L.apemost_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.apemost_hip_host_free.argtypes = [vp]
    L.apemost_hip_swap_pair.argtypes = [C.c_uint64, C.c_uint64, C.c_int64]
    L.apemost_hip_swap_pair.restype = C.c_int64
    L.apemost_hip_sampler_swap_pair.argtypes = [C.c_void_p, C.c_uint64]
    L.apemost_hip_sampler_swap_pair.restype = C.c_int64
    L.apemost_hip_rounds_within_shard.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
    L.apemost_hip_rounds_within_shard.restype = C.c_int64
    L.apemost_hip_edge_doubles.argtypes = [C.c_int32]
    L.apemost_hip_edge_doubles.restype = C.c_int32
    L.apemost_hip_edge_export.argtypes = [vp, C.c_int, vp]
    L.apemost_hip_edge_import.argtypes = [vp, C.c_int, vp]
    L.apemost_hip_edge_exchange.argtypes = [vp, vp]
    L.apemost_hip_run_shards.argtypes = [C.POINTER(vp), C.c_int32, C.c_uint64, C.c_uint32, C.POINTER(vp)]
    L.apemost_hip_calib_defaults.argtypes = [C.POINTER(CalibConfig)]
    L.apemost_hip_calib_defaults.restype = None
    L.apemost_hip_calibrate_chains.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(CalibConfig), C.c_int,
                                               C.POINTER(C.c_int32), _up]
    L.apemost_hip_calibrate_begin.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(CalibConfig), C.c_int]
    L.apemost_hip_calibrate_end.argtypes = [vp, C.POINTER(C.c_int32), _up]
    L.apemost_hip_calibrate_poll.argtypes = [vp, C.POINTER(C.c_int32)]
    L.apemost_hip_calibrate_cancel.argtypes = [vp]
    L.apemost_hip_calibrate_wait_any.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(C.c_int32)]
    L.apemost_hip_calibrate_progress.argtypes = [vp, _dp, C.c_int32, C.POINTER(C.c_int32)]
    L.apemost_hip_calibrate_stats.argtypes = [vp, _up, _up, _up, _dp]
    L.apemost_hip_rng_raw.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32,
                                      C.POINTER(C.c_uint32)]
    L.apemost_hip_rng_attempts.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_int32, C.c_uint64, C.c_uint64,
                                           C.c_int32, _dp, _dp, C.POINTER(C.c_int32), _dp]
    L.apemost_hip_user_model_compile_seconds.argtypes = [vp, _dp]
    L.apemost_hip_timer_begin.argtypes = [vp]
    L.apemost_hip_timer_end.argtypes = [vp, C.POINTER(C.c_float), _up]
    _lib = L
    return L


def check(rc):
    if rc != OK:
        raise ApemostHipError(rc, lib().apemost_hip_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib().apemost_hip_device_count(C.byref(n)))
    return n.value


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus, mem = C.c_int(0), C.c_uint64(0)
    check(lib().apemost_hip_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return name.value.decode(), cus.value, mem.value


def calib_defaults(**overrides):
    c = CalibConfig()
    lib().apemost_hip_calib_defaults(C.byref(c))
    for k, v in overrides.items():
        setattr(c, k, v)
    return c


def swap_pair(seed, round_, n_global):
    return int(lib().apemost_hip_swap_pair(seed, round_, n_global))


def rng_raw(seed, subsequence, offset, n, device=0):
    out = np.zeros(n, dtype=np.uint32)
    check(lib().apemost_hip_rng_raw(device, seed, subsequence, offset, n,
                                    out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def rng_attempts(seed, chain, slot, tick, q0, n, device=0):
    y, s, valid = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
    lu = C.c_double(0)
    check(lib().apemost_hip_rng_attempts(device, seed, chain, slot, tick, q0, n, y.ctypes.data_as(_dp),
                                         s.ctypes.data_as(_dp), valid.ctypes.data_as(C.POINTER(C.c_int32)),
                                         C.byref(lu)))
    return y, s, valid.astype(bool), lu.value
