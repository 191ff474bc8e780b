"""The C ABI fails loudly: bad arguments are rejected, and without a gfx950 device every compute
entry point returns APEMOST_HIP_ERR_NO_DEVICE -- there is no CPU fallback to fall into."""
import ctypes as C

import numpy as np
import pytest

from apemost_amd import build, capi, workloads as wl


def _cfg(**kw):
    base = dict(abi_version=capi.ABI_VERSION, device=0, model=wl.MODEL_SIMPLESIN, n_par=4, n_chains=4, n_data=16,
                n_cols=2, waves_per_chain=0, lds_policy=0, flags=0, chain_offset=0, n_chains_global=4, seed=1,
                sigma=0.5, hmin=1e-6)
    base.update(kw)
    return capi.Config(**base)


def _create(cfg):
    build.build_hip()
    h = C.c_void_p()
    rc = capi.lib().apemost_hip_create(C.byref(cfg), C.byref(h))
    if rc == capi.OK:
        capi.lib().apemost_hip_destroy(h)
    return rc, capi.lib().apemost_hip_last_error().decode()


@pytest.mark.parametrize("kw,needle", [
    (dict(abi_version=99), "ABI version"),
    (dict(n_par=0), "n_par"),
    (dict(n_par=63), "n_par"),
    (dict(n_chains=0), "n_chains"),
    (dict(n_cols=1), "n_cols"),
    (dict(chain_offset=2), "outside ladder"),
    (dict(model=17), "unknown device model"),
    (dict(model=wl.MODEL_PULSE, n_par=5), "pulse needs"),
    (dict(model=wl.MODEL_PULSE_VROT, n_par=6), "pulse_vrot needs"),
    (dict(model=wl.MODEL_SINE3, n_par=4), "sine3 needs"),
])
def test_create_rejects_bad_configs(kw, needle):
    rc, msg = _create(_cfg(**kw))
    assert rc in (capi.ERR_INVALID, capi.ERR_UNSUPPORTED) and needle in msg, (rc, msg)


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, msg = _create(_cfg())
    assert rc == capi.ERR_NO_DEVICE and "no HIP device" in msg
    with pytest.raises(capi.ApemostHipError):
        capi.rng_raw(0, 0, 0, 4)
    with pytest.raises(capi.ApemostHipError):
        from apemost_amd.sampler import HipSampler
        HipSampler(wl.MODEL_SIMPLESIN, 4, 2, np.zeros((8, 2)))


def test_swap_pair_is_host_arithmetic_only():
    # the one ABI function that needs no device: which pair the next swap attempt touches
    pairs = [capi.swap_pair(7, r, 128) for r in range(200)]
    assert all(0 <= a < 127 for a in pairs) and len(set(pairs)) > 50
    assert capi.swap_pair(7, 3, 1) == -1
