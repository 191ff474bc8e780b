"""CPU: the device sources of the reference's three other example likelihoods
(apemost_amd/host/examples/device_models/) compile against the engine's kernel templates exactly as
apemost_hip_create hands them to hiprtc (no GPU needed to compile for gfx950), for the default kernels
and for the variant instantiation; a source with an error comes back with the compiler's message."""
import os

import pytest

from apemost_amd import device_model

MODELS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "apemost_amd", "host", "examples",
                      "device_models")


@pytest.mark.parametrize("name", ["simplesin2", "normal", "bernoulli_example"])
def test_example_device_models_compile_for_gfx950(name):
    for variant in (False, True):
        ok, log, size = device_model.compile_check(os.path.join(MODELS, name + ".hip"), variant=variant)
        assert ok and size > 10000, log


def test_a_broken_device_model_is_reported_with_the_compilers_words(tmp_path):
    bad = tmp_path / "bad.hip"
    bad.write_text('#include "apemost_device_model.h"\n'
                   '__device__ double apemost_user_term(const apemost_model_ctx *c, int i) { return undefined_thing; }\n')
    ok, log, _ = device_model.compile_check(str(bad))
    assert not ok and "undefined_thing" in log and "bad.hip" in log
