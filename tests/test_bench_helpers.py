"""bench.py's host-side pieces that need no GPU: the kernel instantiation a sampler launches as rocprofv3 names it
(what decides whether profiled counters may be quoted), the hash of the kernel sources registered with those
counters, the slices of an engine-owned sample buffer (--no-torch), and the rank spawner's failure path."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("apemost_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_expected_kernel_names_are_the_ones_in_the_profile_registry():
    b = _bench()
    assert b.expected_kernel(0, 4, True, True, 0) == "apemost::pt_round_ob_kernel<0, 4, true, false>"
    assert b.expected_kernel(1, 8, True, True, 32) == "apemost::pt_round_ob_kernel<9, 8, true, false>"   # RANDOMSWAP: MODEL + 8
    assert b.expected_kernel(2, 4, True, True, 0, helper=True) == "apemost::pt_round_ob_kernel<2, 4, true, true>"   # with a helper wavefront
    assert b.expected_kernel(3, 1, False, False, 0) == "apemost::pt_round_kernel<3, 1, false, false>"
    assert b.expected_kernel(0, 4, True, False, 4) == "apemost::pt_round_kernel<0, 4, true, true>"       # two-phase, producers
    # every stepping-kernel entry of the registry names a kernel this function can produce
    names = set()
    for model in range(4):
        for variant in (0, 8):
            for waves in (1, 2, 4, 6, 8):
                for lds in (False, True):
                    for ob in (False, True):
                        for helper in (False, True):
                            names.add(b.expected_kernel(model, waves, lds, ob, variant, helper))
    for e in json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))):
        if "kernel" in e:
            assert e["kernel"].replace("void ", "").split("(")[0] in names, e["kernel"]


def test_kernel_sources_hash_is_stable_and_sees_a_change(tmp_path, monkeypatch):
    b = _bench()
    h = b.kernel_sources_sha1()
    assert h == b.kernel_sources_sha1() and len(h) == 40
    root = tmp_path / "r"
    (root / "apemost_amd" / "csrc").mkdir(parents=True)
    (root / "apemost_amd" / "csrc" / "a.h").write_text("x")
    monkeypatch.setattr(b, "ROOT", str(root))
    h1 = b.kernel_sources_sha1()
    (root / "apemost_amd" / "csrc" / "a.h").write_text("y")
    assert b.kernel_sources_sha1() != h1 != h


def test_engine_rows_slices_address_whole_rounds():
    b = _bench()

    class S:
        n_chains, n_par = 3, 4
    rows = b.EngineRows(S(), 10, 5, base=1000)
    per_round = 5 * 3 * 6
    assert rows.data_ptr() == 1000 and rows[2:7].data_ptr() == 1000 + 8 * 2 * per_round and rows[2:7].rounds == 5
    assert rows[2:7][1:3].data_ptr() == 1000 + 8 * 3 * per_round


def test_spawned_ranks_end_with_the_failing_ranks_code():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["APEMOST_BENCH_TEST_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
                          "--cpu-seconds", "0"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 7 and not out.stdout.strip()
