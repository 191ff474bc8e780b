"""bench.py with more than one rank, rehearsed on the one GPU of the test box: two ranks under
torch.distributed.run, both on cuda:0, edge records over gloo (RCCL refuses two ranks on one device; the
driver's multi-GPU run is the same code with backend nccl).  Checks the line a scaling run would
record: ranks as torch.distributed reports them, the per-rank rates, the edge exchanges, weak scaling."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_of_a_two_rank_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--same-device",
                          "--cpu-seconds", "0", "--launches-per-step", "3", "--burn-in", "400"],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
    d = line["config"]["distributed"]
    assert line["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and line["scaling"] == "weak"
    assert line["config"]["chains_per_gpu"] == 128 and line["config"]["n_swap"] == 15
    assert d["edge_exchanges_all_ranks"] > 0 and d["edge_exchanges_all_ranks"] % 2 == 0
    assert 0 < d["rank_steps_per_s_min"] <= d["rank_steps_per_s_max"]
    # whole-job steps: both ranks' chains over the slower rank's time
    assert line["value"] <= d["rank_steps_per_s_min"] * 2 * 1.001 and line["value"] > 1e7
    assert line["calibration"]["ok"] == 128 and 0.3 < line["config"]["acceptance_rate_rank0"] < 0.7


def _check_two_rank_line(out):
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                     # ONE JSON line: rank 0's
    line = json.loads(lines[0])
    d = line["config"]["distributed"]
    assert line["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and line["scaling"] == "weak"
    assert d["edge_exchanges_all_ranks"] > 0 and line["value"] > 1e7
    return line


def test_bench_starts_its_own_ranks_when_not_under_torchrun():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment (the way the driver calls N = 1):
    bench.py starts the two ranks itself -- fresh child processes, the parent never touches the GPU -- and
    prints rank 0's line (this used to exit with a message)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--backend", "gloo", "--same-device", "--cpu-seconds", "0", "--launches-per-step", "3",
                          "--burn-in", "400"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    line = _check_two_rank_line(out)
    assert line["roofline"]["kernel"] == "apemost::pt_round_ob_kernel<0, 4, true, false>"


def test_bench_ends_every_rank_when_one_fails():
    """a rank that dies (test hook: rank 1 exits with code 7 before the rendezvous) ends the whole job with
    its exit code instead of leaving rank 0 waiting for it"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["APEMOST_BENCH_TEST_FAIL_RANK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--backend", "gloo", "--same-device", "--cpu-seconds", "0"],
                         cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 7 and not [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
