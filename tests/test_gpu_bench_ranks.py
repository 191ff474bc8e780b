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
