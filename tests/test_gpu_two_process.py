"""Two processes share the one GPU of the test box, each drives its shard of the ladder through
ShardedLadder + HipShardEngine and exchanges edge records over torch.distributed (gloo here; the
driver's multi-GPU run uses the same code with backend nccl = RCCL).  Result must equal the whole
ladder on one sampler, bit for bit."""
import os
import socket

import numpy as np
import pytest

from apemost_amd import workloads as wl
from tests.helpers import make_pair

pytestmark = pytest.mark.gpu
FIELDS = ("params", "params_best", "prob", "prob_best", "accept", "reject", "swapcount", "ticks", "n_iter")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, n_rounds, n_swap, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from apemost_amd.distributed import HipShardEngine, ShardedLadder, shard_bounds
    from apemost_amd.sampler import HipSampler
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = wl.pulse(n_data=96, n_chain=n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    st, _, _ = make_pair(w, n_global, seed=seed)
    s = HipSampler(w.model, w.n_par, hi - lo, w.data, seed=seed, chain_offset=lo, n_chains_global=n_global)
    s.set_state(st.slice(lo, hi))
    ladder = ShardedLadder(HipShardEngine(s, torch), n_global, lo, hi - lo, rank, world, dist)
    ladder.prime()
    samples = torch.zeros((n_rounds, n_swap, hi - lo, w.n_par + 2), dtype=torch.float64, device="cuda")
    ladder.run_sampler(n_rounds, n_swap, samples)
    s.synchronize()
    got = s.get_state()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), exchanges=ladder.exchanges,
             samples=samples.cpu().numpy(), **{f: getattr(got, f) for f in FIELDS})
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_gpu_equal_whole_ladder(tmp_path):
    import torch
    import torch.multiprocessing as mp
    from apemost_amd.sampler import HipSampler
    n_global, n_rounds, n_swap, seed, world = 12, 120, 3, 23, 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, n_rounds, n_swap, seed, str(tmp_path)))
             for port in [_free_port()] for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    w = wl.pulse(n_data=96, n_chain=n_global)
    st, _, _ = make_pair(w, n_global, seed=seed)
    s = HipSampler(w.model, w.n_par, n_global, w.data, seed=seed)
    s.set_state(st)
    d = torch.zeros((n_rounds, n_swap, n_global, w.n_par + 2), dtype=torch.float64, device="cuda")
    s.run_sampler(n_rounds, n_swap, d.data_ptr())
    s.synchronize()
    ref = s.get_state()
    s.close()
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for f in FIELDS:
        assert np.array_equal(np.concatenate([p[f] for p in parts]), getattr(ref, f)), f
    assert np.array_equal(np.concatenate([p["samples"] for p in parts], axis=2), d.cpu().numpy())
    assert sum(int(p["exchanges"]) for p in parts) >= 2


def _calib_worker(rank, world, port, n_global, seed, burn, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from apemost_amd import capi
    from apemost_amd.distributed import calibrate_rest_sharded, shard_bounds
    from apemost_amd.sampler import HipSampler
    from apemost_amd.state import LadderState
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = wl.simplesin(n_data=128, n_chain=n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    st = LadderState.from_params(n_global, w.start, w.pmin, w.pmax, w.step * 0.2)
    s = HipSampler(w.model, 4, hi - lo, w.data, seed=seed, chain_offset=lo, n_chains_global=n_global)
    s.set_state(st.slice(lo, hi))
    cfg = capi.calib_defaults(burn_in_iterations=burn)
    status, beta_0, factors = calibrate_rest_sharded(s, n_global, lo, cfg, dist=dist, rank=rank, torch=torch)
    got = s.get_state()
    np.savez(os.path.join(out_dir, "calib%d.npz" % rank), status=status, beta_0=beta_0, beta=got.beta, step=got.step,
             params=got.params, ticks=got.ticks, prob_best=got.prob_best)
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_calibrate_rest_equals_whole_ladder(tmp_path):
    """calibrate_rest on two shards (two processes, one broadcast of 3*n_par+2 doubles) gives the very
    same betas, step widths and start points as the whole ladder on one sampler"""
    import torch.multiprocessing as mp
    from apemost_amd import capi
    from apemost_amd.sampler import HipSampler
    from apemost_amd.state import LadderState
    n_global, seed, burn, world = 10, 41, 600, 2
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_calib_worker, args=(r, world, port, n_global, seed, burn, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    w = wl.simplesin(n_data=128, n_chain=n_global)
    st = LadderState.from_params(n_global, w.start, w.pmin, w.pmax, w.step * 0.2)
    s = HipSampler(w.model, 4, n_global, w.data, seed=seed)
    s.set_state(st)
    status, beta_0, factors = s.calibrate_rest(capi.calib_defaults(burn_in_iterations=burn))
    ref = s.get_state()
    s.close()
    parts = [np.load(os.path.join(str(tmp_path), "calib%d.npz" % r)) for r in range(world)]
    assert status == 0 and all(int(p["status"]) == 0 for p in parts)
    assert all(float(p["beta_0"]) == beta_0 for p in parts)
    for f in ("beta", "step", "params", "ticks", "prob_best"):
        assert np.array_equal(np.concatenate([p[f] for p in parts]), getattr(ref, f)), f
    assert ref.beta[0] == 1.0 and abs(ref.beta[-1] - beta_0) < 1e-15
