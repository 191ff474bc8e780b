"""Multi-process test of the sharded-ladder host logic (apemost_amd/distributed.py) on CPU:
world_size 2 and 3 over gloo, shard engines backed by the oracle.  The result must be bit-identical
to the whole ladder run in one process: the swap pair is derived from the replicated swap stream,
edge records travel point-to-point only when the pair straddles a shard edge."""
import os
import socket

import numpy as np
import pytest

from apemost_amd import workloads as wl
from apemost_amd.distributed import ShardedLadder, shard_bounds
from oracle import oracle as orc
from tests.helpers import OracleShardEngine, make_pair

FIELDS = ("params", "params_best", "prob", "prob_best", "prior", "accept", "reject", "n_iter", "swapcount",
          "params_accepts")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_global, n_rounds, n_swap, seed, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = wl.simplesin(n_data=64, n_chain=n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    st, lad, _ = make_pair(w, hi - lo, seed=seed, chain_offset=lo, n_global=n_global)
    eng = OracleShardEngine(lad, seed, n_global, torch)
    ladder = ShardedLadder(eng, n_global, lo, hi - lo, rank, world, dist)
    ladder.prime()
    ladder.run_sampler(n_rounds, n_swap)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), exchanges=ladder.exchanges, ticks=eng.rng.ticks,
             **{f: getattr(lad, f) for f in FIELDS})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_global", [(2, 8), (3, 7)])
def test_sharded_ladder_equals_whole_ladder(world, n_global, tmp_path):
    import torch.multiprocessing as mp
    n_rounds, n_swap, seed = 60, 5, 31
    mp.spawn(_worker, args=(world, _free_port(), n_global, n_rounds, n_swap, seed, str(tmp_path)), nprocs=world,
             join=True)
    w = wl.simplesin(n_data=64, n_chain=n_global)
    _, whole, rng = make_pair(w, n_global, seed=seed)
    orc.run_sampler(whole, rng, n_rounds, n_swap)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for f in FIELDS:
        got = np.concatenate([p[f] for p in parts])
        assert np.array_equal(got, getattr(whole, f)), f
    assert np.array_equal(np.concatenate([p["ticks"] for p in parts]), rng.ticks)
    ex = [int(p["exchanges"]) for p in parts]
    assert sum(ex) > 0 and sum(ex) % 2 == 0          # every edge exchange involves exactly two ranks
    assert whole.swapcount.sum() > 0


def test_shard_bounds_partition():
    for n in (1, 7, 8, 128, 2048, 16384):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
