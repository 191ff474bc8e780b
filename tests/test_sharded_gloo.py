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


def _worker(rank, world, port, n_global, n_rounds, n_swap, seed, out_dir, max_rounds=5):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = wl.simplesin(n_data=64, n_chain=n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    st, lad, _ = make_pair(w, hi - lo, seed=seed, chain_offset=lo, n_global=n_global)
    eng = OracleShardEngine(lad, seed, n_global, torch, max_rounds=max_rounds)
    launches = [0]
    inner = eng.launch_rounds

    def counted(*a):
        launches[0] += 1
        return inner(*a)
    eng.launch_rounds = counted
    ladder = ShardedLadder(eng, n_global, lo, hi - lo, rank, world, dist)
    ladder.prime()
    ladder.run_sampler(n_rounds, n_swap)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), exchanges=ladder.exchanges, ticks=eng.rng.ticks, launches=launches[0],
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


def test_eight_ranks_at_config_4_proportions(tmp_path):
    """world 8, 64 chains, n_swap 1 (the reference's 2000 / n_beta rule gives BASELINE config 4 and 5 a
    swap attempt after every step), 640 rounds, launches of up to 64 rounds: each of the seven shard
    edges is straddled many times, every rank cuts its launches at the rounds where ITS edges fire
    (rounds_within_shard), so the ranks' launch sequences differ -- and the ladder is still the whole
    ladder, bit for bit."""
    import torch.multiprocessing as mp
    world, n_global, n_rounds, n_swap, seed = 8, 64, 640, 1, 47
    mp.spawn(_worker, args=(world, _free_port(), n_global, n_rounds, n_swap, seed, str(tmp_path), 64), nprocs=world,
             join=True)
    w = wl.simplesin(n_data=64, n_chain=n_global)
    _, whole, rng = make_pair(w, n_global, seed=seed)
    orc.run_sampler(whole, rng, n_rounds, n_swap)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for f in FIELDS:
        assert np.array_equal(np.concatenate([p[f] for p in parts]), getattr(whole, f)), f
    assert np.array_equal(np.concatenate([p["ticks"] for p in parts]), rng.ticks)
    ex = [int(p["exchanges"]) for p in parts]
    # the pairs of the 640 attempts as every rank derived them: edge k | k+1 is straddled when the pair is 8k+7
    eng = OracleShardEngine(whole, seed, n_global, None)
    pairs = [eng.swap_pair(r) for r in range(n_rounds)]
    per_edge = [sum(1 for a in pairs if a == 8 * k + 7) for k in range(7)]
    assert min(per_edge) >= 1 and sum(per_edge) >= 30
    assert ex == [per_edge[0]] + [per_edge[k - 1] + per_edge[k] for k in range(1, 7)] + [per_edge[6]]
    launches = [int(p["launches"]) for p in parts]
    assert len(set(launches)) > 1 and max(launches) < n_rounds // 2      # multi-round launches, cut differently per rank
    assert whole.swapcount.sum() > 0


def test_shard_bounds_partition():
    for n in (1, 7, 8, 128, 2048, 16384):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
