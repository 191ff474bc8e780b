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


class _OrderedDist:
    """torch.distributed over gloo that SAYS it is nccl: ShardedLadder then takes its stream-ordered branch
    (export kernel -> batched isend / irecv -> wait -> import kernel, no host fence anywhere), the one a
    multi-GPU run over RCCL takes and no one-GPU box can execute.  Every call is logged."""

    def __init__(self, dist, log):
        self._d, self.log = dist, log
        self.P2POp, self.isend, self.irecv = dist.P2POp, dist.isend, dist.irecv

    def get_backend(self):
        return "nccl"

    def barrier(self):
        self.log.append("barrier")
        return self._d.barrier()

    def batch_isend_irecv(self, ops):
        self.log.append("p2p:%s" % ",".join("%s>%d" % (op.op.__name__, op.peer) for op in ops))
        reqs = self._d.batch_isend_irecv(ops)
        log = self.log

        class _Req:
            def __init__(self, r):
                self.r = r

            def wait(self):
                log.append("wait")
                return self.r.wait()
        return [_Req(r) for r in reqs]


def _ordered_worker(rank, world, port, n_global, n_rounds, n_swap, seed, out_dir):
    import json
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = wl.simplesin(n_data=64, n_chain=n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    st, lad, _ = make_pair(w, hi - lo, seed=seed, chain_offset=lo, n_global=n_global)
    eng = OracleShardEngine(lad, seed, n_global, torch, max_rounds=16)
    log = []
    alive = []
    for name in ("edge_export", "edge_import", "fence", "launch_rounds"):
        inner = getattr(eng, name)

        def logged(*a, _inner=inner, _name=name):
            log.append(_name if _name != "launch_rounds" else "launch")
            out = _inner(*a)
            if _name == "edge_export":
                alive.append(out)
            return out
        setattr(eng, name, logged)
    ladder = ShardedLadder(eng, n_global, lo, hi - lo, rank, world, _OrderedDist(dist, log))
    ladder.prime()
    primed = len(log)
    ladder.run_sampler(n_rounds, n_swap)
    # the buffers of the last exchange are still referenced by the ladder (the streams may not be done with them)
    held = ladder.exchanges == 0 or (ladder._in_flight[0] is alive[-1] and ladder._in_flight[1] is not None)
    json.dump({"log": log, "primed": primed, "held": bool(held), "exchanges": ladder.exchanges},
              open(os.path.join(out_dir, "log%d.json" % rank), "w"))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ticks=eng.rng.ticks, **{f: getattr(lad, f) for f in FIELDS})
    dist.barrier()
    dist.destroy_process_group()


def test_stream_ordered_branch_issues_export_transfer_import_in_order(tmp_path):
    """The branch a run over RCCL takes (dist.get_backend() == "nccl"), driven here by gloo under a wrapper that
    reports nccl: per edge exchange exactly export -> ONE batched isend + irecv with the neighbour -> both
    waits -> import -> the launch that consumes the record, no host fence between them (fences only while
    priming, before anything is timed), the first collective (a barrier) ahead of the first batched
    point-to-point exchange as c10d requires, the buffers of the exchange kept alive by the ladder -- and
    the ladder still equals the whole ladder bit for bit (RCCL with more than one rank
    has never executed; this is the order it will execute)."""
    import json
    import torch.multiprocessing as mp
    world, n_global, n_rounds, n_swap, seed = 2, 8, 120, 3, 53
    mp.spawn(_ordered_worker, args=(world, _free_port(), n_global, n_rounds, n_swap, seed, str(tmp_path)), nprocs=world, join=True)
    w = wl.simplesin(n_data=64, n_chain=n_global)
    _, whole, rng = make_pair(w, n_global, seed=seed)
    orc.run_sampler(whole, rng, n_rounds, n_swap)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for f in FIELDS:
        assert np.array_equal(np.concatenate([p[f] for p in parts]), getattr(whole, f)), f
    assert np.array_equal(np.concatenate([p["ticks"] for p in parts]), rng.ticks)
    for r in range(world):
        rec = json.load(open(os.path.join(str(tmp_path), "log%d.json" % r)))
        log, primed = rec["log"], rec["primed"]
        peer = 1 - r
        assert log[0] == "barrier" and rec["held"] and rec["exchanges"] > 0
        assert log[:primed] == ["barrier", "edge_export", "fence", "p2p:isend>%d,irecv>%d" % (peer, peer), "wait", "wait", "fence"]
        run = log[primed:]
        assert "fence" not in run and "barrier" not in run
        seq = "p2p:isend>%d,irecv>%d" % (peer, peer)
        n_ex = 0
        i = 0
        while i < len(run):
            if run[i] == "edge_export":
                assert run[i:i + 6] == ["edge_export", seq, "wait", "wait", "edge_import", "launch"], run[i:i + 6]
                n_ex += 1
                i += 6
            else:
                assert run[i] == "launch", run[i]
                i += 1
        assert n_ex == rec["exchanges"]


def test_shard_bounds_partition():
    for n in (1, 7, 8, 128, 2048, 16384):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
