#!/usr/bin/env python3
"""Probe: can two ranks share ONE GPU under backend nccl (= RCCL) on this box?  If so, the sharded
driver's RCCL path (export kernel -> isend/irecv on the sampler's stream -> import kernel) can be
exercised without a second GPU: runs a 2-shard ladder and compares it with the whole ladder.
Exit code 0 = ran and matched, 3 = RCCL refused two ranks on one device (nothing wrong with the code)."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    from apemost_amd import workloads as wl
    from apemost_amd.distributed import HipShardEngine, ShardedLadder, shard_bounds
    from apemost_amd.sampler import HipSampler
    from tests.helpers import make_pair
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("rank %d: RCCL refused: %s" % (rank, str(e).splitlines()[0][:200]), flush=True)
        os._exit(3)
    n_global, n_rounds, n_swap, seed = 12, 120, 3, 23
    w = wl.pulse(n_data=96, n_chain=n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    st, _, _ = make_pair(w, n_global, seed=seed)
    s = HipSampler(w.model, w.n_par, hi - lo, w.data, seed=seed, chain_offset=lo, n_chains_global=n_global)
    s.set_state(st.slice(lo, hi))
    ladder = ShardedLadder(HipShardEngine(s, torch), n_global, lo, hi - lo, rank, world, dist)
    ladder.prime()
    ladder.run_sampler(n_rounds, n_swap)
    s.synchronize()
    got = s.get_state()
    np.savez(os.path.join(out, "rank%d.npz" % rank), exchanges=ladder.exchanges, params=got.params, prob=got.prob,
             accept=got.accept, swapcount=got.swapcount)
    s.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    import tempfile
    import numpy as np
    import torch.multiprocessing as mp
    out = tempfile.mkdtemp()
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    codes = [p.exitcode for p in procs]
    if any(c == 3 for c in codes):
        print("RCCL does not allow two ranks on one device here: path not exercised")
        sys.exit(3)
    if any(c != 0 for c in codes):
        print("worker exit codes", codes)
        sys.exit(1)
    import torch
    from apemost_amd import workloads as wl
    from apemost_amd.sampler import HipSampler
    from tests.helpers import make_pair
    w = wl.pulse(n_data=96, n_chain=12)
    st, _, _ = make_pair(w, 12, seed=23)
    s = HipSampler(w.model, w.n_par, 12, w.data, seed=23)
    s.set_state(st)
    s.run_sampler(120, 3)
    s.synchronize()
    ref = s.get_state()
    parts = [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(2)]
    for f in ("params", "prob", "accept", "swapcount"):
        assert np.array_equal(np.concatenate([p[f] for p in parts]), getattr(ref, f)), f
    print("two ranks on one GPU over RCCL: %d edge exchanges, identical to the whole ladder" % sum(int(p["exchanges"]) for p in parts))


if __name__ == "__main__":
    main()
