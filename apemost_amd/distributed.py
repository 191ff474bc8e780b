"""Sharded beta ladders: one process per GPU, the ladder block-partitioned over ranks.

The reference attempts ONE neighbour swap per round for the whole ladder
(src/parallel_tempering_interaction.c:87-97,125-141).  Every rank derives the same pair index
from the replicated swap stream, so a round needs communication only when the chosen pair
straddles a shard edge: then exactly the two ranks involved exchange one edge record
(beta, prob, prob_best, params, params_best = 2*n_par+3 doubles) point-to-point -- RCCL
send/recv over xGMI on GPUs, gloo in the CPU tests.  No collective sits on the data path.
"""
import numpy as np


def shard_bounds(n_global, world, rank):
    """block partition: chain i lives on rank floor(i*world/n_global)"""
    lo = (rank * n_global + world - 1) // world
    hi = ((rank + 1) * n_global + world - 1) // world
    return lo, hi


class ShardedLadder:
    """Drives one shard engine.  The engine provides
         launch_rounds(n_rounds, n_steps, apply_swap, samples) -- rounds of steps with the swaps
                                                      between them, all inside the local shard
         max_rounds_per_launch()
         edge_export(side) -> tensor, edge_import(side, tensor), fence()
         swap_pair(round) -> a,  comm_stream() context manager
       `HipShardEngine` below is the product engine; the gloo tests plug in an oracle-backed one."""

    def __init__(self, engine, n_global, chain_offset, n_local, rank, world, dist=None):
        self.e, self.n_global, self.lo, self.hi = engine, n_global, chain_offset, chain_offset + n_local
        self.rank, self.world, self.dist = rank, world, dist
        self.round = 0
        self.swap_pending = False
        self.exchanges = 0

    def _exchange_if_edge(self):
        a = self.e.swap_pair(self.round)
        if a < 0:
            return
        if a == self.lo - 1:
            side, peer = 0, self.rank - 1
        elif a == self.hi - 1 and a + 1 < self.n_global:
            side, peer = 1, self.rank + 1
        else:
            return
        dist = self.dist
        with self.e.comm_stream():
            send = self.e.edge_export(side)
            recv = send.new_empty(send.shape)
            # Edge exchanges are rare (one pair in n_beta per round), so the two hand-overs between
            # the engine's stream and the communication backend are plain host synchronisations:
            # correct for every backend, whatever stream it works on.
            self.e.fence()
            ops = [dist.P2POp(dist.isend, send, peer), dist.P2POp(dist.irecv, recv, peer)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            self.e.fence()
            self.e.edge_import(side, recv)
        self.exchanges += 1

    def prime(self):
        """Exchange one dummy record with each neighbour so that the communicator and its P2P
        channels exist before anything is timed (RCCL creates them lazily on first use)."""
        if self.dist is None or self.world == 1:
            return
        dist = self.dist
        with self.e.comm_stream():
            for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):
                if peer < 0 or peer >= self.world:
                    continue
                send = self.e.edge_export(side)
                recv = send.new_empty(send.shape)
                self.e.fence()
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send, peer),
                                                   dist.P2POp(dist.irecv, recv, peer)]):
                    req.wait()
                self.e.fence()

    def _straddles(self, swap_index):
        a = self.e.swap_pair(swap_index)
        return a >= 0 and (a == self.lo - 1 or (a == self.hi - 1 and a + 1 < self.n_global))

    def launch_rounds(self, n_rounds, n_steps, samples=None):
        """one engine launch: [pending swap] steps [swap] steps ...; the caller guarantees that the
        n_rounds-1 swaps inside the launch stay within this shard"""
        if self.swap_pending:
            self._exchange_if_edge()
        self.e.launch_rounds(n_rounds, n_steps, self.swap_pending, samples)
        self.round += (1 if self.swap_pending else 0) + (n_rounds - 1)
        self.swap_pending = n_steps > 0

    def launch_round(self, n_steps, samples=None):
        self.launch_rounds(1, n_steps, samples)

    def run_sampler(self, n_rounds, n_swap, samples=None, finalize=True):
        """run_sampler (src/parallel_tempering.c:392-409) on this shard.  samples: array/tensor
        [n_rounds][n_swap][n_local][n_par+2] or None.  Rounds are batched into one engine launch
        up to the next swap attempt that needs a neighbour's record.  With finalize=False the last
        swap attempt stays pending and is applied at the start of the next call."""
        limit = self.e.max_rounds_per_launch()
        r = 0
        while r < n_rounds:
            first_inside = self.round + (1 if self.swap_pending else 0)  # swap index after the launch's 1st round
            k = 1
            while k < min(limit, n_rounds - r) and not self._straddles(first_inside + k - 1):
                k += 1
            self.launch_rounds(k, n_swap, None if samples is None else samples[r:r + k])
            r += k
        if finalize and self.swap_pending:
            self.launch_rounds(1, 0)   # apply the last swap attempt (run_sampler returns after it)


class HipShardEngine:
    """ShardedLadder engine on top of HipSampler + torch tensors in HBM (plumbing only)."""

    def __init__(self, sampler, torch):
        self.s, self.torch = sampler, torch
        self.n_rec = 3 + 2 * sampler.n_par
        self._ext = torch.cuda.ExternalStream(sampler.stream)

    def swap_pair(self, round_):
        from . import capi
        return capi.swap_pair(self.s.seed, round_, self.s.n_chains_global)

    def comm_stream(self):
        # enqueue the NCCL(=RCCL) ops relative to the engine's own stream: no host sync needed
        return self.torch.cuda.stream(self._ext)

    def fence(self):
        """everything queued on the engine's stream and on torch's streams of this device is done"""
        self.s.synchronize()
        self.torch.cuda.synchronize()

    def edge_export(self, side):
        buf = self.torch.empty(self.n_rec, dtype=self.torch.float64, device="cuda")
        self.s.edge_export(side, buf.data_ptr())
        return buf

    def edge_import(self, side, buf):
        self.s.edge_import(side, buf.data_ptr())
        self._keep = buf   # keep alive until the next launch consumed it

    def max_rounds_per_launch(self):
        return self.s.max_rounds_per_launch

    def launch_rounds(self, n_rounds, n_steps, apply_swap, samples):
        self.s.launch_rounds(n_rounds, n_steps, apply_swap, 0 if samples is None else samples.data_ptr())
