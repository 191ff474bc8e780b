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
         swap_pair(round) -> a,  rounds_within_shard(first, max) -> k,  comm_stream() context manager
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
            # RCCL (backend "nccl") is stream-ordered end to end: under comm_stream() the current
            # stream IS the engine's stream, so the export kernel, the send/recv pair (c10d makes
            # its communication stream wait for the current one, and wait() makes the current one
            # wait for the transfer) and the import kernel are chained by events, with no host
            # synchronisation.  Other backends (gloo in the tests) stage through the host and
            # need the data complete on both sides of the transfer.
            ordered = dist.get_backend() == "nccl"
            send = self.e.edge_export(side)
            recv = send.new_empty(send.shape)
            if not ordered:
                self.e.fence()
            ops = [dist.P2POp(dist.isend, send, peer), dist.P2POp(dist.irecv, recv, peer)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if not ordered:
                self.e.fence()
            self.e.edge_import(side, recv)
            self._in_flight = (send, recv)   # keep the buffers alive until the streams are done with them
        self.exchanges += 1

    def prime(self):
        """Exchange one dummy record with each neighbour so that the communicator and its P2P
        channels exist before anything is timed (RCCL creates them lazily on first use)."""
        if self.dist is None or self.world == 1:
            return
        dist = self.dist
        # c10d: batched P2P among a subset of the ranks is defined only after the group's first
        # collective, which every rank must take part in
        dist.barrier()
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
                self.e.fence()   # (priming happens once, before anything is timed: plain fences)

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
        r = 0
        while r < n_rounds:
            # (asked anew for every launch: a refused cooperative launch or a failed hand-off takes the
            # engine to one round per launch, and the next launch must not ask for more)
            limit = self.e.max_rounds_per_launch()
            first_inside = self.round + (1 if self.swap_pending else 0)  # swap index after the launch's 1st round
            # the launch's first round, plus as many more as have their opening swap attempt inside
            # the shard (one call: a ctypes call per round would cost more than the round at n_swap 1)
            k = 1 + self.e.rounds_within_shard(first_inside, min(limit, n_rounds - r) - 1)
            self.launch_rounds(k, n_swap, None if samples is None else samples[r:r + k])
            r += k
        if finalize and self.swap_pending:
            self.launch_rounds(1, 0)   # apply the last swap attempt (run_sampler returns after it)


def calibrate_rest_sharded(sampler, n_global, lo, cfg=None, ladder_kind=0, beta_0=-0.001,
                           skip_calibrate_allchains=False, dist=None, rank=0, torch=None):
    """calibrate_rest() (src/parallel_tempering.c:115-207) on a block-partitioned ladder.

    Entry state: rank 0's chain 0 carries the calibrated steps/params of `calibrate_first`
    (read_calibration_file(chains, 1)), every beta = 1.  Rank 0 (which must hold chains 0 and 1)
    calibrates chain 1 alone to get the per-parameter stepwidth factors and beta_0, broadcasts
    (status, beta_0, factors, steps of chain 0, best point of chain 0) -- 3*n_par+2 doubles, the
    only communication -- and then every rank seeds and calibrates its own chains concurrently,
    one workgroup per chain.  Returns (status, beta_0, factors)."""
    import numpy as np
    from . import capi
    from .sampler import calc_beta_0, get_chain_beta
    n_par, n_local = sampler.n_par, sampler.n_chains
    cfg = cfg or capi.calib_defaults()
    msg = np.zeros(2 + 3 * n_par)
    if rank == 0:
        assert lo == 0 and (n_global == 1 or n_local >= 2), "rank 0 must hold chains 0 and 1"
        st = sampler.get_state()
        factors = np.ones(n_par)
        status = 0
        if n_global > 1:
            b0 = calc_beta_0(st, 0, factors) if beta_0 < 0 else beta_0
            b1 = get_chain_beta(ladder_kind, 1, n_global, b0)
            st.beta[1], st.swapcount[1] = b1, 0
            st.step[1] = st.step[0] * b1 ** -0.5
            st.params[1] = st.params_best[0]
            sampler.set_state(st, ("beta", "swapcount", "step", "params"))
            sampler.calc_model(1, 1)
            stat, _ = sampler.markov_chain_calibrate(1, 1, cfg)
            status = int(stat[0])
            st = sampler.get_state()
            factors = factors * st.beta[1] ** -0.5
            factors = factors * st.step[0]
            factors = factors / st.step[1]
        if beta_0 < 0:
            beta_0 = calc_beta_0(st, 0, factors)
        msg[:] = np.concatenate([[status, beta_0], factors, st.step[0], st.params_best[0]])
    if dist is not None:
        t = torch.from_numpy(msg)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
        msg = t.cpu().numpy()
    status, beta_0 = int(msg[0]), float(msg[1])
    factors, step0, best0 = msg[2:2 + n_par], msg[2 + n_par:2 + 2 * n_par], msg[2 + 2 * n_par:]
    if status != 0:
        return status, None, None
    first = 1 if lo == 0 else 0          # chain 0 itself is never touched again
    if n_local - first > 0:
        st = sampler.get_state()
        for i in range(first, n_local):
            b = get_chain_beta(ladder_kind, lo + i, n_global, beta_0)
            st.beta[i], st.swapcount[i] = b, 0
            st.step[i] = step0 * b ** -0.5
            st.step[i] = st.step[i] * factors
            st.params[i] = best0
        sampler.set_state(st, ("beta", "swapcount", "step", "params"))
        sampler.calc_model(first, n_local - first)
        stat, _ = sampler.markov_chain_calibrate(first, n_local - first, cfg,
                                                 burn_in_only=skip_calibrate_allchains)
        bad = stat[stat != 0]
        if len(bad):
            status = int(bad[0])
    if dist is not None:
        t = torch.tensor([float(status)], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        status = int(t.item())
    return status, beta_0, factors


class HipShardEngine:
    """ShardedLadder engine on top of HipSampler + torch tensors in HBM (plumbing only)."""

    def __init__(self, sampler, torch):
        self.s, self.torch = sampler, torch
        self.n_rec = 3 + 2 * sampler.n_par
        # (torch None: a single rank without torch in the process -- bench.py --no-torch; no exchange ever happens)
        self._ext = torch.cuda.ExternalStream(sampler.stream) if torch is not None else None

    def swap_pair(self, round_):
        return self.s.swap_pair(round_)

    def rounds_within_shard(self, first_round, max_rounds):
        return self.s.rounds_within_shard(first_round, max_rounds)

    def comm_stream(self):
        # enqueue the NCCL(=RCCL) ops relative to the engine's own stream: no host sync needed
        return self.torch.cuda.stream(self._ext)

    def fence(self):
        """everything queued on the engine's stream and on torch's streams of this device is done"""
        self.s.synchronize()
        if self.torch is not None:
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
