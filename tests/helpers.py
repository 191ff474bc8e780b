"""Shared helpers for parity tests: build matching (device state, oracle ladder) pairs."""
import numpy as np

from apemost_amd import workloads as wl
from apemost_amd.state import LadderState, ALL_FIELDS
from oracle import oracle as orc


def make_pair(w, n_chain, beta_0=0.02, seed=1234, chain_offset=0, n_global=None, init_prob=False):
    """A calibrated-looking ladder: chebyshev betas, steps = step0 * beta^-1/2 (the
    SKIP_CALIBRATE_ALLCHAINS prediction, src/parallel_tempering.c:190-196), identical in the
    device mirror and the oracle."""
    n_global = n_global or n_chain
    st = LadderState.from_params(n_chain, w.start, w.pmin, w.pmax, w.step * 0.3)
    for i in range(n_chain):
        b = orc.get_chain_beta(orc.LADDER_CHEBYSHEV_BETA, chain_offset + i, n_global, beta_0)
        st.beta[i] = b
        st.step[i] = np.minimum(st.step[i] * b ** -0.5, (w.pmax - w.pmin))
    lad = orc.Ladder(w.model, n_chain, w.n_par, w.data, chain_offset=chain_offset)
    to_oracle(st, lad)
    if init_prob:
        for c in range(n_chain):
            orc.calc_model(lad, c)
        st.prob[:] = lad.prob
        st.prior[:] = lad.prior
    rng = orc.Rng(orc.RNG_STREAMS, seed, lad)
    return st, lad, rng


def to_oracle(st, lad):
    for n in ALL_FIELDS:
        if n != "ticks":
            getattr(lad, n)[...] = getattr(st, n)


def assert_match(dev, lad, rng=None, rtol=1e-9, what=""):
    """integer fields bit-exact, fp64 fields to rtol (sin/log differ by ulps between ocml and libm
    and the device sums the data vector in tree order)."""
    for n in ("accept", "reject", "n_iter", "swapcount", "params_accepts", "params_rejects"):
        assert np.array_equal(getattr(dev, n), getattr(lad, n)), "%s %s" % (what, n)
    if rng is not None:
        assert np.array_equal(dev.ticks, rng.ticks), what + " ticks"
    for n in ("params", "params_best", "step", "beta", "prob", "prior", "prob_best"):
        np.testing.assert_allclose(getattr(dev, n), getattr(lad, n), rtol=rtol, atol=1e-300,
                                   err_msg="%s %s" % (what, n))


def small_workloads():
    return {
        "simplesin": wl.simplesin(n_data=256, n_chain=8),
        "sine3": wl.sine3(n_data=300, n_chain=8),
        "pulse": wl.pulse(n_data=257, n_chain=8),
        "pulse_vrot": wl.pulse_vrot(n_data=200, n_chain=8),
    }


class OracleShardEngine:
    """ShardedLadder engine backed by the CPU oracle (test infrastructure): lets the sharding and
    edge-exchange host logic run under gloo without a GPU.  Same interface as HipShardEngine."""

    def __init__(self, lad, seed, n_global, torch, max_rounds=5):
        import contextlib
        self.max_rounds = max_rounds
        self.lad, self.n_global, self.torch, self.seed = lad, n_global, torch, seed
        self.rng = orc.Rng(orc.RNG_STREAMS, seed, lad)
        self.halo = {0: None, 1: None}
        self._null = contextlib.nullcontext

    def swap_pair(self, round_):
        if self.n_global <= 1:
            return -1
        u = orc.philox_stream(self.seed, 2 ** 63, 1, start=4 * round_)[0] / 2 ** 32
        return orc.lib().orc_swap_pair_index(u, self.n_global)

    def rounds_within_shard(self, first_round, max_rounds):
        lo, hi = self.lad.chain_offset, self.lad.chain_offset + self.lad.n_chain
        k = 0
        while k < max_rounds:
            a = self.swap_pair(first_round + k)
            if a >= 0 and (a == lo - 1 or (a == hi - 1 and a + 1 < self.n_global)):
                break
            k += 1
        return k

    def comm_stream(self):
        return self._null()

    def fence(self):
        pass

    def edge_export(self, side):
        c = 0 if side == 0 else self.lad.n_chain - 1
        rec = np.concatenate([[self.lad.beta[c], self.lad.prob[c], self.lad.prob_best[c]],
                              self.lad.params[c], self.lad.params_best[c]])
        return self.torch.from_numpy(rec.copy())

    def edge_import(self, side, buf):
        self.halo[side] = buf.numpy().copy()

    def max_rounds_per_launch(self):
        return self.max_rounds

    def launch_rounds(self, n_rounds, n_steps, apply_swap, samples):
        for j in range(n_rounds):
            if apply_swap or j > 0:
                # swaps inside a launch never involve a halo (the driver guarantees it)
                halos = (self.halo[0], self.halo[1]) if j == 0 else (None, None)
                orc.tempering_interaction_shard(self.lad, self.rng, self.n_global, *halos)
                if j == 0:
                    self.halo = {0: None, 1: None}
            if n_steps:
                out = orc.run_steps(self.lad, self.rng, n_steps, record=samples is not None)
                if samples is not None:
                    samples[j][...] = self.torch.from_numpy(out)
