"""Host mirror of a ladder of chains: the flattened (structure-of-arrays) form of the
reference's `mcmc` struct (src/mcmc_struct.h:30-106) + `parallel_tempering_mcmc`
(src/parallel_tempering_beta.h:65-76).  numpy only; this is what crosses the C ABI."""
import ctypes as C

import numpy as np

from . import capi

F64_2D = ("params", "params_best", "step", "pmin", "pmax")
U64_2D = ("params_accepts", "params_rejects")
F64_1D = ("beta", "prob", "prior", "prob_best")
U64_1D = ("accept", "reject", "n_iter", "swapcount")
U64_1D = U64_1D + ("ticks",)
ALL_FIELDS = F64_2D + U64_2D + F64_1D + U64_1D


class LadderState:
    def __init__(self, n_chain, n_par):
        self.n_chain, self.n_par = int(n_chain), int(n_par)
        for n in F64_2D:
            setattr(self, n, np.zeros((n_chain, n_par)))
        for n in U64_2D:
            setattr(self, n, np.zeros((n_chain, n_par), dtype=np.uint64))
        for n in U64_1D:
            setattr(self, n, np.zeros(n_chain, dtype=np.uint64))
        self.beta = np.ones(n_chain)                 # setup_chains: set_beta(chains[i], 1)
        self.prob = np.full(n_chain, -1e10)          # mcmc_init, src/mcmc.c:47
        self.prior = np.zeros(n_chain)
        self.prob_best = np.full(n_chain, -1e10)     # src/mcmc.c:49

    @classmethod
    def from_params(cls, n_chain, start, pmin, pmax, step):
        """setup_chains() (src/parallel_tempering_config.c:95-123): N_BETA copies of the params
        file; step < 0 means 10 % of the range (src/mcmc_parser.c:84-87)."""
        start, pmin, pmax, step = (np.asarray(a, dtype=np.float64) for a in (start, pmin, pmax, step))
        st = cls(n_chain, len(start))
        st.params[:] = start
        st.params_best[:] = start
        st.pmin[:], st.pmax[:] = pmin, pmax
        st.step[:] = np.where(step < 0, (pmax - pmin) * 0.1, step)
        return st

    def copy(self):
        o = LadderState(self.n_chain, self.n_par)
        for n in ALL_FIELDS:
            getattr(o, n)[...] = getattr(self, n)
        return o

    def slice(self, lo, hi):
        o = LadderState(hi - lo, self.n_par)
        for n in ALL_FIELDS:
            getattr(o, n)[...] = getattr(self, n)[lo:hi]
        return o

    def view(self, fields=ALL_FIELDS):
        v = capi.StateView()
        for n in fields:
            a = getattr(self, n)
            assert a.flags.c_contiguous, n
            if a.dtype == np.float64:
                setattr(v, n, a.ctypes.data_as(C.POINTER(C.c_double)))
            else:
                assert a.dtype == np.uint64, n
                setattr(v, n, a.ctypes.data_as(C.POINTER(C.c_uint64)))
        return v

    # -- reference file formats --------------------------------
    def calibration_results_text(self):
        """write_calibrations_file, src/parallel_tempering_config.c:176-202"""
        lines = []
        for j in range(self.n_chain):
            vals = [self.beta[j]] + list(self.step[j]) + list(self.params[j])
            lines.append("\t".join("%.15e" % v for v in vals) + "\n")
        return "".join(lines)

    def read_calibration_results(self, text, n_chains=None):
        """read_calibration_file, src/parallel_tempering_config.c:130-174 (params_best := params)"""
        toks = text.split()
        n = self.n_chain if n_chains is None else n_chains
        w = 1 + 2 * self.n_par
        if len(toks) < n * w:
            raise ValueError("could not read %d chain calibrations" % n)
        for j in range(n):
            row = [float(t) for t in toks[j * w:(j + 1) * w]]
            self.beta[j] = row[0]
            self.swapcount[j] = 0
            self.step[j] = row[1:1 + self.n_par]
            self.params[j] = row[1 + self.n_par:]
            self.params_best[j] = self.params[j]
