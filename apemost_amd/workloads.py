"""Synthetic inputs for the BASELINE configs.

Pure numpy; identical bytes for the HIP engine, the CPU oracle and the tests.
Each workload = (model id, params table, data matrix) in the reference's own
file conventions: a `params` table of (start, min, max, name, step) rows
(src/mcmc_parser.c:47-95) and a row-major `data` matrix (:97-122).
"""
import numpy as np

MODEL_SIMPLESIN, MODEL_PULSE, MODEL_PULSE_VROT, MODEL_SINE3 = 0, 1, 2, 3
MODEL_USER = 4   # device source supplied by the user (include/apemost_device_model.h)
MODEL_NAMES = {0: "simplesin", 1: "pulse", 2: "pulse_vrot", 3: "sine3"}


class Workload:
    def __init__(self, name, model, params, data, n_chain, n_swap):
        self.name, self.model, self.n_chain, self.n_swap = name, model, n_chain, n_swap
        self.names = [p[3] for p in params]
        self.start = np.array([p[0] for p in params], dtype=np.float64)
        self.pmin = np.array([p[1] for p in params], dtype=np.float64)
        self.pmax = np.array([p[2] for p in params], dtype=np.float64)
        step = np.array([p[4] for p in params], dtype=np.float64)
        # step < 0 => 10 % of the range (src/mcmc_parser.c:84-87)
        self.step = np.where(step < 0, (self.pmax - self.pmin) * 0.1, step)
        self.data = np.ascontiguousarray(data, dtype=np.float64)

    @property
    def n_par(self):
        return len(self.start)

    @property
    def n_data(self):
        return self.data.shape[0]

    def bytes_per_step(self):
        """ALGORITHMIC bytes per Metropolis step, SURVEY.md 8(d)."""
        return 16 * self.n_data + 8 * (5 * self.n_par + 4) + 8 * (self.n_par + 2)

    def params_file_text(self):
        return "".join("%.15e\t%.15e\t%.15e\t%s\t%.15e\n" % (s, lo, hi, n, st) for s, lo, hi, n, st in
                       zip(self.start, self.pmin, self.pmax, self.names, self.step))

    def data_file_text(self):
        return "".join("\t".join("%.17e" % v for v in row) + "\n" for row in self.data)


SIMPLESIN_PARAMS = [(0.9, 0, 2, "amplitude", -1), (0.2, 0, 0.3, "frequency", -1),
                    (0.4, 0, 1.0, "phase", -1), (0.5, 0, 2, "offset", -1)]


def simplesin(n_data=1024, n_chain=128, seed=12345, n_swap=None):
    """configs 1/2: x_i = 100+0.5 i ; y = sin(2 pi (0.2 x+0.4)) + 0.5 + N(0,0.5)."""
    rs = np.random.RandomState(seed)
    x = 100 + 0.5 * np.arange(n_data)
    y = 1.0 * np.sin(2 * np.pi * (0.2 * x + 0.4)) + 0.5 + rs.normal(0, 0.5, n_data)
    return Workload("simplesin", MODEL_SIMPLESIN, SIMPLESIN_PARAMS, np.stack([x, y], 1), n_chain,
                    n_swap if n_swap else max(1, 2000 // n_chain))


def sine3(n_data=8192, n_chain=1024, seed=12346, n_swap=None):
    """config 3 ("simplesin5, 10 parameters"): the reference's apps/simplesin5.c is stale
    and has 4 parameters; this is y = sum_k A_k sin(2 pi (f_k x + phi_k)) + o."""
    rs = np.random.RandomState(seed)
    truth = [(1.0, 0.2, 0.4), (0.6, 0.11, 0.1), (0.3, 0.27, 0.7)]
    x = 100 + 0.5 * np.arange(n_data)
    y = sum(a * np.sin(2 * np.pi * (f * x + ph)) for a, f, ph in truth) + 0.5
    y = y + rs.normal(0, 0.5, n_data)
    params = []
    for k, (a, f, ph) in enumerate(truth):
        params += [(0.9 * a, 0, 2, "amplitude%d" % k, -1), (f, max(0.0, f - 0.05), f + 0.05,
                                                            "frequency%d" % k, -1),
                   (ph, 0, 1.0, "phase%d" % k, -1)]
    params.append((0.5, 0, 2, "offset", -1))
    return Workload("sine3", MODEL_SINE3, params, np.stack([x, y], 1), n_chain,
                    n_swap if n_swap else max(1, 2000 // n_chain))


def _lorentz(nu, tau, modes, background):
    y = np.full_like(nu, background)
    for f, h in modes:
        y = y + h / (1 + (2 * np.pi * (f - nu) * tau) ** 2)
    return y


def pulse(n_data=1024, n_chain=2048, seed=7, n_swap=1):
    """config 4: nu = linspace(10,12,n), tau=5, modes (10.6,4.0),(11.3,2.5), background 0.05,
    d = y * Exp(1).  n_swap must be explicit: 2000/n_beta is 0 for n_beta > 2000."""
    rs = np.random.RandomState(seed)
    nu = np.linspace(10, 12, n_data)
    y = _lorentz(nu, 5.0, [(10.6, 4.0), (11.3, 2.5)], 0.05)
    d = y * rs.exponential(1.0, n_data)
    params = [(5.0, 0.1, 50, "lifetime", -1), (0.05, 0, 1, "p1", -1),
              (10.6, 10, 12, "freq0", -1), (4.0, 0, 20, "height0", -1),
              (11.3, 10, 12, "freq1", -1), (2.5, 0, 20, "height1", -1)]
    return Workload("pulse", MODEL_PULSE, params, np.stack([nu, d], 1), n_chain, n_swap)


def pulse_vrot(n_data=65536, n_chain=16384, seed=7, n_swap=1):
    """config 5: as pulse plus rotational splitting vrot = 0.05 on the second mode."""
    rs = np.random.RandomState(seed)
    nu = np.linspace(10, 12, n_data)
    v = 0.05
    y = _lorentz(nu, 5.0, [(10.6, 4.0), (11.3 - v, 2.5), (11.3, 2.5), (11.3 + v, 2.5)], 0.05)
    d = y * rs.exponential(1.0, n_data)
    params = [(5.0, 0.1, 50, "lifetime", -1), (0.05, 0, 1, "p1", -1), (0.05, 0, 0.5, "vrot", -1),
              (10.6, 10, 12, "freq0", -1), (4.0, 0, 20, "height0", -1),
              (11.3, 10, 12, "freq1", -1), (2.5, 0, 20, "height1", -1)]
    return Workload("pulse_vrot", MODEL_PULSE_VROT, params, np.stack([nu, d], 1), n_chain, n_swap)


def by_name(name, **kw):
    return {"simplesin": simplesin, "sine3": sine3, "pulse": pulse, "pulse_vrot": pulse_vrot}[name](**kw)
