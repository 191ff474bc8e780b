// pt_device.h -- device-side building blocks of the gfx950 parallel-tempering engine.
//
// One workgroup (WAVES wavefronts of 64 lanes) owns one tempered chain.
//   * wave 0 carries the chain: lane p < n_par owns parameter p (value, best,
//     step width, bounds, per-parameter counters and the rocRAND stream that
//     feeds its proposals); lane n_par owns the accept stream.
//   * every wave evaluates a slice of the data vector; partial sums are combined
//     with a fixed-order butterfly + LDS tree, so results are run-to-run identical.
//
// Reference semantics restated here (APEMoST tree, file:line):
//   do_step_for        src/markov_chain.c:226-240      (Gaussian proposal, redraw at bounds)
//   check_accept       src/markov_chain.c:282-311
//   markov_chain_step  src/markov_chain.c:369-386
//   markov_chain_step_for src/markov_chain.c:317-333
//   mcmc_check_best    src/mcmc_calculate.c:35-41
//   gsl_ran_gaussian (polar), gsl_rng_uniform[_pos] via src/mcmc_gettersetter.c:283-309
//   calc_model         apps/simplesin.c:12-38, apps/pulse.c:12-54, apps/pulse_vrot.c:12-65
#pragma once

#include <hip/hip_runtime.h>

#define ROCRAND_DETAIL_BM_NOT_IN_STATE
#include <rocrand/rocrand_philox4x32_10.h>

#include "apemost_hip.h"

namespace apemost {

typedef unsigned long long u64;

constexpr int kWave = 64;
constexpr double kTwoPi = 2.0 * 3.14159265358979323846264338328; // 2*M_PI, exact in fp64

// ---------------------------------------------------------------------------
// rocRAND Philox4x32-10 stream with a draw counter (the counter is the only
// RNG state kept in HBM between launches).
// ---------------------------------------------------------------------------
// rocRAND's engine indexes its 4-word result block with a run-time subscript, which
// hipcc lowers to scratch (private) memory; this subclass keeps rocRAND's counter layout, key
// schedule and ten_rounds() but selects the word with compares so the state stays in VGPRs.
struct PhiloxRegs : public rocrand_device::philox4x32_10_engine {
    __device__ __forceinline__ PhiloxRegs() {}
    __device__ __forceinline__ void start(u64 seed, u64 subsequence, u64 offset) {
        this->seed(seed, subsequence, offset);
    }
    __device__ __forceinline__ unsigned int next_word() {
        const uint4 r = m_state.result;
        const unsigned int ss = m_state.substate;
        const unsigned int ret = ss == 0 ? r.x : (ss == 1 ? r.y : (ss == 2 ? r.z : r.w));
        if (ss == 3) {
            m_state.substate = 0;
            this->discard_state();
            m_state.result = this->ten_rounds(m_state.counter, m_state.key);
        } else {
            m_state.substate = ss + 1;
        }
        return ret;
    }
};

struct Stream {
    PhiloxRegs st;
    u64 n;

    __device__ __forceinline__ void init(u64 seed, u64 subsequence, u64 offset) {
        st.start(seed, subsequence, offset);
        n = offset;
    }
    __device__ __forceinline__ unsigned int next() {
        n++;
        return st.next_word();
    }
    // gsl_rng_uniform of a 32-bit generator: x / 2^32 in [0,1)
    __device__ __forceinline__ double uniform() { return next() * (1.0 / 4294967296.0); }
    __device__ __forceinline__ double uniform_pos() {
        double x;
        do {
            x = uniform();
        } while (x == 0);
        return x;
    }
    // gsl_ran_gaussian: polar Box-Muller, second variate discarded
    __device__ __forceinline__ double gaussian(double sigma) {
        double x, y, r2;
        do {
            x = -1 + 2 * uniform_pos();
            y = -1 + 2 * uniform_pos();
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0);
        return sigma * y * sqrt(-2.0 * log(r2) / r2);
    }
    // gsl_sf_log(gsl_rng_uniform()): ln 0 is -inf here (real GSL aborts; quirk Q8)
    __device__ __forceinline__ double alog_uniform() { return log(uniform()); }
};

// Cross-lane moves on the DPP path (no LDS crossbar): the 64-bit value travels as two dwords.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double read_lane(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, same value in every lane, fixed association:
// pairs, quads, octets, rows of 16 (DPP quad_perm / row_half_mirror / row_mirror), then
// (row0+row1)+(row2+row3) through scalar registers.
__device__ __forceinline__ double wave_allreduce_sum(double v) {
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v); // row_half_mirror
    v += dpp_move<0x140>(v); // row_mirror
    const double r0 = read_lane(v, 0), r1 = read_lane(v, 16), r2 = read_lane(v, 32), r3 = read_lane(v, 48);
    return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ double lane_bcast(double v, int src) { return __shfl(v, src, kWave); }

// ---------------------------------------------------------------------------
// Likelihood models.  term() is one data point's contribution, finish() turns
// the reduced sum into m->prob (and m->prior).  par points to the proposed
// parameter vector in LDS (broadcast reads).
// ---------------------------------------------------------------------------
struct ModelConsts {
    double sigma;
    double hmin;
};

template <int MODEL>
struct Model;

template <>
struct Model<APEMOST_MODEL_SIMPLESIN> {
    static constexpr bool kHasPrior = false;
    double a, f, ph, o;
    __device__ __forceinline__ void load(const double *par, int) {
        a = par[0];
        f = par[1];
        ph = par[2];
        o = par[3];
    }
    __device__ __forceinline__ double term(double x, double y) const {
        double m = a * sin(kTwoPi * (f * x + ph)) + o;
        double d = m - y;
        return d * d;
    }
    __device__ __forceinline__ double finish(double sum, double beta, const ModelConsts &c,
                                             double *prior) const {
        (void)prior;
        return beta * sum / (-2 * c.sigma * c.sigma);
    }
};

template <>
struct Model<APEMOST_MODEL_SINE3> {
    static constexpr bool kHasPrior = false;
    double a[3], f[3], ph[3], o;
    __device__ __forceinline__ void load(const double *par, int) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            a[k] = par[3 * k];
            f[k] = par[3 * k + 1];
            ph[k] = par[3 * k + 2];
        }
        o = par[9];
    }
    __device__ __forceinline__ double term(double x, double y) const {
        double m = 0;
#pragma unroll
        for (int k = 0; k < 3; k++)
            m += a[k] * sin(kTwoPi * (f[k] * x + ph[k]));
        m += o;
        double d = m - y;
        return d * d;
    }
    __device__ __forceinline__ double finish(double sum, double beta, const ModelConsts &c,
                                             double *prior) const {
        (void)prior;
        return beta * sum / (-2 * c.sigma * c.sigma);
    }
};

template <>
struct Model<APEMOST_MODEL_PULSE> {
    static constexpr bool kHasPrior = true;
    const double *p;
    int n_par;
    double lifetime;
    __device__ __forceinline__ void load(const double *par, int n) {
        p = par;
        n_par = n;
        lifetime = par[0];
    }
    __device__ __forceinline__ double term(double freq, double d) const {
        double y = 0;
        for (int j = 2; j < n_par; j += 2) {
            double distance = p[j] - freq;
            double q = kTwoPi * distance * lifetime;
            y += p[j + 1] / (1 + q * q);
        }
        return log(y) + d / y;
    }
    __device__ __forceinline__ double finish(double sum, double beta, const ModelConsts &c,
                                             double *prior) const {
        double pr = 0;
        for (int j = 2; j < n_par; j += 2)
            pr += log(p[j + 1] + c.hmin);
        pr = -pr / (double)((unsigned)(n_par - 2) / 2u);
        *prior = pr;
        return pr + -beta * (p[1] + sum);
    }
};

template <>
struct Model<APEMOST_MODEL_PULSE_VROT> {
    static constexpr bool kHasPrior = true;
    double lifetime, p1, vrot, fa, ha, fb, hb;
    __device__ __forceinline__ void load(const double *par, int) {
        lifetime = par[0];
        p1 = par[1];
        vrot = par[2];
        fa = par[3];
        ha = par[4];
        fb = par[5];
        hb = par[6];
    }
    __device__ __forceinline__ double term(double freq, double d) const {
        double y = 0, distance, q;
        distance = fa - freq;
        q = kTwoPi * distance * lifetime;
        y += ha / (1 + q * q);
        distance = fb - freq + -1 * vrot;
        q = kTwoPi * distance * lifetime;
        y += hb / (1 + q * q);
        distance = fb - freq;
        q = kTwoPi * distance * lifetime;
        y += hb / (1 + q * q);
        distance = fb - freq + 1 * vrot;
        q = kTwoPi * distance * lifetime;
        y += hb / (1 + q * q);
        return log(y) + d / y;
    }
    __device__ __forceinline__ double finish(double sum, double beta, const ModelConsts &c,
                                             double *prior) const {
        double pr = 0;
        pr += log(ha + c.hmin);
        pr += log(hb + c.hmin);
        pr = -pr / 2.0;
        *prior = pr;
        return pr + -beta * (p1 + sum);
    }
};

// ---------------------------------------------------------------------------
// Device memory of one sampler.  "rows" arrays have n_chains+2 rows: row 0 and
// row n_chains+1 are halo slots for the swap partner on a neighbouring GPU.
// The fields a swap reads from the partner chain are double-buffered so the
// fused swap-in of round r+1 reads round r's values race-free.
// ---------------------------------------------------------------------------
struct DevArrays {
    double *params[2];      // [rows][n_par]
    double *params_best[2]; // [rows][n_par]
    double *prob[2];        // [rows]
    double *prob_best[2];   // [rows]
    double *prior[2];       // [rows]
    double *beta;           // [rows]
    double *step;           // [n_chains][n_par]
    double *pmin;           // [n_chains][n_par]
    double *pmax;           // [n_chains][n_par]
    u64 *params_accepts;    // [n_chains][n_par]
    u64 *params_rejects;    // [n_chains][n_par]
    u64 *accept;            // [n_chains]
    u64 *reject;            // [n_chains]
    u64 *n_iter;            // [n_chains]
    u64 *swapcount;         // [n_chains]
    u64 *rng_offsets;       // [n_chains][n_par+1]
    const double *data;     // column-major [n_cols][n_data]: x = col 0, y = col 1
};

struct ChainShape {
    int n_par;
    int n_data;
    int n_chains;
    long long chain_offset;
    long long n_global;
    u64 seed;
    ModelConsts consts;
};

// ---------------------------------------------------------------------------
// The per-workgroup engine
// ---------------------------------------------------------------------------
template <int MODEL, int WAVES, bool LDS_DATA>
struct Engine {
    static constexpr int kThreads = WAVES * kWave;

    // geometry
    int tid, lane, wave;
    int n_par, n_data;
    ModelConsts consts;
    // data (LDS or global), scratch
    const double *xs, *ys;
    double *s_par;  // [2][64] proposed parameter vectors
    double *s_part; // [2][WAVES] per-wave partial sums
    int parity;

    // chain registers (meaningful in wave 0)
    double cur, best, stepw, lo, hi; // lane p < n_par
    u64 pacc, prej;                  // lane p < n_par
    Stream rng;                      // lanes 0..n_par
    double prob, prior, prob_best, beta; // uniform across wave 0
    u64 accept, reject;                  // uniform across wave 0

    // sum over the data vector of Model::term at the parameter vector par[]
    __device__ __forceinline__ double reduce_data(const double *par, double *prior_new) {
        Model<MODEL> m;
        m.load(par, n_par);
        double acc = 0;
        for (int i = tid; i < n_data; i += kThreads)
            acc += m.term(xs[i], ys[i]);
        acc = wave_allreduce_sum(acc);
        if (WAVES > 1) {
            if (lane == 0)
                s_part[parity * WAVES + wave] = acc;
            __syncthreads();
            acc = s_part[parity * WAVES];
#pragma unroll
            for (int w = 1; w < WAVES; w++)
                acc += s_part[parity * WAVES + w];
        }
        return m.finish(acc, beta_for_all(), consts, prior_new);
    }

    // beta is needed by every wave for finish(); it is uniform per workgroup
    double beta_all;
    __device__ __forceinline__ double beta_for_all() const { return beta_all; }

    // calc_model at the current parameters (no proposal)
    __device__ __forceinline__ void calc_model_current() {
        double *par = s_par + parity * kWave;
        if (wave == 0 && lane < n_par)
            par[lane] = cur;
        __syncthreads();
        double prior_new = prior;
        double p = reduce_data(par, &prior_new);
        if (wave == 0) {
            prob = p;
            if (Model<MODEL>::kHasPrior)
                prior = prior_new;
        }
        parity ^= 1;
    }

    // One Metropolis update.  which < 0: all parameters (markov_chain_step);
    // which = p: parameter p only (markov_chain_step_for).
    __device__ __forceinline__ void step(int which) {
        double *par = s_par + parity * kWave;
        double prop = cur;
        if (wave == 0 && lane < n_par) {
            if (which < 0 || which == lane) {
                do {
                    prop = cur + rng.gaussian(stepw);
                } while (prop > hi || prop < lo);
            }
            par[lane] = prop;
        }
        __syncthreads();
        double prior_new = prior;
        double prob_new = reduce_data(par, &prior_new);
        if (wave == 0) {
            bool acc;
            if (prob_new == prob) {
                acc = true;
            } else if (prob_new > prob) {
                acc = true;
            } else {
                double lu = 0;
                if (lane == n_par)
                    lu = rng.alog_uniform();
                lu = lane_bcast(lu, n_par);
                acc = lu < (prob_new - prob);
            }
            if (Model<MODEL>::kHasPrior)
                prior = prior_new; // not restored on reject (quirk Q7)
            if (acc) {
                cur = prop;
                prob = prob_new;
                if (which < 0) {
                    accept++;
                    pacc++;
                } else if (lane == which) {
                    pacc++;
                }
            } else {
                if (which < 0) {
                    reject++;
                    prej++;
                } else if (lane == which) {
                    prej++;
                }
            }
        }
        parity ^= 1;
    }

    __device__ __forceinline__ void check_best() {
        if (prob > prob_best) {
            prob_best = prob;
            best = cur;
        }
    }
    __device__ __forceinline__ void restart_from_best() {
        cur = best;
        prob = prob_best;
    }
    __device__ __forceinline__ void reset_accept_rejects() {
        pacc = prej = 0;
        accept = reject = 0;
    }
};

} // namespace apemost
