/*
 * apemost_oracle.c -- CPU restatement of APEMoST's parallel-tempering hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see apemost_oracle.h).  Plain C, libm, optional
 * OpenMP for the timed CPU baseline.  Written from the behaviour of the
 * reference (file:line cited per function), not copied from it.
 *
 * Pinning: the mt19937 and Philox streams, the manual's eval example, the reference's parser fixtures and the
 * survey's recorded config-1 counters are reproduced (tests/test_oracle_pins.py, test_oracle_workflow.py).
 * PARITY UNPINNED by reference-held fixtures for sampler trajectories, swaps, calibration and the pulse
 * likelihoods: the reference holds none and cannot be built here (no GSL); DESIGN.md 2.
 */
#include "apemost_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846264338328
#endif

/* =========================================================================
 * RNG layer
 * ========================================================================= */

/* gsl_rng_mt19937 (GSL rng/mt.c; used through src/mcmc.c:27-35): 2002 seeding,
 * seed 0 is replaced by 4357.  KATs: GSL rng/test.c "mt19937, 4357, 1000th =
 * 1186927261"; SURVEY 8(c). */
void orc_mt_seed(orc_mt19937 *g, unsigned long seed) {
    int i;
    if (seed == 0)
        seed = 4357;
    g->mt[0] = (uint32_t)(seed & 0xffffffffUL);
    for (i = 1; i < 624; i++)
        g->mt[i] = (uint32_t)(1812433253UL * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i);
    g->mti = 624;
}

uint32_t orc_mt_next(orc_mt19937 *g) {
    uint32_t k;
    if (g->mti >= 624) {
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (g->mt[kk] & 0x80000000U) | (g->mt[kk + 1] & 0x7fffffffU);
            g->mt[kk] = g->mt[kk + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        for (; kk < 623; kk++) {
            y = (g->mt[kk] & 0x80000000U) | (g->mt[kk + 1] & 0x7fffffffU);
            g->mt[kk] = g->mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        }
        y = (g->mt[623] & 0x80000000U) | (g->mt[0] & 0x7fffffffU);
        g->mt[623] = g->mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
        g->mti = 0;
    }
    k = g->mt[g->mti++];
    k ^= (k >> 11);
    k ^= (k << 7) & 0x9d2c5680U;
    k ^= (k << 15) & 0xefc60000U;
    k ^= (k >> 18);
    return k;
}

/* Philox4x32-10 (Salmon et al., SC'11; Random123 KATs).  Round function and
 * key schedule as published; word order of the output matches rocRAND's
 * philox4x32_10 engine so that stream n == rocrand_init(seed, subseq, n). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    int round;
    for (round = 0; round < 10; round++) {
        uint64_t p0 = (uint64_t)0xD2511F53U * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57U * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9U;
        k1 += 0xBB67AE85U;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t orc_philox_at(uint64_t seed, uint64_t subsequence, uint64_t n) {
    uint32_t ctr[4], key[2], out[4];
    uint64_t block = n >> 2;
    ctr[0] = (uint32_t)block;
    ctr[1] = (uint32_t)(block >> 32);
    ctr[2] = (uint32_t)subsequence;
    ctr[3] = (uint32_t)(subsequence >> 32);
    key[0] = (uint32_t)seed;
    key[1] = (uint32_t)(seed >> 32);
    orc_philox4x32_10(ctr, key, out);
    return out[n & 3];
}

/* gsl_rng_uniform for mt19937: get()/2^32 in [0,1)  (src/mcmc_gettersetter.c:286-288) */
double orc_uniform(orc_rng *r) {
    r->draws++;
    return orc_mt_next(&r->mt) / 4294967296.0;
}

static double uniform_pos(orc_rng *r) {
    double x;
    do {
        x = orc_uniform(r);
    } while (x == 0);
    return x;
}

/* gsl_ran_gaussian, polar Box-Muller, second variate discarded
 * (src/mcmc_gettersetter.c:290-305 default branch) */
double orc_gaussian(orc_rng *r, double sigma) {
    double x, y, r2;
    do {
        x = -1 + 2 * uniform_pos(r);
        y = -1 + 2 * uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0);
    return sigma * y * sqrt(-2.0 * log(r2) / r2);
}

/* One attempt of the same polar method on a tick-addressed Philox block. */
int orc_gaussian_attempt(uint64_t seed, uint64_t chain_global, int slot, uint64_t tick, uint64_t q,
                         double *y_out, double *s_out) {
    const uint64_t subseq = chain_global * ORC_STREAMS_PER_CHAIN + (uint64_t)slot;
    const uint64_t block = (tick << ORC_TICK_SHIFT) | q;
    const uint32_t w0 = orc_philox_at(seed, subseq, 4 * block + 0);
    const uint32_t w1 = orc_philox_at(seed, subseq, 4 * block + 1);
    const double x = -1 + 2 * (w0 / 4294967296.0);
    const double y = -1 + 2 * (w1 / 4294967296.0);
    const double r2 = x * x + y * y;
    if (w0 == 0 || w1 == 0 || r2 > 1.0 || r2 == 0)
        return 0;
    *y_out = y;
    *s_out = sqrt(-2.0 * log(r2) / r2);
    return 1;
}

/* get_next_random_jump (src/mcmc_gettersetter.c:290-305) on the global stream.  GSL 1.x is not
 * vendored in the reference; its published algorithms (randist/logistic.c, randist/flat.c):
 *   gsl_ran_logistic(r, a): do x = gsl_rng_uniform_pos(r) while (x == 1); return a * log(x / (1 - x))
 *   gsl_ran_flat(r, a, b):  u = gsl_rng_uniform(r); return a * (1 - u) + b * u                     */
double orc_jump(orc_rng *r, double sigma, int proposal) {
    if (proposal == ORC_PROPOSAL_LOGISTIC) {
        double x, z;
        do {
            x = uniform_pos(r);
        } while (x == 1);
        z = log(x / (1 - x));
        return sigma * z;
    }
    if (proposal == ORC_PROPOSAL_UNIFORM) {
        const double u = orc_uniform(r);
        return (-sigma) * (1 - u) + sigma * u;
    }
    return orc_gaussian(r, sigma);
}

/* The same three laws on a tick-addressed Philox block: attempt q of (chain, slot) at tick t.
 * Returns 0 when the attempt's uniforms are rejected by the law itself (polar pair outside the
 * unit disc, a zero word where GSL redraws); the caller then moves on to attempt q+1, which is
 * the redraw GSL would make on a sequential stream.  The logistic law reads word 0 of the
 * block, the flat law word 0 (it never rejects). */
int orc_jump_attempt(uint64_t seed, uint64_t chain_global, int slot, uint64_t tick, uint64_t q,
                     int proposal, double sigma, double *jump_out) {
    if (proposal == ORC_PROPOSAL_GAUSSIAN) {
        double y, sq;
        if (!orc_gaussian_attempt(seed, chain_global, slot, tick, q, &y, &sq))
            return 0;
        *jump_out = sigma * y * sq;
        return 1;
    } else {
        const uint64_t subseq = chain_global * ORC_STREAMS_PER_CHAIN + (uint64_t)slot;
        const uint64_t block = (tick << ORC_TICK_SHIFT) | q;
        const uint32_t w0 = orc_philox_at(seed, subseq, 4 * block + 0);
        const double x = w0 / 4294967296.0;
        if (proposal == ORC_PROPOSAL_LOGISTIC) {
            if (w0 == 0)
                return 0;
            *jump_out = sigma * log(x / (1 - x));
        } else {
            *jump_out = (-sigma) * (1 - x) + sigma * x;
        }
        return 1;
    }
}

/* get_next_alog_urandom (src/mcmc_gettersetter.c:307-309).  Real GSL aborts on
 * log(0) (quirk Q8); the restatement defines ln 0 = -inf. */
double orc_accept_log_uniform(uint64_t seed, uint64_t chain_global, int n_par, uint64_t tick) {
    const uint64_t subseq = chain_global * ORC_STREAMS_PER_CHAIN + (uint64_t)n_par;
    const uint32_t w0 = orc_philox_at(seed, subseq, 4 * (tick << ORC_TICK_SHIFT));
    const double u = w0 / 4294967296.0;
    return (u > 0) ? log(u) : -INFINITY;
}

static double alog_urandom(orc_rng *r, const orc_state *s, int chain) {
    if (r->kind == ORC_RNG_GLOBAL_MT) {
        double u = orc_uniform(r);
        return (u > 0) ? log(u) : -INFINITY;
    }
    return orc_accept_log_uniform(r->seed, (uint64_t)(s->chain_offset + chain), s->n_par, r->ticks[chain]);
}

/* mod_double macro, src/mcmc_internal.h:46-48 */
double orc_mod_double(double x, double y) {
    return (x < 0) ? x - y * (int)(x / y - 1) : x - y * (int)(x / y);
}

/* =========================================================================
 * Likelihoods (user plugins of the reference)
 * ========================================================================= */

/* apps/simplesin.c:12-38 ; gsl_sf_sin stands for sin() to ~1 ulp */
static double ll_simplesin(const double *p, const double *data, int n, int nc, double beta,
                           double sigma) {
    double amplitude = p[0], frequency = p[1], phase = p[2], offset = p[3];
    double square_sum = 0;
    int i;
    for (i = 0; i < n; i++) {
        double x = data[(size_t)i * nc + 0];
        double y = data[(size_t)i * nc + 1];
        double m = amplitude * sin(2.0 * M_PI * (frequency * x + phase)) + offset;
        double deltay = m - y;
        square_sum += deltay * deltay;
    }
    return beta * square_sum / (-2 * sigma * sigma);
}

/* own 10-parameter model for BASELINE config 3:
 * y = sum_k A_k sin(2 pi (f_k x + phi_k)) + o ; params (A,f,phi)x3, o */
static double ll_sine3(const double *p, const double *data, int n, int nc, double beta,
                       double sigma) {
    double square_sum = 0;
    int i, k;
    for (i = 0; i < n; i++) {
        double x = data[(size_t)i * nc + 0];
        double y = data[(size_t)i * nc + 1];
        double m = 0;
        double deltay;
        for (k = 0; k < 3; k++)
            m += p[3 * k] * sin(2.0 * M_PI * (p[3 * k + 1] * x + p[3 * k + 2]));
        m += p[9];
        deltay = m - y;
        square_sum += deltay * deltay;
    }
    return beta * square_sum / (-2 * sigma * sigma);
}

/* apps/pulse.c:12-54 */
static double ll_pulse(int n_par, const double *p, const double *data, int n, int nc,
                       double beta, double hmin, double *prior_out) {
    double prior = 0, prob = p[1], lifetime = p[0];
    unsigned int nm;
    int i, j;
    for (j = 2; j < n_par; j += 2)
        prior += log(p[j + 1] + hmin);
    nm = (unsigned int)(n_par - 2) / 2;
    prior = -prior / nm;
    for (i = 0; i < n; i++) {
        double y = 0;
        double freq = data[(size_t)i * nc + 0];
        for (j = 2; j < n_par; j += 2) {
            double distance = p[j] - freq;
            double q = 2 * M_PI * distance * lifetime;
            y += p[j + 1] / (1 + q * q);
        }
        prob += log(y) + data[(size_t)i * nc + 1] / y;
    }
    *prior_out = prior;
    return prior + -beta * prob;
}

/* apps/pulse_vrot.c:12-65 */
static double ll_pulse_vrot(const double *p, const double *data, int n, int nc, double beta,
                            double hmin, double *prior_out) {
    double prior = 0, prob = p[1], lifetime = p[0], vrot = p[2];
    int i;
    prior += log(p[4] + hmin);
    prior += log(p[6] + hmin);
    prior = -prior / 2u;
    for (i = 0; i < n; i++) {
        double y = 0, distance, q;
        double freq = data[(size_t)i * nc + 0];
        distance = p[3] - freq;
        q = 2 * M_PI * distance * lifetime;
        y += p[4] / (1 + q * q);
        distance = p[5] - freq + -1 * vrot;
        q = 2 * M_PI * distance * lifetime;
        y += p[6] / (1 + q * q);
        distance = p[5] - freq;
        q = 2 * M_PI * distance * lifetime;
        y += p[6] / (1 + q * q);
        distance = p[5] - freq + 1 * vrot;
        q = 2 * M_PI * distance * lifetime;
        y += p[6] / (1 + q * q);
        prob += log(y) + data[(size_t)i * nc + 1] / y;
    }
    *prior_out = prior;
    return prior + -beta * prob;
}

/* The reference's other example likelihoods: checkers for the engine's
 * user-supplied device models (apemost_amd/host/examples/device_models/). */
/* apps/simplesin2.c:12-34: amplitude and frequency, phase fixed at 0.3312 */
static double ll_sine2(const double *p, const double *data, int n, int nc, double beta, double sigma) {
    double square_sum = 0;
    int i;
    for (i = 0; i < n; i++) {
        double x = data[(size_t)i * nc];
        double y = p[0] * sin(2.0 * M_PI * (p[1] * x + 0.3312)) - data[(size_t)i * nc + 1];
        square_sum += y * y;
    }
    return beta * square_sum / (-2 * sigma * sigma);
}

/* apps/normal.c:8-34: one parameter, ten peaks, the tallest one counts */
static double ll_normal(const double *p, double beta) {
    double x = p[0], a, b = 0, sigma, pos, height;
    unsigned int i;
    for (i = 0; i < 10; i++) {
        pos = exp(i);
        height = 10 * pow(1.0, i);
        sigma = i;
        if (i % 2 == 0)
            a = -sigma * pow((x - pos) / sigma, 2) / 2 + height;
        else if (x > pos)
            a = -height * (x - pos) / sigma + height;
        else
            a = -height * (pos - x) / sigma + height;
        if (a > b)
            b = a;
    }
    return beta * b;
}

/* apps/bernoulli_example.c:10-49 (its own SIGMA 2): column 0 the outcome, columns 1.. the regressors */
static double ll_bernoulli(int n_par, const double *p, const double *data, int n, int nc, double beta,
                           double *prior_out) {
    double prior = 0, prob = 0;
    int i, j;
    for (j = 1; j < nc; j++)
        prior += -pow(p[j] / 2, 2) / 2;
    for (i = 0; i < n; i++) {
        double eta_i = p[0], p_i;
        for (j = 1; j < n_par; j++)
            eta_i += data[(size_t)i * nc + j] * p[j];
        if (eta_i > 0)
            p_i = 1 / (1 + exp(-eta_i));
        else
            p_i = exp(eta_i) / (1 + exp(eta_i));
        prob += data[(size_t)i * nc] == 0 ? log(1 - p_i) : log(p_i);
    }
    *prior_out = prior;
    return prior + beta * prob;
}

double orc_loglike(int model, int n_par, const double *params, const double *data,
                   int n_data, int n_cols, double beta, double sigma, double hmin,
                   double *prior_out) {
    double prior = 0, prob;
    switch (model) {
    case ORC_MODEL_SIMPLESIN:
        prob = ll_simplesin(params, data, n_data, n_cols, beta, sigma);
        break;
    case ORC_MODEL_SINE3:
        prob = ll_sine3(params, data, n_data, n_cols, beta, sigma);
        break;
    case ORC_MODEL_PULSE:
        prob = ll_pulse(n_par, params, data, n_data, n_cols, beta, hmin, &prior);
        break;
    case ORC_MODEL_PULSE_VROT:
        prob = ll_pulse_vrot(params, data, n_data, n_cols, beta, hmin, &prior);
        break;
    case ORC_MODEL_SINE2:
        prob = ll_sine2(params, data, n_data, n_cols, beta, sigma);
        break;
    case ORC_MODEL_NORMAL:
        prob = ll_normal(params, beta);
        break;
    case ORC_MODEL_BERNOULLI:
        prob = ll_bernoulli(n_par, params, data, n_data, n_cols, beta, &prior);
        break;
    default:
        prob = NAN;
    }
    if (prior_out)
        *prior_out = prior;
    return prob;
}

/* calc_model(m, old): sets m->prob, and m->prior for models that have one.
 * simplesin never touches prior (apps/simplesin.c:36). */
void orc_calc_model(orc_state *s, int c) {
    double prior = 0;
    double prob = orc_loglike(s->model, s->n_par, s->params + (size_t)c * s->n_par, s->data,
                              s->n_data, s->n_cols, s->beta[c], s->sigma, s->hmin, &prior);
    s->prob[c] = prob;
    if (s->model == ORC_MODEL_PULSE || s->model == ORC_MODEL_PULSE_VROT || s->model == ORC_MODEL_BERNOULLI)
        s->prior[c] = prior;
}

/* =========================================================================
 * Single-chain Metropolis (src/markov_chain.c)
 * ========================================================================= */

/* do_step_for: src/markov_chain.c:226-270.  Default: redraw until inside [min,max] (:235-240).
 * With CIRCULAR_PARAMS (:241-265) the first jump is kept; if it leaves the box a circular
 * parameter wraps it (min + mod_double(new-min, max-min)) and any other parameter falls back to
 * the redraw loop -- for those the two branches are the same rule. */
static void do_step_for(orc_state *s, orc_rng *r, int c, int p) {
    size_t k = (size_t)c * s->n_par + p;
    const double step = s->step[k], old_value = s->params[k];
    const double max = s->pmax[k], min = s->pmin[k];
    const int circular = (int)((s->circular >> p) & 1);
    double new_value;
    if (r->kind == ORC_RNG_GLOBAL_MT) {
        new_value = old_value + orc_jump(r, step, s->proposal);
        if (new_value > max || new_value < min) {
            if (circular) {
                new_value = min + orc_mod_double(new_value - min, max - min);
            } else {
                do {
                    new_value = old_value + orc_jump(r, step, s->proposal);
                } while (new_value > max || new_value < min);
            }
        }
    } else {
        uint64_t q = 0;
        for (;; q++) {
            double jump;
            if (!orc_jump_attempt(r->seed, (uint64_t)(s->chain_offset + c), p, r->ticks[c], q, s->proposal,
                                  step, &jump))
                continue;
            new_value = old_value + jump;
            if (!(new_value > max || new_value < min))
                break;
            if (circular) {
                new_value = min + orc_mod_double(new_value - min, max - min);
                break;
            }
        }
    }
    s->params[k] = new_value;
}

/* check_accept: src/markov_chain.c:282-311.  The uniform is drawn only when
 * prob_new < prob_old (or either is NaN). */
int orc_check_accept(double prob_old, double prob_new, orc_rng *r, const orc_state *s,
                     int chain, int *drew) {
    if (drew)
        *drew = 0;
    if (prob_new == prob_old)
        return 1;
    if (prob_new > prob_old)
        return 1;
    if (drew)
        *drew = 1;
    return alog_urandom(r, s, chain) < (prob_new - prob_old) ? 1 : 0;
}

/* markov_chain_step: src/markov_chain.c:369-386 ; counters
 * src/mcmc_gettersetter.c:98-109 ; on reject only prob is restored (revert,
 * :313-315), prior keeps the proposal's value (quirk Q7). */
void orc_markov_chain_step(orc_state *s, orc_rng *r, int c) {
    const int n = s->n_par;
    double prob_old = s->prob[c];
    double old_values[ORC_STREAMS_PER_CHAIN];
    double *par = s->params + (size_t)c * n;
    int p;
    memcpy(old_values, par, sizeof(double) * n);
    for (p = 0; p < n; p++)
        do_step_for(s, r, c, p);
    orc_calc_model(s, c);
    if (orc_check_accept(prob_old, s->prob[c], r, s, c, NULL)) {
        s->accept[c]++;
        for (p = 0; p < n; p++)
            s->params_accepts[(size_t)c * n + p]++;
    } else {
        s->prob[c] = prob_old;
        memcpy(par, old_values, sizeof(double) * n);
        s->reject[c]++;
        for (p = 0; p < n; p++)
            s->params_rejects[(size_t)c * n + p]++;
    }
    if (r->kind == ORC_RNG_STREAMS)
        r->ticks[c]++;
}

/* markov_chain_step_for: src/markov_chain.c:317-333 (calc_model_for of every
 * BASELINE app recomputes the full model, e.g. apps/simplesin.c:40-45) */
void orc_markov_chain_step_for(orc_state *s, orc_rng *r, int c, int p) {
    size_t k = (size_t)c * s->n_par + p;
    double prob_old = s->prob[c];
    double old_value = s->params[k];
    do_step_for(s, r, c, p);
    orc_calc_model(s, c);
    if (orc_check_accept(prob_old, s->prob[c], r, s, c, NULL)) {
        s->params_accepts[k]++;
    } else {
        s->prob[c] = prob_old;
        s->params[k] = old_value;
        s->params_rejects[k]++;
    }
    if (r->kind == ORC_RNG_STREAMS)
        r->ticks[c]++;
}

/* mcmc_check_best: src/mcmc_calculate.c:35-41 */
void orc_check_best(orc_state *s, int c) {
    if (s->prob[c] > s->prob_best[c]) {
        s->prob_best[c] = s->prob[c];
        memcpy(s->params_best + (size_t)c * s->n_par, s->params + (size_t)c * s->n_par,
               sizeof(double) * s->n_par);
    }
}

/* restart_from_best: src/markov_chain.c:29-32 */
void orc_restart_from_best(orc_state *s, int c) {
    memcpy(s->params + (size_t)c * s->n_par, s->params_best + (size_t)c * s->n_par,
           sizeof(double) * s->n_par);
    s->prob[c] = s->prob_best[c];
}

/* reset_accept_rejects: src/mcmc_gettersetter.c:119-127 */
void orc_reset_accept_rejects(orc_state *s, int c) {
    int p;
    for (p = 0; p < s->n_par; p++) {
        s->params_accepts[(size_t)c * s->n_par + p] = 0;
        s->params_rejects[(size_t)c * s->n_par + p] = 0;
    }
    s->accept[c] = 0;
    s->reject[c] = 0;
}

/* =========================================================================
 * beta ladder (src/parallel_tempering_beta.c:53-102)
 * ========================================================================= */

double orc_ladder_beta(int kind, unsigned int i, unsigned int n_beta, double beta_0) {
    switch (kind) {
    case ORC_LADDER_EQUIDISTANT_BETA:
        return beta_0 + i * (1 - beta_0) / (n_beta - 1);
    case ORC_LADDER_EQUIDISTANT_TEMPERATURE:
        return 1 / (1 / beta_0 + i * (1 - 1 / beta_0) / (n_beta - 1));
    case ORC_LADDER_CHEBYSHEV_TEMPERATURE:
        return 1 / (1 / beta_0 + (1 - 1 / beta_0) / 2 * (1 - cos(i * M_PI / (n_beta - 1))));
    case ORC_LADDER_CHEBYSHEV_BETA:
        return beta_0 + (1 - beta_0) / 2 * (1 - cos(i * M_PI / (n_beta - 1)));
    case ORC_LADDER_EQUIDISTANT_STEPWIDTH:
        return beta_0 + pow(i * 1.0 / (n_beta - 1), 2) * (1 - beta_0);
    case ORC_LADDER_CHEBYSHEV_STEPWIDTH:
        return beta_0 + (1 - beta_0) * pow((1 - cos(i * M_PI / (n_beta - 1))) / 2, 2);
    case ORC_LADDER_HOT_CHAINS:
        return beta_0 + 0 * i * n_beta;
    }
    return NAN;
}

/* get_chain_beta: :85-90, reversed so that chain 0 has beta = 1 */
double orc_get_chain_beta(int kind, unsigned int i, unsigned int n_beta, double beta_0) {
    if (n_beta == 1)
        return 1.0;
    return orc_ladder_beta(kind, n_beta - i - 1, n_beta, beta_0);
}

/* calc_beta_0: :92-102 with BETA_0_STEPWIDTH = 1.0 (beta.h:56) */
double orc_calc_beta_0(const orc_state *s, int c, const double *stepwidth_factors) {
    double max = -INFINITY;
    int p;
    for (p = 0; p < s->n_par; p++) {
        size_t k = (size_t)c * s->n_par + p;
        double v = (s->pmax[k] - s->pmin[k]) * 1.0;
        v /= s->step[k];
        v /= stepwidth_factors[p];
        if (v > max)
            max = v;
    }
    return pow(max, -0.5);
}

/* =========================================================================
 * swap (src/parallel_tempering_interaction.c)
 * ========================================================================= */

/* pair choice of parallel_tempering_decide_swap_now: :92 */
int orc_swap_pair_index(double u, int n_beta) {
    return (int)(n_beta * 1000 * u) % (n_beta - 1);
}

/* check_swap_probability: :25-42 */
int orc_swap_decision(double a_beta, double b_beta, double a_prob, double b_prob,
                      double log_u, double *r_out) {
    double r = a_beta * b_prob / b_beta + b_beta * a_prob / a_beta - (a_prob + b_prob);
    if (r_out)
        *r_out = r;
    return (r > log_u) ? 1 : 0;
}

/* parallel_tempering_do_swap: :99-123 -- params exchanged, prob NOT (quirk Q1);
 * the larger prob_best and its params_best are copied to the other chain (Q3) */
static void do_swap(orc_state *s, int a) {
    const int n = s->n_par, b = a + 1;
    double tmp[ORC_STREAMS_PER_CHAIN];
    memcpy(tmp, s->params + (size_t)a * n, sizeof(double) * n);
    memcpy(s->params + (size_t)a * n, s->params + (size_t)b * n, sizeof(double) * n);
    memcpy(s->params + (size_t)b * n, tmp, sizeof(double) * n);
    if (s->prob_best[a] > s->prob_best[b]) {
        s->prob_best[b] = s->prob_best[a];
        memcpy(s->params_best + (size_t)b * n, s->params_best + (size_t)a * n, sizeof(double) * n);
    } else {
        s->prob_best[a] = s->prob_best[b];
        memcpy(s->params_best + (size_t)a * n, s->params_best + (size_t)b * n, sizeof(double) * n);
    }
}

/* tempering_interaction: :125-141 (default branch, decide_swap_now :87-97).
 * GLOBAL_MT: both uniforms come from the global stream.  STREAMS: from the
 * swap stream at position 4*round (+0 pair, +1 accept), then round++.
 * Only valid when this state holds the whole ladder (chain_offset == 0). */
int orc_tempering_interaction(orc_state *s, orc_rng *r, double *trace) {
    const int n_beta = s->n_chain;
    double u0 = 0, u, u2, c, rr = NAN;
    int a, swapped = 0;
    if (r->kind == ORC_RNG_STREAMS) {
        const int k = s->randomswap ? 1 : 0; /* word 0 is then the swap_probability draw */
        u0 = orc_philox_at(r->seed, ORC_SWAP_SUBSEQUENCE, 4 * r->round + 0) / 4294967296.0;
        u = orc_philox_at(r->seed, ORC_SWAP_SUBSEQUENCE, 4 * r->round + k) / 4294967296.0;
        u2 = orc_philox_at(r->seed, ORC_SWAP_SUBSEQUENCE, 4 * r->round + k + 1) / 4294967296.0;
        r->round++;
        if (n_beta == 1)
            return -1;
    } else {
        if (n_beta == 1)
            return -1;
        if (s->randomswap)
            u0 = orc_uniform(r);
        if (s->randomswap && !(u0 < 1.0 / 1))
            return -1; /* before the pair is drawn, :56-57 */
        u = orc_mt_next(&r->mt) / 4294967296.0;
        u2 = orc_mt_next(&r->mt) / 4294967296.0;
        r->draws += 2;
    }
    /* -DRANDOMSWAP: parallel_tempering_decide_swap_random(chains, n_beta, 1), :47-64: one more
     * uniform first, compared with 1.0 / n_swap where the caller passes n_swap = 1 (:131); the
     * partner is (a + 1) % n_beta = a + 1 because a < n_beta - 1 */
    if (s->randomswap && !(u0 < 1.0 / 1))
        return -1;
    a = orc_swap_pair_index(u, n_beta);
    c = (u2 > 0) ? log(u2) : -INFINITY;
    swapped = orc_swap_decision(s->beta[a], s->beta[a + 1], s->prob[a], s->prob[a + 1], c, &rr);
    if (trace) {
        trace[0] = a;
        trace[1] = rr;
        trace[2] = c;
    }
    if (swapped) {
        do_swap(s, a);
        s->swapcount[a]++;
        return a;
    }
    return -1;
}

/* =========================================================================
 * run loop (src/parallel_tempering.c:392-409, single-thread order)
 * ========================================================================= */

static void run_chain_round(orc_state *s, orc_rng *r, int c, uint64_t round, unsigned int n_swap,
                            double *samples) {
    const int n = s->n_par;
    unsigned int sub;
    for (sub = 0; sub < n_swap; sub++) {
        orc_markov_chain_step(s, r, c);
        orc_check_best(s, c);
        s->n_iter[c]++; /* mcmc_append_current_parameters, src/mcmc_calculate.c:30-33 */
        if (samples) {
            double *row = samples + (((size_t)round * n_swap + sub) * s->n_chain + c) * (n + 2);
            memcpy(row, s->params + (size_t)c * n, sizeof(double) * n);
            row[n] = s->prob[c];                    /* prob-chain<i>.dump col 1 */
            row[n + 1] = s->prob[c] - s->prior[c]; /* col 2, parallel_tempering.c:399-401 */
        }
    }
}

void orc_run_steps(orc_state *s, orc_rng *r, unsigned int n_steps, double *samples, int n_threads) {
    int c;
    if (r->kind == ORC_RNG_STREAMS && n_threads > 1) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
        for (c = 0; c < s->n_chain; c++) {
            orc_rng local = *r;
            run_chain_round(s, &local, c, 0, n_steps, samples);
        }
    } else {
        for (c = 0; c < s->n_chain; c++)
            run_chain_round(s, r, c, 0, n_steps, samples);
    }
}

/* tempering_interaction on one shard of a block-partitioned ladder: same pair choice,
 * criterion and do_swap as above (src/parallel_tempering_interaction.c:25-42,87-141), the
 * partner across a shard edge comes from a halo record */
int orc_tempering_interaction_shard(orc_state *s, orc_rng *r, int64_t n_global, const double *halo_lo,
                                    const double *halo_hi, int *swapped_out) {
    const int n = s->n_par;
    const int64_t lo = s->chain_offset, hi = s->chain_offset + s->n_chain;
    double u0, u, u2, c, rr;
    int64_t a;
    int swapped = 0;
    if (swapped_out)
        *swapped_out = 0;
    {
        const int k = s->randomswap ? 1 : 0;
        u0 = orc_philox_at(r->seed, ORC_SWAP_SUBSEQUENCE, 4 * r->round + 0) / 4294967296.0;
        u = orc_philox_at(r->seed, ORC_SWAP_SUBSEQUENCE, 4 * r->round + k) / 4294967296.0;
        u2 = orc_philox_at(r->seed, ORC_SWAP_SUBSEQUENCE, 4 * r->round + k + 1) / 4294967296.0;
    }
    r->round++;
    if (n_global <= 1)
        return -1;
    if (s->randomswap && !(u0 < 1.0 / 1))
        return -1;
    a = orc_swap_pair_index(u, (int)n_global);
    c = (u2 > 0) ? log(u2) : -INFINITY;
    if (a >= lo && a + 1 < hi) { /* both local */
        int la = (int)(a - lo);
        swapped = orc_swap_decision(s->beta[la], s->beta[la + 1], s->prob[la], s->prob[la + 1], c, &rr);
        if (swapped) {
            do_swap(s, la);
            s->swapcount[la]++;
        }
    } else if (a == lo - 1 || a == hi - 1) { /* straddles an edge of this shard */
        const int mine_is_a = (a == hi - 1);
        const int lc = mine_is_a ? s->n_chain - 1 : 0;
        const double *h = mine_is_a ? halo_hi : halo_lo;
        const double h_beta = h[0], h_prob = h[1], h_best = h[2];
        const double *h_params = h + 3, *h_params_best = h + 3 + n;
        double a_beta = mine_is_a ? s->beta[lc] : h_beta, b_beta = mine_is_a ? h_beta : s->beta[lc];
        double a_prob = mine_is_a ? s->prob[lc] : h_prob, b_prob = mine_is_a ? h_prob : s->prob[lc];
        double a_best = mine_is_a ? s->prob_best[lc] : h_best, b_best = mine_is_a ? h_best : s->prob_best[lc];
        swapped = orc_swap_decision(a_beta, b_beta, a_prob, b_prob, c, &rr);
        if (swapped) {
            const int a_wins = a_best > b_best;
            memcpy(s->params + (size_t)lc * n, h_params, sizeof(double) * n);
            if (mine_is_a != a_wins) { /* this chain receives the other one's best */
                s->prob_best[lc] = a_wins ? a_best : b_best;
                memcpy(s->params_best + (size_t)lc * n, h_params_best, sizeof(double) * n);
            }
            if (mine_is_a)
                s->swapcount[lc]++;
        }
    }
    if (swapped_out)
        *swapped_out = swapped;
    return (int)a;
}

/* adapt() with -DADAPT (src/parallel_tempering.c:282-301), called once per round between the
 * n_swap steps and tempering_interaction (:404): the ratio is accepts / REJECTS (not / total),
 * both summed over the parameters (src/mcmc_gettersetter.c:25-41); all step widths of the chain
 * scale by 0.99 or by the double 1 / 0.99; past 100000 counted updates the counters restart */
void orc_adapt(orc_state *s, int c) {
    const int n = s->n_par;
    uint64_t acc = 0, rej = 0;
    double ratio;
    int p;
    for (p = 0; p < n; p++) {
        acc += s->params_accepts[(size_t)c * n + p];
        rej += s->params_rejects[(size_t)c * n + p];
    }
    if (acc + rej < 20000)
        return;
    ratio = acc * 1.0 / rej;
    if (ratio < s->adapt_target - 0.05) {
        for (p = 0; p < n; p++)
            s->step[(size_t)c * n + p] *= 0.99;
    } else if (ratio > s->adapt_target + 0.05) {
        for (p = 0; p < n; p++)
            s->step[(size_t)c * n + p] *= 1 / 0.99;
    }
    if (acc + rej > 100000)
        orc_reset_accept_rejects(s, c);
}

/* adapt() with -DRWM (src/parallel_tempering.c:268-281): per chain, once per round between the n_swap
 * steps and tempering_interaction:  prob_old = get_prob; markov_chain_step; rmw_adapt_stepwidth(prob_old).
 * (The reference's call `markov_chain_step(chains[i], 0)` has one argument too many and does not compile;
 * the function has ONE parameter, src/markov_chain.h, and that is what is restated.)  No mcmc_check follows
 * that step: the best point and n_iter are what the round's own steps left.
 * rmw_adapt_stepwidth (src/markov_chain.c:342-367): alpha = min(1, exp(prob - prob_old)) with the chain's
 * prob AFTER the step (a rejected step leaves prob = prob_old: alpha = 1); every step width
 *   step += U / sqrt(n_iter) * (alpha - TARGET_ACCEPTANCE_RATE) * (max - min),
 * clamped to [MINIMAL_STEPWIDTH, MAXIMAL_STEPWIDTH] * (max - min) = [1e-7, 1e6] * range (:335-340);
 * U = get_next_uniform_random, one per parameter, in parameter order. */
double orc_rwm_uniform(uint64_t seed, uint64_t chain_global, int n_par, uint64_t tick, int p) {
    const uint64_t subseq = chain_global * ORC_STREAMS_PER_CHAIN + (uint64_t)n_par;
    const uint64_t block = (tick << ORC_TICK_SHIFT) | (uint64_t)(1 + p / 4);
    return orc_philox_at(seed, subseq, 4 * block + (uint64_t)(p % 4)) / 4294967296.0;
}

void orc_rwm(orc_state *s, orc_rng *r, int c) {
    const int n = s->n_par;
    const double prob_old = s->prob[c];
    const uint64_t tick = r->kind == ORC_RNG_STREAMS ? r->ticks[c] : 0;
    double alpha;
    int p;
    orc_markov_chain_step(s, r, c);
    alpha = exp(s->prob[c] - prob_old);
    if (alpha > 1)
        alpha = 1;
    for (p = 0; p < n; p++) {
        const size_t k = (size_t)c * n + p;
        const double scale = s->pmax[k] - s->pmin[k];
        const double lo = 0.0000001 * scale, hi = 1000000 * scale;
        const double u = r->kind == ORC_RNG_STREAMS
                             ? orc_rwm_uniform(r->seed, (uint64_t)(s->chain_offset + c), n, tick, p)
                             : orc_uniform(r);
        double step = s->step[k];
        step += u / sqrt(s->n_iter[c]) * (alpha - s->adapt_target) * scale;
        if (step < lo)
            step = lo;
        if (step > hi)
            step = hi;
        s->step[k] = step;
    }
}

void orc_run_sampler(orc_state *s, orc_rng *r, uint64_t n_rounds, unsigned int n_swap,
                     double *samples, int n_threads) {
    uint64_t round;
    int c;
    for (round = 0; round < n_rounds; round++) {
        if (r->kind == ORC_RNG_STREAMS && n_threads > 1) {
            /* chains are independent between swaps; per-stream RNG makes the
             * result independent of the thread count.  `draws` is a statistic
             * only and is not maintained here. */
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
            for (c = 0; c < s->n_chain; c++) {
                orc_rng local = *r;
                run_chain_round(s, &local, c, round, n_swap, samples);
            }
        } else {
            for (c = 0; c < s->n_chain; c++)
                run_chain_round(s, r, c, round, n_swap, samples);
        }
        if (s->rwm) /* (adapt(): the RWM block precedes the ADAPT block) */
            for (c = 0; c < s->n_chain; c++)
                orc_rwm(s, r, c);
        if (s->adapt)
            for (c = 0; c < s->n_chain; c++)
                orc_adapt(s, c);
        orc_tempering_interaction(s, r, NULL);
    }
}

/* =========================================================================
 * calibration (src/markov_chain.c:34-79, src/markov_chain_calibrate.c:1039-1204)
 * ========================================================================= */

void orc_calib_defaults(orc_calib_cfg *c) {
    c->burn_in_iterations = 10000;
    c->rat_limit = 0.5;
    c->target_global = 0.5;
    c->max_ar_deviation = 0.01;
    c->iter_limit = 100000;
    c->mul = 0.85;
    c->adjust_step = 0.5;
    c->iter_readjust = 200;
    c->no_rescaling_limit = 15;
}

/* burn_in: src/markov_chain.c:34-79 */
void orc_burn_in(orc_state *s, orc_rng *r, int c, unsigned int burn_in_iterations) {
    const int n = s->n_par;
    double original_steps[ORC_STREAMS_PER_CHAIN];
    double *step = s->step + (size_t)c * n;
    unsigned long iter, subiter;
    int p;
    memcpy(original_steps, step, sizeof(double) * n);
    for (p = 0; p < n; p++) {
        size_t k = (size_t)c * n + p;
        step[p] = (s->pmax[k] - s->pmin[k]) * 0.1;
    }
    for (iter = 0; iter < burn_in_iterations / 2;) {
        for (subiter = 0; subiter < 200; subiter++)
            orc_markov_chain_step(s, r, c);
        iter += subiter;
        orc_check_best(s, c);
    }
    orc_restart_from_best(s, c);
    for (p = 0; p < n; p++)
        step[p] *= 0.5;
    for (; iter < burn_in_iterations;) {
        for (subiter = 0; subiter < 200; subiter++)
            orc_markov_chain_step(s, r, c);
        iter += subiter;
        orc_check_best(s, c);
    }
    memcpy(step, original_steps, sizeof(double) * n);
}

/* markov_chain_calibrate_orig: src/markov_chain_calibrate.c:1039-1180 */
/* calibration_progress.data (src/markov_chain_calibrate.c:1052, 1141-1146): the reference opens the
 * file "w" at the start of EVERY chain's calibration, in the current directory, and appends one line
 * per parameter at every readjustment.  The oracle writes it only when a path has been set (NULL:
 * off, the default); with several threads the chains clobber one file, as in the reference. */
static const char *orc_progress_path = NULL;
void orc_set_progress_path(const char *path) { orc_progress_path = path; }

int orc_calibrate_orig(orc_state *s, orc_rng *r, int c, const orc_calib_cfg *cfg,
                       uint64_t *iters_out) {
    const int n = s->n_par;
    double *step = s->step + (size_t)c * n;
    const double *pmin = s->pmin + (size_t)c * n, *pmax = s->pmax + (size_t)c * n;
    uint64_t *pa = s->params_accepts + (size_t)c * n, *pr = s->params_rejects + (size_t)c * n;
    double rat_limit = pow(cfg->rat_limit, 1.0 / n);
    unsigned long iter = 0, subiter;
    int nchecks_without_rescaling = 0, reached_perfection = 0, rescaled, p;
    int status = ORC_CALIB_OK;
    FILE *progress_plot_file = orc_progress_path ? fopen(orc_progress_path, "w") : NULL;

    for (p = 0; p < n; p++)
        step[p] *= cfg->adjust_step;
    orc_reset_accept_rejects(s, c);

    while (1) {
        for (p = 0; p < n; p++) {
            orc_markov_chain_step_for(s, r, c, p);
            orc_check_best(s, c);
        }
        iter++;
        if (iter % cfg->iter_readjust == 0) {
            double delta;
            rescaled = 0;
            for (p = 0; p < n; p++) {
                double ar = (double)pa[p] / ((double)pr[p] + (double)pa[p]);
                if (ar > rat_limit + 0.05) {
                    step[p] = step[p] / cfg->mul;
                    if (rescaled == 0)
                        rescaled = -1;
                    if (step[p] / (pmax[p] - pmin[p]) > 1) {
                        step[p] = 1 * (pmax[p] - pmin[p]);
                        if (rescaled == -1)
                            rescaled = 0;
                    }
                    if (step[p] / (pmax[p] - pmin[p]) > 10000) {
                        status = ORC_CALIB_STEP_TOO_LARGE;
                        goto done;
                    }
                    if (rescaled == -1)
                        rescaled = 1;
                }
                if (ar < rat_limit - 0.05) {
                    step[p] = step[p] * cfg->mul;
                    rescaled = 1;
                }
            }
            if (rescaled == 0)
                nchecks_without_rescaling++;
            orc_restart_from_best(s, c);
            orc_reset_accept_rejects(s, c);
            for (subiter = 0; subiter < cfg->iter_readjust; subiter++) {
                orc_markov_chain_step(s, r, c);
                orc_check_best(s, c);
            }
            if (progress_plot_file) {
                for (p = 0; p < n; p++) /* :1141-1146 */
                    fprintf(progress_plot_file, "%d\t%lu\t%f\t%f\t%f\n", p, iter, step[p] / (pmax[p] - pmin[p]),
                            (double)pa[p] / ((double)pr[p] + (double)pa[p]), -1.);
                fflush(progress_plot_file);
            }
            delta = (double)s->accept[c] / (double)(s->accept[c] + s->reject[c]) -
                    cfg->target_global;
            if ((delta < 0 ? -delta : delta) < cfg->max_ar_deviation) {
                reached_perfection = 1;
            } else {
                reached_perfection = 0;
                if (delta < 0)
                    rat_limit /= 0.99;
                else
                    rat_limit *= 0.99;
            }
            if (nchecks_without_rescaling >= cfg->no_rescaling_limit && reached_perfection == 1 &&
                rescaled == 0)
                break;
            if (iter > cfg->iter_limit) {
                status = ORC_CALIB_ITER_LIMIT;
                goto done;
            }
        }
    }
    orc_reset_accept_rejects(s, c);
done:
    if (progress_plot_file)
        fclose(progress_plot_file);
    if (iters_out)
        *iters_out = iter;
    return status;
}

/* markov_chain_calibrate: :1182-1204 (default dispatch) */
int orc_markov_chain_calibrate(orc_state *s, orc_rng *r, int c, const orc_calib_cfg *cfg,
                               uint64_t *iters_out) {
    orc_burn_in(s, r, c, cfg->burn_in_iterations);
    return orc_calibrate_orig(s, r, c, cfg, iters_out);
}

/* calibrate_first: src/parallel_tempering.c:78-95 (file output left to the caller) */
int orc_calibrate_first(orc_state *s, orc_rng *r, const orc_calib_cfg *cfg) {
    orc_calc_model(s, 0);
    return orc_markov_chain_calibrate(s, r, 0, cfg, NULL);
}

static void prepare_rest_chain(orc_state *s, int i, double beta, const double *factors) {
    const int n = s->n_par;
    int p;
    s->beta[i] = beta;
    s->swapcount[i] = 0; /* set_beta zeroes it, src/parallel_tempering_beta.c:25-28 */
    for (p = 0; p < n; p++) {
        s->step[(size_t)i * n + p] = s->step[p] * pow(beta, -0.5);
        if (factors)
            s->step[(size_t)i * n + p] *= factors[p];
        s->params[(size_t)i * n + p] = s->params_best[p];
    }
    orc_calc_model(s, i);
}

/* calibrate_rest: src/parallel_tempering.c:115-207.  Entry state = what
 * setup_chains + read_calibration_file(chains, 1) leave: chain 0 carries the
 * calibrated steps/params (params_best := params), every chain beta = 1. */
int orc_calibrate_rest(orc_state *s, orc_rng *r, const orc_calib_cfg *cfg, int ladder_kind,
                       double beta_0, int skip_calibrate_allchains, int n_threads,
                       double *beta_0_out, double *stepwidth_factors_out) {
    const int n = s->n_par, n_beta = s->n_chain;
    double factors[ORC_STREAMS_PER_CHAIN];
    int p, i, status = ORC_CALIB_OK;
    for (p = 0; p < n; p++)
        factors[p] = 1;
    if (n_beta > 1) {
        double b1 = (beta_0 < 0)
                        ? orc_get_chain_beta(ladder_kind, 1, n_beta, orc_calc_beta_0(s, 0, factors))
                        : orc_get_chain_beta(ladder_kind, 1, n_beta, beta_0);
        int st;
        prepare_rest_chain(s, 1, b1, NULL);
        st = orc_markov_chain_calibrate(s, r, 1, cfg, NULL);
        if (st != ORC_CALIB_OK)
            return st;
        for (p = 0; p < n; p++) {
            factors[p] *= pow(s->beta[1], -0.5);
            factors[p] *= s->step[p];
            factors[p] /= s->step[(size_t)1 * n + p];
        }
    }
    if (beta_0 < 0)
        beta_0 = orc_calc_beta_0(s, 0, factors);
    if (beta_0_out)
        *beta_0_out = beta_0;
    if (stepwidth_factors_out)
        memcpy(stepwidth_factors_out, factors, sizeof(double) * n);

    if (r->kind == ORC_RNG_STREAMS && n_threads > 1) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(dynamic)
#endif
        for (i = 1; i < n_beta; i++) {
            orc_rng local = *r;
            int st;
            prepare_rest_chain(s, i, orc_get_chain_beta(ladder_kind, i, n_beta, beta_0), factors);
            if (!skip_calibrate_allchains) {
                st = orc_markov_chain_calibrate(s, &local, i, cfg, NULL);
                if (st != ORC_CALIB_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
                    status = st;
                }
            } else {
                orc_burn_in(s, &local, i, cfg->burn_in_iterations);
            }
        }
    } else {
        for (i = 1; i < n_beta; i++) {
            prepare_rest_chain(s, i, orc_get_chain_beta(ladder_kind, i, n_beta, beta_0), factors);
            if (!skip_calibrate_allchains) {
                int st = orc_markov_chain_calibrate(s, r, i, cfg, NULL);
                if (st != ORC_CALIB_OK)
                    return st;
            } else {
                orc_burn_in(s, r, i, cfg->burn_in_iterations);
            }
        }
    }
    return status;
}
