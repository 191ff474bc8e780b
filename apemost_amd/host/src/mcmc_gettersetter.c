/* Accessors of the chain struct (behaviour of reference src/mcmc_gettersetter.c:25-309). */
#include "mcmc.h"
#include "gsl_helper.h"
#include <gsl/gsl_rng.h>
#include <gsl/gsl_randist.h>
#include <gsl/gsl_sf.h>

static unsigned long sum_counters(const unsigned long *c, unsigned int n) {
    unsigned long s = 0;
    unsigned int i;
    for (i = 0; i < n; i++)
        s += c[i];
    return s;
}

unsigned long get_params_accepts_sum(const mcmc *m) { return sum_counters(m->params_accepts, get_n_par(m)); }
unsigned long get_params_rejects_sum(const mcmc *m) { return sum_counters(m->params_rejects, get_n_par(m)); }
unsigned long get_params_accepts_for(const mcmc *m, const unsigned int i) { return m->params_accepts[i]; }
unsigned long get_params_rejects_for(const mcmc *m, const unsigned int i) { return m->params_rejects[i]; }
unsigned long get_params_accepts_global(const mcmc *m) { return m->accept; }
unsigned long get_params_rejects_global(const mcmc *m) { return m->reject; }

double get_accept_rate_for(const mcmc *m, const unsigned int i) {
    return m->params_accepts[i] / (double)(m->params_accepts[i] + m->params_rejects[i]);
}
double get_accept_rate_global(const mcmc *m) { return m->accept / (double)(m->accept + m->reject); }

gsl_vector *get_accept_rate(const mcmc *m) {
    const unsigned int n = get_n_par(m);
    gsl_vector *rate = gsl_vector_alloc(n);
    unsigned int i;
    for (i = 0; i < n; i++) {
        const double a = (double)m->params_accepts[i], r = (double)m->params_rejects[i];
        gsl_vector_set(rate, i, a / (r + a));
    }
    return rate;
}

const char **get_params_descr(const mcmc *m) { return m->params_descr; }

void inc_params_accepts_for(mcmc *m, const unsigned int i) { m->params_accepts[i]++; }
void inc_params_rejects_for(mcmc *m, const unsigned int i) { m->params_rejects[i]++; }
void inc_params_accepts(mcmc *m) {
    unsigned int i;
    m->accept++;
    for (i = 0; i < get_n_par(m); i++)
        m->params_accepts[i]++;
}
void inc_params_rejects(mcmc *m) {
    unsigned int i;
    m->reject++;
    for (i = 0; i < get_n_par(m); i++)
        m->params_rejects[i]++;
}
void set_params_accepts_for(mcmc *m, const long v, const unsigned int i) { m->params_accepts[i] = v; }
void set_params_rejects_for(mcmc *m, const long v, const unsigned int i) { m->params_rejects[i] = v; }
void reset_accept_rejects(mcmc *m) {
    unsigned int i;
    for (i = 0; i < get_n_par(m); i++)
        m->params_accepts[i] = m->params_rejects[i] = 0;
    m->accept = 0;
    m->reject = 0;
}

double get_prob(const mcmc *m) { return m->prob; }
double get_prior(const mcmc *m) { return m->prior; }
double get_prob_best(const mcmc *m) { return m->prob_best; }
void set_prob(mcmc *m, const double v) { m->prob = v; }
void set_prior(mcmc *m, const double v) { m->prior = v; }
void set_prob_best(mcmc *m, const double v) { m->prob_best = v; }

#ifndef N_PARAMETERS
unsigned int get_n_par(const mcmc *m) { return m->n_par; }
#endif

gsl_vector *get_params(const mcmc *m) { return m->params; }
double get_params_for(const mcmc *m, const unsigned int i) { return gsl_vector_get(m->params, i); }
gsl_vector *get_params_min(const mcmc *m) { return m->params_min; }
double get_params_min_for(const mcmc *m, const unsigned int i) { return gsl_vector_get(m->params_min, i); }
gsl_vector *get_params_max(const mcmc *m) { return m->params_max; }
double get_params_max_for(const mcmc *m, const unsigned int i) { return gsl_vector_get(m->params_max, i); }
gsl_vector *get_params_best(const mcmc *m) { return m->params_best; }
double get_params_best_for(const mcmc *m, const unsigned int i) { return gsl_vector_get(m->params_best, i); }
gsl_vector *get_steps(const mcmc *m) { return m->params_step; }
double get_steps_for(const mcmc *m, const unsigned int i) { return gsl_vector_get(m->params_step, i); }
double get_steps_for_normalized(const mcmc *m, const unsigned int i) {
    return get_steps_for(m, i) / (get_params_max_for(m, i) - get_params_min_for(m, i));
}
const gsl_matrix *get_data(const mcmc *m) { return m->data; }
gsl_rng *get_random(const mcmc *m) { return m->random; }

void set_minmax_for(mcmc *m, const double lo, const double hi, const unsigned int i) {
    gsl_vector_set(m->params_min, i, lo);
    gsl_vector_set(m->params_max, i, hi);
}
void set_steps_for(mcmc *m, const double step, const unsigned int i) { gsl_vector_set(m->params_step, i, step); }
void set_steps_for_normalized(mcmc *m, const double step, const unsigned int i) {
    gsl_vector_set(m->params_step, i, step * (get_params_max_for(m, i) - get_params_min_for(m, i)));
}
void set_steps_all(mcmc *m, const double *steps) {
    unsigned int i;
    for (i = 0; i < get_n_par(m); i++)
        set_steps_for(m, steps[i], i);
}
void set_params_best(mcmc *m, const gsl_vector *v) { gsl_vector_memcpy(m->params_best, v); }
void set_params_for(mcmc *m, const double v, const unsigned int i) {
    assert(i < m->n_par);
    gsl_vector_set(m->params, i, v);
}
void set_params(mcmc *m, gsl_vector *v) {
    assert(m->n_par == v->size);
    gsl_vector_free(m->params);
    m->params = v;
}
void set_params_descr_all(mcmc *m, const char **d) { m->params_descr = d; }
void set_params_descr_for(mcmc *m, const char *d, const unsigned int i) { m->params_descr[i] = d; }
void set_random(mcmc *m, gsl_rng *r) { m->random = r; }
void set_data(mcmc *m, const gsl_matrix *d) { m->data = d; }

double get_next_uniform_random(const mcmc *m) { return gsl_rng_uniform(get_random(m)); }
double get_next_uniform_plusminus_random(const mcmc *m) { return 2 * get_next_uniform_random(m) - 1; }
double get_next_alog_urandom(const mcmc *m) { return gsl_sf_log(get_next_uniform_random(m)); }
double get_next_random_jump(const mcmc *m, const double sigma) {
#ifdef PROPOSAL_LOGISTIC
    return gsl_ran_logistic(get_random(m), sigma);
#elif defined PROPOSAL_UNIFORM
    return gsl_ran_flat(get_random(m), -sigma, sigma);
#else
    return gsl_ran_gaussian(get_random(m), sigma);
#endif
}
