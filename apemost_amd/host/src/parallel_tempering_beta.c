/* Tempering data of a chain and the beta ladders (reference
 * src/parallel_tempering_beta.c:25-102). */
#include <math.h>
#include "mcmc.h"
#include "gsl_helper.h"
#include "debug.h"
#include "parallel_tempering_beta.h"
#include "define_defaults.h"

static parallel_tempering_mcmc *pt(const mcmc *m) { return (parallel_tempering_mcmc *)m->additional_data; }

void set_beta(mcmc *m, const double newbeta) {
    pt(m)->beta = newbeta;
    pt(m)->swapcount = 0;
}
double get_beta(const mcmc *m) { return pt(m)->beta; }
void inc_swapcount(mcmc *m) { pt(m)->swapcount++; }
unsigned long get_swapcount(const mcmc *m) { return pt(m)->swapcount; }

/* one block per chain: how often it swapped, where it is, the best point it has seen */
static void print_point(int chain, const char *label, double prob, const gsl_vector *point) {
    printf("\tchain %d: %s %f: ", chain, label, prob);
    dump_vectorln(point);
}

void print_current_positions(const mcmc **chains, const int n_beta) {
    int chain;
    printf("printing chain parameters: \n");
    for (chain = 0; chain < n_beta; chain++) {
        const mcmc *m = chains[chain];
        printf("\tchain %d: swapped %lu times: ", chain, get_swapcount(m));
        print_point(chain, "current", get_prob(m), get_params(m));
        print_point(chain, "best", get_prob_best(m), get_params_best(m));
    }
    fflush(stdout);
}

/* position i of n_beta between beta_0 (i = 0) and 1 (i = n_beta-1) */
double equidistant_beta(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return beta_0 + i * (1 - beta_0) / (n_beta - 1);
}
double equidistant_temperature(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return 1 / (1 / beta_0 + i * (1 - 1 / beta_0) / (n_beta - 1));
}
double chebyshev_temperature(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return 1 / (1 / beta_0 + (1 - 1 / beta_0) / 2 * (1 - cos(i * M_PI / (n_beta - 1))));
}
double chebyshev_beta(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return beta_0 + (1 - beta_0) / 2 * (1 - cos(i * M_PI / (n_beta - 1)));
}
double equidistant_stepwidth(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return beta_0 + pow(i * 1.0 / (n_beta - 1), 2) * (1 - beta_0);
}
double chebyshev_stepwidth(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return beta_0 + (1 - beta_0) * pow((1 - cos(i * M_PI / (n_beta - 1))) / 2, 2);
}
double hot_chains(const unsigned int i, const unsigned int n_beta, const double beta_0) {
    return beta_0 + 0 * i * n_beta;
}

/* chain 0 is the cold one (beta = 1) */
double get_chain_beta(unsigned int i, unsigned int n_beta, double beta_0) {
    if (n_beta == 1)
        return 1.0;
    return BETA_ALIGNMENT(n_beta - i - 1, n_beta, beta_0);
}

/* beta of the hottest chain: the one at which the predicted proposal width of the parameter that
 * fills its range soonest, step * beta^-1/2 * factor, reaches BETA_0_STEPWIDTH times that range:
 * beta_0 = max_p( BETA_0_STEPWIDTH * (max_p - min_p) / (step_p * factor_p) )^-1/2.  Same operations
 * on every component as the vector arithmetic of the reference (range scaled, divided by the step,
 * divided by the factor), without the temporary. */
double calc_beta_0(mcmc *m, gsl_vector *stepwidth_factors) {
    const unsigned int n = get_n_par(m);
    double widest = 0;
    unsigned int p;
    for (p = 0; p < n; p++) {
        double reach = get_params_max_for(m, p) - get_params_min_for(m, p);
        reach = reach * BETA_0_STEPWIDTH;
        reach = reach / get_steps_for(m, p);
        reach = reach / gsl_vector_get(stepwidth_factors, p);
        if (p == 0 || reach > widest)
            widest = reach;
    }
    return pow(widest, -0.5);
}
