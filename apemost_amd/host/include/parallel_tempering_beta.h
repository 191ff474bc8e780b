/* Temperature ladder and per-chain tempering data (reference
 * src/parallel_tempering_beta.h:27-84). */
#ifndef _PARALLEL_TEMPERING_BETA_H
#define _PARALLEL_TEMPERING_BETA_H
#include <gsl/gsl_sf.h>

#include "mcmc.h"
#include "parallel_tempering.h"
#include "parallel_tempering_interaction.h"

/* BETA_ALIGNMENT: one of equidistant_beta, equidistant_temperature, chebyshev_beta,
 * chebyshev_temperature, equidistant_stepwidth, chebyshev_stepwidth, hot_chains */
#ifndef BETA_ALIGNMENT
#define BETA_ALIGNMENT chebyshev_beta
#endif
#define BETA_0_STEPWIDTH 1.0

/* exactly the reference's two fields (src/parallel_tempering_beta.h:65-76): applications allocate
 * this struct themselves (mem_malloc(sizeof(parallel_tempering_mcmc)), apps/eval_main.c:50), so its
 * size is part of the binary contract.  What the engine needs per chain beyond it -- the chain's
 * address in the device RNG streams -- lives in a side table of the bridge, keyed by the mcmc
 * pointer (apemost_bridge.h: apemost_chain_address). */
typedef struct {
    double beta;             /* inverse temperature */
    unsigned long swapcount; /* accepted swaps with the next-hotter chain */
} parallel_tempering_mcmc;

void set_beta(mcmc *m, const double newbeta);
double get_beta(const mcmc *m);
void inc_swapcount(mcmc *m);
unsigned long get_swapcount(const mcmc *m);
void print_current_positions(const mcmc **chains, const int n_beta);
double get_chain_beta(unsigned int i, unsigned int n_beta, double beta_0);
double calc_beta_0(mcmc *m, gsl_vector *stepwidth_factors);

double equidistant_beta(const unsigned int i, const unsigned int n_beta, const double beta_0);
double equidistant_temperature(const unsigned int i, const unsigned int n_beta, const double beta_0);
double chebyshev_temperature(const unsigned int i, const unsigned int n_beta, const double beta_0);
double chebyshev_beta(const unsigned int i, const unsigned int n_beta, const double beta_0);
double equidistant_stepwidth(const unsigned int i, const unsigned int n_beta, const double beta_0);
double chebyshev_stepwidth(const unsigned int i, const unsigned int n_beta, const double beta_0);
double hot_chains(const unsigned int i, const unsigned int n_beta, const double beta_0);

#endif
