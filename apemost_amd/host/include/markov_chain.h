/* Single-chain Metropolis operators (reference src/markov_chain.h:59-116).  In this
 * build they run on the MI355X engine through the bridge (src/apemost_bridge.c). */
#ifndef MCMC_MARKOV_CHAIN_H_
#define MCMC_MARKOV_CHAIN_H_

#include "mcmc.h"
#include "define_defaults.h"

#define DEFAULT_ADJUST_STEP 0.5
#ifndef NO_RESCALING_LIMIT
#define NO_RESCALING_LIMIT 15
#endif
#ifndef ITER_READJUST
#define ITER_READJUST 200
#endif
#ifndef CIRCULAR_PARAMS
#define CIRCULAR_PARAMS 0
#endif

/* Compile-time variants of the reference that the device engine does not implement: refuse to
 * build rather than silently sample something else. */
#ifdef PROPOSAL_LOGISTIC
#error "PROPOSAL_LOGISTIC: the MI355X engine implements the default Gaussian proposal only"
#endif
#ifdef PROPOSAL_UNIFORM
#error "PROPOSAL_UNIFORM: the MI355X engine implements the default Gaussian proposal only"
#endif
#ifdef RANDOMSWAP
#error "RANDOMSWAP: the MI355X engine implements the default swap schedule (decide_swap_now) only"
#endif
#if defined(RWM) || defined(ADAPT)
#error "RWM / ADAPT: adaptive step widths during the run are not implemented by the MI355X engine"
#endif
#if defined(CALIBRATE_MULTILIN) || defined(CALIBRATE_QUADRATIC) || defined(CALIBRATE_ALTERNATE)
#error "alternate calibrators are not implemented by the MI355X engine (default markov_chain_calibrate_orig only)"
#endif
#ifndef ACCURACY_DEVIATION_FACTOR
#define ACCURACY_DEVIATION_FACTOR 0.25
#endif

/* burn-in, then step-width calibration towards the acceptance rate */
void markov_chain_calibrate(mcmc *m, const unsigned int burn_in_iterations, double desired_acceptance_rate,
                            const double max_ar_deviation, const unsigned int iter_limit, double mul,
                            const double adjust_step);
void markov_chain_step(mcmc *m);
void markov_chain_step_for(mcmc *m, const unsigned int index);
void rmw_adapt_stepwidth(mcmc *m, double prob_old);
void burn_in(mcmc *m, const unsigned int burn_in_iterations);
unsigned int assess_acceptance_rate(mcmc *m, unsigned int param, double desired_acceptance_rate,
                                    double min_accuracy, double max_accuracy, double *acceptance_rate,
                                    double *accuracy);
void restart_from_best(mcmc *m);

#endif
