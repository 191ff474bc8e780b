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

/* Compile-time variants of the reference.  -DPROPOSAL_LOGISTIC, -DPROPOSAL_UNIFORM
 * (src/mcmc_gettersetter.c:290-305), -DRANDOMSWAP (src/parallel_tempering_interaction.c:130-131)
 * and -DADAPT, -DRWM (src/parallel_tempering.c:268-301) are carried to the device engine as
 * apemost_hip_config.flags (src/apemost_bridge.c).  What the engine does not implement refuses
 * to build rather than silently sample something else. */
#if defined(PROPOSAL_LOGISTIC) && defined(PROPOSAL_UNIFORM)
#error "PROPOSAL_LOGISTIC and PROPOSAL_UNIFORM are alternatives"
#endif
/* -DRWM (src/parallel_tempering.c:268-281, src/markov_chain.c:342-367): carried as APEMOST_HIP_FLAG_RWM since
 * round 4 -- the reference's own call site does not compile (a two-argument call of markov_chain_step);
 * include/apemost_hip.h states the semantics the engine gives it.  MINIMAL_STEPWIDTH / MAXIMAL_STEPWIDTH keep
 * their defaults on the device. */
#if defined(RWM) && (defined(MINIMAL_STEPWIDTH) || defined(MAXIMAL_STEPWIDTH))
#error "RWM: the device engine clamps step widths to the reference's default [1e-7, 1e6] x range"
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
