/* Single-chain operators of the reference API (src/markov_chain.c:29-386,
 * src/markov_chain_calibrate.c:1182-1204), executed by the device engine: the chain is
 * mirrored into a one-chain device ladder, the kernel runs, the chain is read back.
 * The bulk phases (calibrate_*, run) never come through here; they keep the whole ladder
 * resident (parallel_tempering.c). */
#include <math.h>
#include "mcmc.h"
#include "mcmc_internal.h"
#include "apemost_bridge.h"
#include "gsl_helper.h"
#include "debug.h"

void restart_from_best(mcmc *m) {
    set_params(m, dup_vector(get_params_best(m)));
    set_prob(m, get_prob_best(m));
}

static void fill_calib(apemost_hip_calib_config *c, unsigned int burn_in_iterations, double desired,
                       double max_ar_deviation, unsigned int iter_limit, double mul, double adjust_step) {
    apemost_hip_calib_defaults(c);
    c->burn_in_iterations = burn_in_iterations;
    c->iter_limit = iter_limit;
    c->iter_readjust = ITER_READJUST;
    c->no_rescaling_limit = NO_RESCALING_LIMIT;
    c->rat_limit = desired;
    c->target_global = TARGET_ACCEPTANCE_RATE;
    c->max_ar_deviation = max_ar_deviation;
    c->mul = mul;
    c->adjust_step = adjust_step;
}

/* one all-parameter Metropolis update: proposal with redraw at the bounds, likelihood,
 * accept test, counters.  n_iter and prob_best are left to the caller, as in the reference. */
void markov_chain_step(mcmc *m) {
    apemost_ladder *l;
    const unsigned long n_iter = m->n_iter;
    const double prob_best = m->prob_best;
    gsl_vector *best = dup_vector(m->params_best);
    mcmc_check(m);
    l = apemost_single(m);
    apemost_ladder_upload(l);
    apemost_hip_or_die(apemost_hip_launch_round(apemost_ladder_sampler(l), 1, 0, NULL), "markov_chain_step");
    apemost_ladder_download(l);
    /* the round kernel also did check_best and n_iter++; this API call does neither */
    m->n_iter = n_iter;
    m->prob_best = prob_best;
    gsl_vector_memcpy(m->params_best, best);
    gsl_vector_free(best);
}

/* one single-parameter update (only parameter `index` is proposed and counted) */
void markov_chain_step_for(mcmc *m, const unsigned int index) {
    apemost_ladder *l;
    const unsigned long n_iter = m->n_iter;
    const double prob_best = m->prob_best;
    gsl_vector *best = dup_vector(m->params_best);
    mcmc_check(m);
    l = apemost_single(m);
    apemost_ladder_upload(l);
    apemost_hip_or_die(apemost_hip_launch_round_for(apemost_ladder_sampler(l), 1, (int)index, NULL),
                       "markov_chain_step_for");
    apemost_ladder_download(l);
    m->n_iter = n_iter;
    m->prob_best = prob_best;
    gsl_vector_memcpy(m->params_best, best);
    gsl_vector_free(best);
}

void burn_in(mcmc *m, const unsigned int burn_in_iterations) {
    apemost_ladder *l = apemost_single(m);
    apemost_hip_calib_config c;
    int32_t status = 0;
    uint64_t iters = 0;
    fill_calib(&c, burn_in_iterations, TARGET_ACCEPTANCE_RATE, MAX_AR_DEVIATION, ITER_LIMIT, MUL, DEFAULT_ADJUST_STEP);
    mcmc_check(m);
    apemost_ladder_upload(l);
    apemost_hip_or_die(apemost_hip_calibrate_chains(apemost_ladder_sampler(l), 0, 1, &c, 1, &status, &iters), "burn_in");
    apemost_ladder_download(l);
}

void markov_chain_calibrate(mcmc *m, const unsigned int burn_in_iterations, double desired_acceptance_rate,
                            const double max_ar_deviation, const unsigned int iter_limit, double mul,
                            const double adjust_step) {
    apemost_ladder *l = apemost_single(m);
    apemost_hip_calib_config c;
    int32_t status = 0;
    uint64_t iters = 0;
    int rc;
    fill_calib(&c, burn_in_iterations, desired_acceptance_rate, max_ar_deviation, iter_limit, mul, adjust_step);
    mcmc_check(m);
    apemost_ladder_upload(l);
    rc = apemost_hip_calibrate_chains(apemost_ladder_sampler(l), 0, 1, &c, 0, &status, &iters);
    if (rc == APEMOST_HIP_ERR_CALIBRATION) {
        /* the reference exits here too (markov_chain_calibrate.c:1107-1109, 1169-1173) */
        if (status == 1)
            fprintf(stderr, "calibration failed: a step width became too large.\n");
        else
            fprintf(stderr, "calibration failed: limit of %u iterations reached.", iter_limit);
        exit(1);
    }
    apemost_hip_or_die(rc, "markov_chain_calibrate");
    apemost_ladder_download(l);
}

/* adaptive random-walk Metropolis of the reference (:342-367) is behind -DRWM, which does
 * not compile there (SURVEY component 3); kept as host arithmetic on the step widths */
#ifndef MINIMAL_STEPWIDTH
#define MINIMAL_STEPWIDTH 0.0000001
#endif
#ifndef MAXIMAL_STEPWIDTH
#define MAXIMAL_STEPWIDTH 1000000
#endif
void rmw_adapt_stepwidth(mcmc *m, const double prob_old) {
    unsigned int i;
    double alpha = exp(get_prob(m) - prob_old);
    if (alpha > 1)
        alpha = 1;
    for (i = 0; i < get_n_par(m); i++) {
        const double scale = get_params_max_for(m, i) - get_params_min_for(m, i);
        double step = get_steps_for(m, i);
        step += get_next_uniform_random(m) / sqrt(m->n_iter) * (alpha - TARGET_ACCEPTANCE_RATE) * scale;
        if (step < MINIMAL_STEPWIDTH * scale)
            step = MINIMAL_STEPWIDTH * scale;
        if (step > MAXIMAL_STEPWIDTH * scale)
            step = MAXIMAL_STEPWIDTH * scale;
        set_steps_for(m, step, i);
    }
}

/* only the alternate calibrators (-DCALIBRATE_*) use this; they are out of scope */
unsigned int assess_acceptance_rate(mcmc *m, unsigned int param, double desired_acceptance_rate,
                                    double min_accuracy, double max_accuracy, double *acceptance_rate,
                                    double *accuracy) {
    (void)m;
    (void)param;
    (void)desired_acceptance_rate;
    (void)min_accuracy;
    (void)max_accuracy;
    (void)acceptance_rate;
    (void)accuracy;
    fprintf(stderr, "assess_acceptance_rate: alternate calibrators are not part of this engine.\n");
    exit(1);
    return 0;
}
