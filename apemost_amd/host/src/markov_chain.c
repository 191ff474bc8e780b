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

/* n Metropolis updates of one kind in ONE launch (which < 0: markov_chain_step, else
 * markov_chain_step_for(m, which)), each followed by mcmc_check_best as the round kernel does it;
 * accepted[k] = step k moved the chain (its sample row differs from the one before: a proposal is a
 * continuous draw, so an accepted step never lands on the point it started from).  One upload, one
 * launch and one download for the whole batch instead of one of each per step. */
static void steps_on_device(mcmc *m, int which, unsigned int n, unsigned char *accepted) {
    apemost_ladder *l;
    apemost_hip_sampler *s;
    const unsigned int n_par = get_n_par(m);
    const unsigned long n_iter = m->n_iter;
    const size_t row = n_par + 2;
    double *d_rows = NULL, *rows = (double *)malloc((size_t)(n + 1) * row * sizeof(double));
    unsigned int k, p;
    assert(rows != NULL);
    mcmc_check(m);
    for (p = 0; p < n_par; p++)
        rows[p] = gsl_vector_get(m->params, p); /* the point the first step starts from */
    l = apemost_single(m);
    s = apemost_ladder_sampler(l);
    apemost_ladder_upload(l);
    apemost_hip_or_die(apemost_hip_samples_alloc(s, n, &d_rows), "samples_alloc");
    if (which < 0)
        apemost_hip_or_die(apemost_hip_launch_round(s, n, 0, d_rows), "markov_chain_step");
    else
        apemost_hip_or_die(apemost_hip_launch_round_for(s, n, which, d_rows), "markov_chain_step_for");
    apemost_hip_or_die(apemost_hip_samples_read(s, d_rows, n, rows + row), "samples_read");
    apemost_hip_or_die(apemost_hip_samples_free(s, d_rows), "samples_free");
    apemost_ladder_download(l);
    m->n_iter = n_iter; /* (the steps of this API do not count as iterations) */
    for (k = 0; k < n; k++) {
        const double *was = rows + (size_t)k * row, *is = was + row;
        accepted[k] = 0;
        for (p = 0; p < n_par; p++)
            if (is[p] != was[p])
                accepted[k] = 1;
    }
    free(rows);
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
    c.progress_chain = 0;
    mcmc_check(m);
    apemost_ladder_upload(l);
    rc = apemost_hip_calibrate_chains(apemost_ladder_sampler(l), 0, 1, &c, 0, &status, &iters);
    apemost_write_calibration_progress(apemost_ladder_sampler(l), get_n_par(m)); /* markov_chain_calibrate.c:1052, 1141-1146 */
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

/* Step-width adaptation of the reference's (non-default, there uncompilable) -DRWM variant,
 * src/markov_chain.c:342-367: every width moves by a random fraction of its parameter range,
 * towards wider steps when the last move was accepted more readily than TARGET_ACCEPTANCE_RATE,
 * towards narrower ones otherwise; the pull fades as 1/sqrt(n_iter).  Host arithmetic only. */
#ifndef MINIMAL_STEPWIDTH
#define MINIMAL_STEPWIDTH 0.0000001
#endif
#ifndef MAXIMAL_STEPWIDTH
#define MAXIMAL_STEPWIDTH 1000000
#endif
void rmw_adapt_stepwidth(mcmc *m, const double prob_old) {
    const double ratio = exp(get_prob(m) - prob_old);
    const double excess = (ratio < 1 ? ratio : 1) - TARGET_ACCEPTANCE_RATE;
    unsigned int p;
    for (p = 0; p < get_n_par(m); p++) {
        const double range = get_params_max_for(m, p) - get_params_min_for(m, p);
        const double narrowest = MINIMAL_STEPWIDTH * range, widest = MAXIMAL_STEPWIDTH * range;
        /* the reference's association, U / sqrt(n) * excess * range: the device kernel and the oracle round alike */
        double width = get_steps_for(m, p) + get_next_uniform_random(m) / sqrt(m->n_iter) * excess * range;
        width = width < narrowest ? narrowest : width;
        width = width > widest ? widest : width;
        set_steps_for(m, width, p);
    }
}

/* assess_acceptance_rate (reference src/markov_chain.c:117-224; used by the alternate calibrators
 * and by applications that tune widths themselves): measure the acceptance rate of parameter
 * `param` (or of the all-parameter step when param >= n_par) to an accuracy that tightens as the
 * rate approaches the desired one.  Every step is a device step (markov_chain_step[_for]), a batch
 * of them per launch.
 *
 * The estimate follows the reference to the letter, including two things a reader might not
 * expect: the rate is (accepts before the LAST step of the batch) / n, and the drift of the
 * running accept count around rate * j is truncated to an integer before it is compared.
 * Returns the number of steps used. */
unsigned int assess_acceptance_rate(mcmc *m, unsigned int param, double desired_acceptance_rate,
                                    double min_accuracy, double max_accuracy, double *acceptance_rate,
                                    double *accuracy) {
    const int single = param < get_n_par(m);
    unsigned int done = 0, n = 40, j;
    unsigned char *accepted = NULL; /* one flag per step, kept over all batches */
    reset_accept_rejects(m);
    for (;;) {
        unsigned long before = 0, running = 0;
        unsigned int drift = 1;
        double rate, wanted;
        accepted = (unsigned char *)realloc(accepted, n);
        assert(accepted != NULL);
        if (done < n) {
            /* the steps done+1 .. n in one launch; `before` = the count ahead of the last one */
            steps_on_device(m, single ? (int)param : -1, n - done, accepted + done);
            done = n;
            before = (single ? get_params_accepts_for(m, param) : get_params_accepts_global(m)) - accepted[n - 1];
        }
        rate = before / (double)n;
        for (j = 0; j < n; j++) {
            int off;
            running += accepted[j];
            off = (int)(running - rate * j);
            if (off < 0)
                off = -off;
            if ((unsigned int)off > drift)
                drift = (unsigned int)off;
        }
        wanted = (rate < desired_acceptance_rate ? desired_acceptance_rate - rate : rate - desired_acceptance_rate) *
                 ACCURACY_DEVIATION_FACTOR;
        if (wanted < 0.005)
            wanted = 0.005;
        if (wanted < min_accuracy)
            wanted = min_accuracy;
        if (wanted > max_accuracy)
            wanted = max_accuracy;
        *acceptance_rate = rate;
        *accuracy = drift / 1. / n;
        if (*accuracy <= wanted)
            break;
        assert(drift / wanted >= n);
        n = ((unsigned int)((drift / 1. / wanted) / 8) + 1) * 8; /* enough steps for that drift to weigh `wanted` */
    }
    free(accepted);
    return n;
}
