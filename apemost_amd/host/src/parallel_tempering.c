/* The phases an application main calls: calibrate_first, calibrate_rest, run
 * (behaviour and file formats of reference src/parallel_tempering.c:36-419).  Every
 * Metropolis step, calibration sweep and swap attempt runs on the MI355X engine; this file
 * only moves state between the files, the host chain objects and the device, and writes
 * the reference's text outputs. */
#include <math.h>
#include <string.h>
#include "mcmc.h"
#include "parallel_tempering.h"
#include "parallel_tempering_beta.h"
#include "parallel_tempering_interaction.h"
#include "parallel_tempering_config.h"
#include "parallel_tempering_run.h"
#include "apemost_bridge.h"
#include "debug.h"
#include "define_defaults.h"
#include "gsl_helper.h"
#include "utils.h"

static void fill_calib(apemost_hip_calib_config *c) {
    apemost_hip_calib_defaults(c);
    c->burn_in_iterations = BURN_IN_ITERATIONS;
    c->iter_limit = ITER_LIMIT;
    c->iter_readjust = ITER_READJUST;
    c->no_rescaling_limit = NO_RESCALING_LIMIT;
    c->rat_limit = TARGET_ACCEPTANCE_RATE;
    c->target_global = TARGET_ACCEPTANCE_RATE;
    c->max_ar_deviation = MAX_AR_DEVIATION;
    c->mul = MUL;
    c->adjust_step = DEFAULT_ADJUST_STEP;
}

/* markov_chain_calibrate() for chains [first, first+count) of a resident ladder; exits like the
 * reference when a chain cannot be calibrated */
static void calibrate_on_device(apemost_ladder *l, unsigned int first, unsigned int count, int burn_in_only) {
    apemost_hip_calib_config c;
    int32_t *status = (int32_t *)calloc(count, sizeof(int32_t));
    int rc;
    unsigned int i;
    fill_calib(&c);
    rc = apemost_hip_calibrate_chains(apemost_ladder_sampler(l), (int)first, (int)count, &c, burn_in_only, status,
                                      NULL);
    if (rc == APEMOST_HIP_ERR_CALIBRATION) {
        for (i = 0; i < count; i++)
            if (status[i] == 1)
                fprintf(stderr, "calibration failed: a step width of chain %u became too large.\n", first + i);
            else if (status[i] == 2)
                fprintf(stderr, "calibration failed: limit of %d iterations reached (chain %u).\n", ITER_LIMIT,
                        first + i);
        exit(1);
    }
    apemost_hip_or_die(rc, "markov_chain_calibrate");
    free(status);
}

static void free_chains(mcmc **chains, unsigned int n_beta) {
    unsigned int i;
    for (i = 0; i < n_beta; i++) {
        mem_free(chains[i]->additional_data);
        if (i != 0)
            set_data(chains[i], NULL); /* aliased: chain 0 frees the matrix */
        chains[i] = mcmc_free(chains[i]);
    }
    mem_free(chains);
}

/* needs: params, data.  provides: line 1 of calibration_results, params_suggested */
void calibrate_first() {
    mcmc **chains = setup_chains();
    apemost_ladder *l;
    printf("Starting markov chain calibration\n");
    fflush(stdout);
    l = apemost_ladder_open(chains, 1);
    apemost_hip_or_die(apemost_hip_calc_model(apemost_ladder_sampler(l), 0, 1), "calc_model");
    calibrate_on_device(l, 0, 1, 0);
    apemost_ladder_download(l);
    apemost_ladder_close(l);
    write_calibrations_file(chains, 1);
    write_params_file(chains[0]);
    free_chains(chains, apemost_n_beta());
}

/* start chain i of the ladder from chain 0's best point with predicted step widths
 * steps0 * beta^-1/2 (* factors) */
static void seed_chain(mcmc **chains, unsigned int i, double beta, const gsl_vector *factors) {
    set_beta(chains[i], beta);
    gsl_vector_memcpy(get_steps(chains[i]), get_steps(chains[0]));
    gsl_vector_scale(get_steps(chains[i]), pow(beta, -0.5));
    if (factors != NULL)
        gsl_vector_mul(get_steps(chains[i]), factors);
    set_params(chains[i], dup_vector(get_params_best(chains[0])));
}

/* needs: params, data, line 1 of calibration_results.  provides: calibration_results for the
 * whole ladder, calibration_summary */
void calibrate_rest() {
    const unsigned int n_beta = apemost_n_beta();
    double beta_0 = BETA_0;
    mcmc **chains = setup_chains();
    const unsigned int n_par = get_n_par(chains[0]);
    gsl_vector *factors = gsl_vector_alloc(n_par);
    apemost_ladder *l;
    unsigned int i;

    read_calibration_file(chains, 1);
    printf("Calibrating chains\n");
    fflush(stdout);
    gsl_vector_set_all(factors, 1);

    if (n_beta > 1) {
        /* the second chain alone first: its calibrated widths against the beta^-1/2 prediction
         * give the per-parameter stepwidth factors */
        const double b1 = get_chain_beta(1, n_beta, beta_0 < 0 ? calc_beta_0(chains[0], factors) : beta_0);
        seed_chain(chains, 1, b1, NULL);
        printf("Calibrating second chain to infer stepwidth factor\n");
        printf("\tChain %2d - beta = %f\tsteps: ", 1, get_beta(chains[1]));
        dump_vectorln(get_steps(chains[1]));
        fflush(stdout);
        l = apemost_ladder_open(chains, n_beta);
        apemost_hip_or_die(apemost_hip_calc_model(apemost_ladder_sampler(l), 1, 1), "calc_model");
        calibrate_on_device(l, 1, 1, 0);
        apemost_ladder_download(l);
        apemost_ladder_close(l);
        gsl_vector_scale(factors, pow(get_beta(chains[1]), -0.5));
        gsl_vector_mul(factors, get_steps(chains[0]));
        gsl_vector_div(factors, get_steps(chains[1]));
    }
    printf("stepwidth factors: ");
    dump_vectorln(factors);
    if (beta_0 < 0) {
        beta_0 = calc_beta_0(chains[0], factors);
        printf("automatic beta_0: %f\n", beta_0);
    }
    fflush(stdout);

    if (n_beta > 1) {
        for (i = 1; i < n_beta; i++) {
            seed_chain(chains, i, get_chain_beta(i, n_beta, beta_0), factors);
            if (n_beta <= 64) {
                printf("\tChain %2d - beta = %f\tsteps: ", i, get_beta(chains[i]));
                dump_vectorln(get_steps(chains[i]));
            }
        }
        fflush(stdout);
        /* all remaining chains calibrate concurrently, one workgroup each */
        l = apemost_ladder_open(chains, n_beta);
        apemost_hip_or_die(apemost_hip_calc_model(apemost_ladder_sampler(l), 1, (int)n_beta - 1), "calc_model");
#ifndef SKIP_CALIBRATE_ALLCHAINS
        calibrate_on_device(l, 1, n_beta - 1, 0);
#else
        calibrate_on_device(l, 1, n_beta - 1, 1);
#endif
        apemost_ladder_download(l);
        apemost_ladder_close(l);
    }
    gsl_vector_free(factors);
    printf("all chains calibrated.\n");
    if (n_beta <= 64)
        for (i = 0; i < n_beta; i++) {
            printf("\tChain %2d - beta = %f \tsteps: ", i, get_beta(chains[i]));
            dump_vectorln(get_steps(chains[i]));
        }
    write_calibration_summary(chains, n_beta);
    write_calibrations_file(chains, n_beta);
    free_chains(chains, n_beta);
}

static void report(const mcmc **chains, const int n_beta) {
    int i;
    print_current_positions(chains, n_beta);
    printf("\nwriting out visited parameters ");
    for (i = 0; i < n_beta; i++) {
        printf(".");
        mcmc_dump_flush(chains[i]);
        fflush(stdout);
#ifndef DUMP_ALL_CHAINS
        break;
#endif
    }
    printf("done.\n");
}

static unsigned long gcd_ul(unsigned long a, unsigned long b) {
    while (b) {
        unsigned long t = a % b;
        a = b;
        b = t;
    }
    return a;
}

/* The sampler loop: batches of rounds on the device, then the batch's sample rows are
 * written in the reference's formats: <name>-chain-<i>.prob.dump (chain 0, or all with
 * -DDUMP_ALL_CHAINS), prob-chain<i>.dump, acceptance_rate.dump(.gnuplot), progress line. */
static void run_sampler(mcmc **chains, const unsigned int n_beta, const unsigned int n_swap,
                        const unsigned long max_iterations, char *mode) {
    const unsigned int n_par = get_n_par(chains[0]);
    const size_t row = (size_t)n_beta * (n_par + 2);
    unsigned long iter = chains[0]->n_iter;
    /* rounds between two acceptance lines; batches never cross such a point */
    const unsigned long interval_rounds = PRINT_PROB_INTERVAL / gcd_ul(PRINT_PROB_INTERVAL, n_swap);
    unsigned long max_rounds = (unsigned long)(((size_t)64 << 20) / (row * n_swap * sizeof(double)));
    FILE **prob_files = (FILE **)mem_calloc(n_beta, sizeof(FILE *));
    FILE *acceptance_file;
    apemost_ladder *l;
    apemost_hip_sampler *s;
    double *d_samples = NULL, *h_samples;
    char name[100];
    unsigned int i, p;

    if (max_rounds < 1)
        max_rounds = 1;
    if (max_rounds > interval_rounds)
        max_rounds = interval_rounds;
    for (i = 0; i < n_beta; i++) {
        sprintf(name, "prob-chain%d.dump", i);
        prob_files[i] = fopen(name, mode);
        if (prob_files[i] == NULL) {
            fprintf(stderr, "opening file %s failed\n", name);
            perror("opening file failed");
            exit(1);
        }
    }
    acceptance_file = fopen("acceptance_rate.dump.gnuplot", "w");
    if (acceptance_file != NULL) {
        fprintf(acceptance_file, "# format: iteration | number of accepts for each chain\nplot ");
        for (i = 0; i < n_beta; i++)
            fprintf(acceptance_file, "\"acceptance_rate.dump\" u 1:%d title \"chain %d, beta = %f\"%s", i + 2, i,
                    get_beta(chains[i]), i + 1 != n_beta ? ", " : "");
        fprintf(acceptance_file, "\n");
        fclose(acceptance_file);
    }
    acceptance_file = fopen("acceptance_rate.dump", mode);
    assert(acceptance_file != NULL);

    l = apemost_ladder_open(chains, n_beta);
    s = apemost_ladder_sampler(l);
    apemost_hip_or_die(apemost_hip_samples_alloc(s, max_rounds * n_swap, &d_samples), "samples_alloc");
    h_samples = (double *)malloc(max_rounds * n_swap * row * sizeof(double));
    assert(h_samples != NULL);
    get_duration();
    run = 1;
    dumpflag = 0;
    printf("starting the analysis\n");
    fflush(stdout);

    while (run && (max_iterations == 0 || iter < max_iterations)) {
        const unsigned long done_rounds = iter / n_swap;
        unsigned long rounds = interval_rounds - done_rounds % interval_rounds, step;
        if (rounds > max_rounds)
            rounds = max_rounds;
        if (max_iterations != 0) {
            const unsigned long left = (max_iterations - iter + n_swap - 1) / n_swap;
            if (rounds > left)
                rounds = left;
        }
        apemost_hip_or_die(apemost_hip_run(s, rounds, n_swap, d_samples), "run_sampler");
        apemost_hip_or_die(apemost_hip_samples_read(s, d_samples, rounds * n_swap, h_samples), "samples_read");
        for (step = 0; step < rounds * n_swap; step++) {
            for (i = 0; i < n_beta; i++) {
                const double *r = h_samples + step * row + (size_t)i * (n_par + 2);
                if (chains[i]->files != NULL)
                    for (p = 0; p < n_par; p++)
                        if (chains[i]->files[p] != NULL)
                            fprintf(chains[i]->files[p], DUMP_FORMAT "\n", r[p]);
                fprintf(prob_files[i], "%6e\t%6e\n", r[n_par], r[n_par + 1]);
            }
        }
        iter += rounds * n_swap;
        apemost_swap_round += rounds;
        if (iter % PRINT_PROB_INTERVAL == 0) {
            apemost_ladder_download(l);
            if (dumpflag) {
                report((const mcmc **)chains, (int)n_beta);
                dumpflag = 0;
                for (i = 0; i < n_beta; i++)
                    fflush(prob_files[i]);
            }
            fprintf(acceptance_file, "%lu", iter);
            for (i = 0; i < n_beta; i++)
                fprintf(acceptance_file, "\t%lu", get_params_accepts_global(chains[i]));
            fprintf(acceptance_file, "\n");
            fflush(acceptance_file);
            printf("iteration: %lu, a/r: %.3f(%lu/%lu), v:", iter,
                   (double)get_params_accepts_global(chains[0]) /
                       (double)(get_params_accepts_global(chains[0]) + get_params_rejects_global(chains[0])),
                   get_params_accepts_global(chains[0]), get_params_rejects_global(chains[0]));
            dump_vector(get_params(chains[0]));
            printf(" [%d/%lu ticks]\r", get_duration(), get_ticks_per_second());
            fflush(stdout);
        }
    }
    apemost_ladder_download(l);
    apemost_hip_samples_free(s, d_samples);
    apemost_ladder_close(l);
    free(h_samples);
    fclose(acceptance_file);
    for (i = 0; i < n_beta; i++)
        fclose(prob_files[i]);
    mem_free(prob_files);
    printf("handled %lu iterations on %d chains\n", iter, n_beta);
}

void prepare_and_run_sampler(const unsigned long max_iterations, int append) {
    const unsigned int n_beta = apemost_n_beta();
    int n_swap = N_SWAP;
    char *mode = (append == 1 ? "a" : "w");
    mcmc **chains = setup_chains();
#ifdef DUMP_ALL_CHAINS
    unsigned int i;
#endif
    read_calibration_file(chains, n_beta);
    mcmc_open_dump_files(chains[0], "-chain", 0, mode);
#ifdef DUMP_ALL_CHAINS
    for (i = 1; i < n_beta; i++)
        mcmc_open_dump_files(chains[i], "-chain", i, mode);
#endif
    if (n_swap < 0) {
        n_swap = 2000 / n_beta;
        printf("automatic n_swap: %d\n", n_swap);
    }
    if (n_swap < 1) {
        /* the reference's rule yields 0 beyond 2000 chains and then never advances */
        fprintf(stderr, "n_swap = %d: set -DN_SWAP to a positive value for ladders of more than 2000 chains\n",
                n_swap);
        exit(1);
    }
    register_signal_handlers();
    run_sampler(chains, n_beta, (unsigned int)n_swap, max_iterations, mode);
    report((const mcmc **)chains, (int)n_beta);
    free_chains(chains, n_beta);
}
