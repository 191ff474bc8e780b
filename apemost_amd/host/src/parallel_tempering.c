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
    rc = apemost_ladder_calibrate(l, first, count, &c, burn_in_only, status);
    if (rc == APEMOST_HIP_ERR_CALIBRATION) {
        for (i = 0; i < count; i++)
            if (status[i] == 1)
                fprintf(stderr, "calibration failed: a step width of chain %u became too large.\n", first + i);
            else if (status[i] == 2)
                fprintf(stderr, "calibration failed: limit of %d iterations reached (chain %u).\n", ITER_LIMIT,
                        first + i);
        exit(1);
    }
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
    apemost_ladder_calc_model(l, 0, 1);
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
        apemost_ladder_calc_model(l, 1, 1);
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
        apemost_ladder_calc_model(l, 1, n_beta - 1);
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

/* ---- sample sink ------------------------------------------------------------------------
 * Where the sample rows of the run phase go.  Selected at run time by APEMOST_DUMP, a comma
 * separated list of
 *   text     (default) the reference's files: <name>-chain-<i>.prob.dump ("%.15e", chain 0 or
 *            all chains with -DDUMP_ALL_CHAINS; src/mcmc_dump.c:79-88) and prob-chain<i>.dump
 *            ("%6e\t%6e" for every chain; src/parallel_tempering.c:399-401)
 *   binary   one file samples.bin holding what the text files hold, as doubles: a 64-byte header,
 *            then per kept iteration the parameter vectors of the chains that have parameter dump
 *            files (chain 0; all chains with -DDUMP_ALL_CHAINS) followed by (prob, prob - prior)
 *            of every chain (tools/samples_bin.py reads it and expands it into the text files)
 *   binary:all   the same with the parameter vector of every chain
 *   thin:N   keep every N-th iteration only (either format)
 * The reference prints one line per chain per step, which at device speed is the whole run time
 * (SURVEY 8 f1); binary and thinned sinks are the additive options for that. */
#define SINK_MAGIC "APEMOSTB"
#ifndef PROB_FILES_OPEN_MAX
#define PROB_FILES_OPEN_MAX 256 /* beyond this many chains prob-chain files are opened per batch */
#endif

typedef struct {
    int binary;               /* 0 text, 1 binary, 2 binary with every chain's parameters */
    unsigned int n_param_chains; /* binary: chains 0..n-1 carry their parameter vectors */
    double *pack;             /* binary: one batch, packed */
    size_t pack_capacity;
    unsigned long thin;
    unsigned int n_beta, n_par;
    const char *mode;  /* "w" or "a" */
    FILE *bin;         /* binary sink */
    FILE **prob_files; /* text sink, ladders up to PROB_FILES_OPEN_MAX chains: kept open */
    int batches;       /* batches written so far (pooled text files switch to append after the first) */
} sample_sink;

static void sink_parse(sample_sink *k) {
    const char *spec = getenv("APEMOST_DUMP");
    k->binary = 0;
    k->thin = 1;
    while (spec != NULL && *spec != 0) {
        if (strncmp(spec, "binary:all", 10) == 0)
            k->binary = 2;
        else if (strncmp(spec, "binary", 6) == 0)
            k->binary = 1;
        else if (strncmp(spec, "text", 4) == 0)
            k->binary = 0;
        else if (strncmp(spec, "thin:", 5) == 0 && atol(spec + 5) > 0)
            k->thin = (unsigned long)atol(spec + 5);
        else {
            fprintf(stderr, "APEMOST_DUMP: expected a comma separated list of text, binary, binary:all, thin:N; got '%s'\n", spec);
            exit(1);
        }
        spec = strchr(spec, ',');
        if (spec != NULL)
            spec++;
    }
}

static FILE *open_or_die(const char *name, const char *mode) {
    FILE *f = fopen(name, mode);
    if (f == NULL) {
        fprintf(stderr, "opening file %s failed\n", name);
        perror("opening file failed");
        exit(1);
    }
    return f;
}

static void sink_open(sample_sink *k, mcmc **chains, unsigned int n_beta, unsigned int n_par, unsigned int n_swap,
                      const char *mode) {
    unsigned int i;
    char name[100];
    sink_parse(k);
    k->n_beta = n_beta;
    k->n_par = n_par;
    k->mode = mode;
    k->bin = NULL;
    k->prob_files = NULL;
    k->batches = 0;
    k->pack = NULL;
    k->pack_capacity = 0;
    k->n_param_chains = 0;
    if (k->binary) {
        unsigned char header[64];
        uint32_t u32[4], u32b;
        uint64_t u64v = k->thin;
        FILE *probe = mode[0] == 'w' ? NULL : fopen("samples.bin", "rb");
        const int fresh = probe == NULL;
        /* the chains the text sink would write parameter files for come first in the ladder */
        while (k->n_param_chains < n_beta && chains[k->n_param_chains]->files != NULL)
            k->n_param_chains++;
        if (k->binary == 2)
            k->n_param_chains = n_beta;
        u32b = k->n_param_chains;
        if (probe != NULL)
            fclose(probe);
        k->bin = open_or_die("samples.bin", fresh ? "wb" : "ab");
        if (fresh) {
            memset(header, 0, sizeof header);
            memcpy(header, SINK_MAGIC, 8);
            u32[0] = 2; /* format version */
            u32[1] = n_beta;
            u32[2] = n_par;
            u32[3] = n_swap;
            memcpy(header + 8, u32, sizeof u32);
            memcpy(header + 24, &u64v, sizeof u64v);
            memcpy(header + 32, &u32b, sizeof u32b);
            fwrite(header, 1, sizeof header, k->bin);
        }
        return;
    }
    if (n_beta <= PROB_FILES_OPEN_MAX) {
        k->prob_files = (FILE **)mem_calloc(n_beta, sizeof(FILE *));
        for (i = 0; i < n_beta; i++) {
            sprintf(name, "prob-chain%d.dump", i);
            k->prob_files[i] = open_or_die(name, mode);
        }
    }
}

/* rows of iterations first+1 .. first+n_steps.  The rows arrive per shard: h[j] =
 * [n_steps][chains of shard j][n_par+2], shard j holding chains [lo[j], lo[j+1]) */
static void sink_write(sample_sink *k, mcmc **chains, double *const *h, const unsigned int *lo, unsigned int n_shards,
                       unsigned long first, unsigned long n_steps) {
    const unsigned int n_par = k->n_par;
    /* first kept step of this batch: iteration numbers count from 1 */
    const unsigned long skip = (k->thin - (first % k->thin) - 1) % k->thin;
    unsigned long step;
    unsigned int i, j, p;
    char name[100];
    if (k->binary) {
        /* one record per kept iteration: params of chains 0..n_param_chains-1, then (prob, prob - prior)
         * of every chain; packed for the whole batch, written with one call */
        const size_t record = (size_t)k->n_param_chains * n_par + 2 * (size_t)k->n_beta;
        const size_t kept = skip < n_steps ? (n_steps - skip + k->thin - 1) / k->thin : 0;
        double *out;
        if (kept * record > k->pack_capacity) {
            free(k->pack);
            k->pack_capacity = kept * record;
            k->pack = (double *)malloc(k->pack_capacity * sizeof(double));
            assert(k->pack != NULL);
        }
        out = k->pack;
        for (step = skip; step < n_steps; step += k->thin) {
            double *probs = out + (size_t)k->n_param_chains * n_par;
            for (j = 0; j < n_shards; j++) {
                const size_t row = (size_t)(lo[j + 1] - lo[j]) * (n_par + 2);
                const double *r = h[j] + step * row;
                for (i = lo[j]; i < lo[j + 1]; i++, r += n_par + 2) {
                    if (i < k->n_param_chains)
                        memcpy(out + (size_t)i * n_par, r, n_par * sizeof(double));
                    probs[2 * i] = r[n_par];
                    probs[2 * i + 1] = r[n_par + 1];
                }
            }
            out += record;
        }
        fwrite(k->pack, sizeof(double), kept * record, k->bin);
        k->batches++;
        return;
    }
    /* chain-major: one file at a time stays hot, and ladders beyond the descriptor limit
     * (the reference asserts n_beta < 100) open, append to and close one prob file at a time */
    for (j = 0; j < n_shards; j++) {
        const size_t row = (size_t)(lo[j + 1] - lo[j]) * (n_par + 2);
        for (i = lo[j]; i < lo[j + 1]; i++) {
            FILE *pf = k->prob_files ? k->prob_files[i] : NULL;
            FILE **vf = chains[i]->files;
            if (pf == NULL) {
                sprintf(name, "prob-chain%d.dump", i);
                pf = open_or_die(name, k->batches == 0 ? k->mode : "a");
            }
            for (step = skip; step < n_steps; step += k->thin) {
                const double *r = h[j] + step * row + (size_t)(i - lo[j]) * (n_par + 2);
                if (vf != NULL)
                    for (p = 0; p < n_par; p++)
                        if (vf[p] != NULL)
                            fprintf(vf[p], DUMP_FORMAT "\n", r[p]);
                fprintf(pf, "%6e\t%6e\n", r[n_par], r[n_par + 1]);
            }
            if (k->prob_files == NULL)
                fclose(pf);
        }
    }
    k->batches++;
}

/* the same batch when the device has already packed it (apemost_hip_samples_pack_read_async; one
 * shard): binary sinks write the pinned buffer as it is, the thinned text sink finds the kept rows
 * [kept][n_beta][n_par+2] */
static void sink_write_packed(sample_sink *k, mcmc **chains, const double *packed, unsigned long kept) {
    const unsigned int n_par = k->n_par;
    unsigned long step;
    unsigned int i, p;
    char name[100];
    if (k->binary) {
        const size_t record = (size_t)k->n_param_chains * n_par + 2 * (size_t)k->n_beta;
        fwrite(packed, sizeof(double), kept * record, k->bin);
        k->batches++;
        return;
    }
    for (i = 0; i < k->n_beta; i++) {
        FILE *pf = k->prob_files ? k->prob_files[i] : NULL;
        FILE **vf = chains[i]->files;
        if (pf == NULL) {
            sprintf(name, "prob-chain%d.dump", i);
            pf = open_or_die(name, k->batches == 0 ? k->mode : "a");
        }
        for (step = 0; step < kept; step++) {
            const double *r = packed + (step * k->n_beta + i) * (size_t)(n_par + 2);
            if (vf != NULL)
                for (p = 0; p < n_par; p++)
                    if (vf[p] != NULL)
                        fprintf(vf[p], DUMP_FORMAT "\n", r[p]);
            fprintf(pf, "%6e\t%6e\n", r[n_par], r[n_par + 1]);
        }
        if (k->prob_files == NULL)
            fclose(pf);
    }
    k->batches++;
}

static void sink_flush(sample_sink *k) {
    unsigned int i;
    if (k->bin)
        fflush(k->bin);
    if (k->prob_files)
        for (i = 0; i < k->n_beta; i++)
            fflush(k->prob_files[i]);
}

static void sink_close(sample_sink *k) {
    unsigned int i;
    free(k->pack);
    if (k->bin)
        fclose(k->bin);
    if (k->prob_files) {
        for (i = 0; i < k->n_beta; i++)
            fclose(k->prob_files[i]);
        mem_free(k->prob_files);
    }
}

/* The sampler loop.  The device runs batches of rounds (apemost_hip_run: n_swap steps per chain,
 * one swap attempt, ... -- the body of the reference's loop, src/parallel_tempering.c:392-409);
 * while batch k+1 runs, the rows of batch k drain into pinned host memory on a second stream and
 * are written by the sink.  Batches end at the iterations where the reference prints its
 * acceptance line (src/parallel_tempering.c:320-326, 405-407); the accept counters for that line are
 * snapshotted in stream order together with the rows, so the loop never stalls the device. */
static void run_sampler(mcmc **chains, const unsigned int n_beta, const unsigned int n_swap,
                        const unsigned long max_iterations, char *mode) {
    const unsigned int n_par = get_n_par(chains[0]);
    const size_t row = (size_t)n_beta * (n_par + 2);
    unsigned long iter = chains[0]->n_iter;
    /* rounds between two acceptance lines; batches never cross such a point */
    const unsigned long interval_rounds = PRINT_PROB_INTERVAL / gcd_ul(PRINT_PROB_INTERVAL, n_swap);
    unsigned long max_rounds = (unsigned long)(((size_t)64 << 20) / (row * n_swap * sizeof(double)));
    unsigned long rounds_now, rounds_next;
    sample_sink sink;
    FILE *acceptance_file;
    apemost_ladder *l;
    /* double-buffered per shard: device rows, pinned host rows, pinned accept/reject snapshot */
    double *d_samples[2][APEMOST_MAX_SHARDS], *h_samples[2][APEMOST_MAX_SHARDS];
    uint64_t *h_counts[2][APEMOST_MAX_SHARDS];
    double *d_packed[2] = {NULL, NULL};
    unsigned int lo[APEMOST_MAX_SHARDS + 1], n_shards, i, j;
    int k = 0, device_pack;

    if (max_rounds < 1)
        max_rounds = 1;
    if (max_rounds > interval_rounds)
        max_rounds = interval_rounds;
    sink_open(&sink, chains, n_beta, n_par, n_swap, mode);
    acceptance_file = fopen("acceptance_rate.dump.gnuplot", "w");
    if (acceptance_file != NULL) {
        fprintf(acceptance_file, "# format: iteration | number of accepts for each chain\nplot ");
        for (i = 0; i < n_beta; i++)
            fprintf(acceptance_file, "\"acceptance_rate.dump\" u 1:%d title \"chain %d, beta = %f\"%s", i + 2, i,
                    get_beta(chains[i]), i + 1 != n_beta ? ", " : "");
        fprintf(acceptance_file, "\n");
        fclose(acceptance_file);
    }
    acceptance_file = open_or_die("acceptance_rate.dump", mode);

    l = apemost_ladder_open(chains, n_beta);
    n_shards = apemost_ladder_shards(l);
    for (j = 0; j <= n_shards; j++)
        lo[j] = apemost_ladder_shard_first(l, j);
    for (i = 0; i < 2; i++)
        for (j = 0; j < n_shards; j++) {
            apemost_hip_sampler *s = apemost_ladder_shard(l, j);
            const size_t n_local = lo[j + 1] - lo[j];
            void *p = NULL;
            apemost_hip_or_die(apemost_hip_samples_alloc(s, max_rounds * n_swap, &d_samples[i][j]), "samples_alloc");
            apemost_hip_or_die(apemost_hip_host_alloc(max_rounds * n_swap * n_local * (n_par + 2) * sizeof(double), &p),
                               "host_alloc");
            h_samples[i][j] = (double *)p;
            /* (+ n_par doubles: chain 0's latest point behind the counters of a packed read) */
            apemost_hip_or_die(apemost_hip_host_alloc((2 * n_local + n_par) * sizeof(uint64_t), &p), "host_alloc");
            h_counts[i][j] = (uint64_t *)p;
        }
    /* One shard and a sink that does not want every row as it is (binary records, thinning): the
     * device packs each batch into what will be written, so that only that crosses PCIe and the host
     * writes the pinned buffer without touching it. */
    device_pack = n_shards == 1 && (sink.binary || sink.thin > 1);
    for (i = 0; i < 2 && device_pack; i++)
        apemost_hip_or_die(apemost_hip_samples_alloc(apemost_ladder_shard(l, 0), max_rounds * n_swap, &d_packed[i]),
                           "samples_alloc");
    get_duration();
    run = 1;
    dumpflag = 0;
    printf("starting the analysis\n");
    fflush(stdout);

#define PLAN_BATCH(at, out)                                                                       \
    do {                                                                                          \
        (out) = 0;                                                                                \
        if (run && (max_iterations == 0 || (at) < max_iterations)) {                              \
            (out) = interval_rounds - ((at) / n_swap) % interval_rounds;                          \
            if ((out) > max_rounds)                                                               \
                (out) = max_rounds;                                                               \
            if (max_iterations != 0 && (out) > (max_iterations - (at) + n_swap - 1) / n_swap)     \
                (out) = (max_iterations - (at) + n_swap - 1) / n_swap;                            \
        }                                                                                         \
    } while (0)

    PLAN_BATCH(iter, rounds_now);
    if (rounds_now > 0)
        apemost_ladder_run(l, rounds_now, n_swap, d_samples[k]);
    while (rounds_now > 0) {
        const unsigned long n_steps = rounds_now * n_swap, iter_after = iter + n_steps;
        uint64_t kept = 0;
        if (device_pack)
            apemost_hip_or_die(apemost_hip_samples_pack_read_async(apemost_ladder_shard(l, 0), d_samples[k][0], n_steps,
                                                                   (sink.thin - (iter % sink.thin) - 1) % sink.thin, sink.thin,
                                                                   (int32_t)sink.n_param_chains, sink.binary ? 0 : 1, d_packed[k],
                                                                   h_samples[k][0], h_counts[k][0], &kept),
                               "samples_pack_read_async");
        for (j = 0; j < n_shards && !device_pack; j++)
            apemost_hip_or_die(apemost_hip_samples_read_async(apemost_ladder_shard(l, j), d_samples[k][j], n_steps,
                                                              h_samples[k][j], h_counts[k][j]),
                               "samples_read_async");
        PLAN_BATCH(iter_after, rounds_next);
        if (rounds_next > 0) /* the device goes on while this batch drains and is written */
            apemost_ladder_run(l, rounds_next, n_swap, d_samples[k ^ 1]);
        for (j = 0; j < n_shards; j++)
            apemost_hip_or_die(apemost_hip_samples_wait(apemost_ladder_shard(l, j)), "samples_wait");
        if (device_pack)
            sink_write_packed(&sink, chains, h_samples[k][0], (unsigned long)kept);
        else
            sink_write(&sink, chains, h_samples[k], lo, n_shards, iter, n_steps);
        iter = iter_after;
        apemost_swap_round += rounds_now;
        if (iter % PRINT_PROB_INTERVAL == 0) {
            /* chain 0's latest row and counters live in shard 0 */
            /* (a packed batch holds the kept iterations only -- an older one, or none: the packed read leaves
             * chain 0's point after the batch's last step behind the counters) */
            const double *last = !device_pack ? h_samples[k][0] + (n_steps - 1) * (size_t)(lo[1] - lo[0]) * (n_par + 2)
                                              : (const double *)(h_counts[k][0] + 2 * (lo[1] - lo[0]));
            const uint64_t accept0 = h_counts[k][0][0], reject0 = h_counts[k][0][lo[1] - lo[0]];
            if (dumpflag) {
                /* a report on request: the ladder as the device holds it now (a batch ahead of
                 * the rows just written when another one is already running) */
                apemost_ladder_download(l);
                report((const mcmc **)chains, (int)n_beta);
                dumpflag = 0;
                sink_flush(&sink);
            }
            fprintf(acceptance_file, "%lu", iter);
            for (j = 0; j < n_shards; j++)
                for (i = 0; i < lo[j + 1] - lo[j]; i++)
                    fprintf(acceptance_file, "\t%lu", (unsigned long)h_counts[k][j][i]);
            fprintf(acceptance_file, "\n");
            fflush(acceptance_file);
            printf("iteration: %lu, a/r: %.3f(%lu/%lu), v:", iter, (double)accept0 / (double)(accept0 + reject0),
                   (unsigned long)accept0, (unsigned long)reject0);
            printf("Vector%ud[", n_par); /* dump_vector's format, from the row instead of a gsl_vector */
            for (i = 0; i < n_par; i++)
                printf("%f%s", last[i], i + 1 < n_par ? ";" : "]");
            printf(" [%d/%lu ticks]\r", get_duration(), get_ticks_per_second());
            fflush(stdout);
        }
        rounds_now = rounds_next;
        k ^= 1;
    }
#undef PLAN_BATCH
    apemost_ladder_download(l);
    for (i = 0; i < 2; i++)
        for (j = 0; j < n_shards; j++) {
            apemost_hip_samples_free(apemost_ladder_shard(l, j), d_samples[i][j]);
            apemost_hip_host_free(h_samples[i][j]);
            apemost_hip_host_free(h_counts[i][j]);
        }
    for (i = 0; i < 2; i++)
        if (d_packed[i] != NULL)
            apemost_hip_samples_free(apemost_ladder_shard(l, 0), d_packed[i]);
    apemost_ladder_close(l);
    fclose(acceptance_file);
    sink_close(&sink);
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
