/*
 * apemost_device_model.h -- what a user-supplied DEVICE likelihood implements.
 *
 * APEMoST's plugin surface is the pair calc_model() / calc_model_for() (src/mcmc.h:164,173): host C,
 * called once per Metropolis update.  The engine's built-in device models cover the BASELINE apps;
 * any other likelihood (apps/simplesin2.c, apps/normal.c, apps/bernoulli_example.c, or a function
 * registered through set_function, apps/library.c:6-22) is given to the engine as device source:
 * a file with the two functions below, named in apemost_hip_config.device_model_source (C host:
 * environment variable APEMOST_DEVICE_MODEL_SRC).  The engine compiles it with hiprtc into its
 * one-wave round, calibration and evaluation kernels when the sampler is created.  The C host layer
 * accepts it only if it reproduces the application's own host calc_model() on probe points
 * (apemost_detect_model: 1e-9).
 *
 * The likelihood is taken as  m->prob = finish( sum_i term(i) ):  term() is what the loop over the
 * data rows of a calc_model() adds per row (the engine sums the terms in a fixed order over the 64
 * lanes of the chain's wavefront), finish() is everything else: get_beta(m), the prior, set_prob /
 * set_prior.  A likelihood without a data loop returns 0 from term().
 *
 * The file is compiled as HIP device code for gfx950 (-O3 -ffp-contract=off -std=c++17); it may use
 * the device math library (sin, exp, log, pow, ...).  No host code, no other includes.
 */
#ifndef APEMOST_DEVICE_MODEL_H
#define APEMOST_DEVICE_MODEL_H

typedef struct {
    const double *data;   /* m->data, column-major: column j of row i is data[j * n_data + i] */
    int n_data, n_cols;   /* m->data->size1, m->data->size2 */
    const double *params; /* m->params of the point being evaluated, [n_par] */
    int n_par;
    double sigma, hmin;   /* apemost_hip_config.sigma / .hmin (the applications' -DSIGMA / -DHMIN) */
} apemost_model_ctx;

/* gsl_matrix_get(m->data, i, j) */
#define APEMOST_DATA(ctx, i, j) ((ctx)->data[(unsigned long)(j) * (unsigned long)(ctx)->n_data + (unsigned long)(i)])

/* what data row i adds to the sum */
__device__ double apemost_user_term(const apemost_model_ctx *ctx, int i);
/* m->prob from the sum over all rows; *prior = m->prior (set_prior), beta = get_beta(m) */
__device__ double apemost_user_finish(const apemost_model_ctx *ctx, double sum, double beta, double *prior);

#endif
