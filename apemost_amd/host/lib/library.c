/* The shared-library flavour of the toolkit (reference apps/library.c:6-29, Makefile:48-49):
 * instead of linking a calc_model() into an executable, a client -- typically another language
 * through its FFI -- loads libapemost.so and registers two callbacks, the log-likelihood and the
 * log-prior, with set_function().  calc_model() below is the plugin the engine then sees.
 *
 * The reference combines them as prob = beta * (prior + loglike), i.e. the prior is tempered too
 * (SURVEY quirk Q6); kept as it is, clients rely on what the reference computes.
 *
 * On this engine the callbacks serve every host-side call of calc_model (eval, benchmark, the
 * model detection).  The sampler phases run on the device, so they need the registered pair to
 * coincide with one of the device likelihoods (apemost_detect_model compares them at points inside
 * the prior box, at more than one beta); otherwise they stop with a message -- there is no CPU
 * sampler behind this library. */
#include <stdio.h>
#include <stdlib.h>
#include <gsl/gsl_sf.h>
#include "mcmc.h"
#include "parallel_tempering.h"

typedef double (*apemost_callback)(mcmc *m, const gsl_vector *old_values);

static apemost_callback client_loglike = NULL, client_prior = NULL;

void set_function(apemost_callback LogLike, apemost_callback Prior) {
    client_loglike = LogLike;
    client_prior = Prior;
}

void calc_model(mcmc *m, const gsl_vector *old_values) {
    double prior;
    if (client_loglike == NULL || client_prior == NULL) {
        fprintf(stderr, "libapemost: calc_model() called before set_function(LogLike, Prior)\n");
        exit(1);
    }
    prior = client_prior(m, old_values);
    set_prior(m, prior);
    set_prob(m, get_beta(m) * (prior + client_loglike(m, old_values)));
}

void calc_model_for(mcmc *m, const unsigned int i, const double old_value) {
    (void)i;
    (void)old_value;
    calc_model(m, NULL);
}
