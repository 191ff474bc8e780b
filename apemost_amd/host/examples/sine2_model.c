/* Example of a likelihood that is NOT one of the engine's built-in device models: a sine with two
 * free parameters (amplitude, frequency) and a fixed phase -- the model of the reference's
 * apps/simplesin2.c.  Own example code for the host tests (on the GPU box the reference tree does not
 * exist).  Its device twin is examples/device_models/simplesin2.hip; run with
 *   APEMOST_DEVICE_MODEL_SRC=.../device_models/simplesin2.hip ./sine2.exe calibrate_first
 * The engine samples with the device source only after it has reproduced this function at probe
 * points (apemost_detect_model). */
#include <gsl/gsl_sf.h>
#include "mcmc.h"
#include "parallel_tempering.h"

#ifndef SIGMA
#define SIGMA 0.5
#endif
#define FIXED_PHASE 0.3312

void calc_model(mcmc *m, const gsl_vector *old_values) {
    const double amplitude = gsl_vector_get(m->params, 0), frequency = gsl_vector_get(m->params, 1);
    double chi = 0;
    unsigned int i;
    (void)old_values;
    for (i = 0; i < m->data->size1; i++) {
        const double d = amplitude * gsl_sf_sin(2.0 * M_PI * (frequency * gsl_matrix_get(m->data, i, 0) + FIXED_PHASE)) -
                         gsl_matrix_get(m->data, i, 1);
        chi += d * d;
    }
    set_prob(m, get_beta(m) * chi / (-2 * SIGMA * SIGMA));
}

void calc_model_for(mcmc *m, const unsigned int i, const double old_value) {
    (void)i;
    (void)old_value;
    calc_model(m, NULL);
}
