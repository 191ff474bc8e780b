/* Example user likelihood written against the plugin API (calc_model / calc_model_for):
 * a sine with Gaussian errors, y = A sin(2 pi (f x + phi)) + o, noise sigma SIGMA.
 * Own example code for the build's host tests; the reference's apps/simplesin.c links
 * against the same headers unchanged (tests/test_host_layer.py, when /root/reference exists). */
#include <gsl/gsl_sf.h>
#include "mcmc.h"
#include "parallel_tempering.h"

#ifndef SIGMA
#define SIGMA 0.5
#endif

void calc_model(mcmc *m, const gsl_vector *old_values) {
    const double a = gsl_vector_get(m->params, 0), f = gsl_vector_get(m->params, 1);
    const double phi = gsl_vector_get(m->params, 2), o = gsl_vector_get(m->params, 3);
    double chi = 0;
    unsigned int i;
    (void)old_values;
    for (i = 0; i < m->data->size1; i++) {
        const double x = gsl_matrix_get(m->data, i, 0), y = gsl_matrix_get(m->data, i, 1);
        const double d = a * gsl_sf_sin(2.0 * M_PI * (f * x + phi)) + o - y;
        chi += d * d;
    }
    set_prob(m, get_beta(m) * chi / (-2 * SIGMA * SIGMA));
}

void calc_model_for(mcmc *m, const unsigned int i, const double old_value) {
    (void)i;
    (void)old_value;
    calc_model(m, NULL);
}
