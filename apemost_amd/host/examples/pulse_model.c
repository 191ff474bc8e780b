/* Example user likelihood written against the plugin API (calc_model / calc_model_for): a power
 * spectrum of Lorentzian modes with exponential (chi^2, 2 d.o.f.) noise -- the model the engine's
 * built-in device model APEMOST_MODEL_PULSE evaluates (the reference's apps/pulse.c is the same model
 * and links against the same headers unchanged).
 *   parameters: lifetime, additive term, then (frequency, height) per mode
 *   data rows:  (frequency, observed power)
 *   log-posterior = prior - beta * (additive + sum_i [ln y_i + d_i / y_i]),
 *   y_i = sum_modes height / (1 + (2 pi (f_mode - f_i) lifetime)^2),
 *   prior = -mean_modes ln(height + HMIN)
 * Own example code for the build's host tests and profiles (tools/experiments/r04_host_profile.sh). */
#include <gsl/gsl_sf.h>
#include "mcmc.h"
#include "parallel_tempering.h"

#ifndef HMIN
#define HMIN 1e-6
#endif

static double height_prior(const mcmc *m) {
    const unsigned int n = get_n_par(m);
    const unsigned int n_modes = (n - 2) / 2;
    double sum = 0;
    unsigned int k;
    for (k = 0; k < n_modes; k++)
        sum += gsl_sf_log(gsl_vector_get(m->params, 3 + 2 * k) + HMIN);
    return -sum / n_modes;
}

void calc_model(mcmc *m, const gsl_vector *old_values) {
    const unsigned int n = get_n_par(m);
    const double lifetime = gsl_vector_get(m->params, 0);
    double total = gsl_vector_get(m->params, 1);
    unsigned int i, k;
    (void)old_values;
    set_prior(m, height_prior(m));
    for (i = 0; i < m->data->size1; i++) {
        const double f_i = gsl_matrix_get(m->data, i, 0), d_i = gsl_matrix_get(m->data, i, 1);
        double y = 0;
        for (k = 2; k < n; k += 2) {
            const double detune = gsl_vector_get(m->params, k) - f_i;
            const double w = 2 * M_PI * detune * lifetime;
            y += gsl_vector_get(m->params, k + 1) / (1 + w * w);
        }
        total += gsl_sf_log(y) + d_i / y;
    }
    set_prob(m, get_prior(m) + -get_beta(m) * total);
}

void calc_model_for(mcmc *m, const unsigned int i, const double old_value) {
    (void)i;
    (void)old_value;
    calc_model(m, NULL);
}
