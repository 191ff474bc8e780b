#ifndef MCMC_INTERNAL_H_
#define MCMC_INTERNAL_H_

#include "mcmc.h"
#include <gsl/gsl_histogram.h>
#include <gsl/gsl_sf.h>

mcmc *mcmc_init(const unsigned int n_pars);
unsigned int countlines(const char *filename);

/* x modulo div for doubles, result in [0, div) */
#define mod_double(x, div) ((x) < 0 ? (x) - (div) * (int)((x) / (div)-1) : (x) - (div) * (int)((x) / (div)))
#define abs_double(x) ((x) < 0 ? -(x) : (x))

#endif
