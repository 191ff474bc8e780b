#ifndef APEMOST_COMPAT_GSL_RANDIST_H
#define APEMOST_COMPAT_GSL_RANDIST_H
#include <gsl/gsl_rng.h>
double gsl_ran_gaussian(const gsl_rng *r, const double sigma);
double gsl_ran_logistic(const gsl_rng *r, const double a);
double gsl_ran_flat(const gsl_rng *r, const double a, const double b);
#endif
