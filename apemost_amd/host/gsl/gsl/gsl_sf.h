#ifndef APEMOST_COMPAT_GSL_SF_H
#define APEMOST_COMPAT_GSL_SF_H
#include <gsl/gsl_math.h>
/* gsl_sf_log: domain error (abort through gsl_error) for x <= 0, like GSL */
double gsl_sf_log(const double x);
/* libm sin; real GSL evaluates its own series and differs by ulps */
double gsl_sf_sin(const double x);
double gsl_sf_cos(const double x);
double gsl_sf_exp(const double x);
#endif
