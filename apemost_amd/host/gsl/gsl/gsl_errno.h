#ifndef APEMOST_COMPAT_GSL_ERRNO_H
#define APEMOST_COMPAT_GSL_ERRNO_H
#include <gsl/gsl_math.h>
#endif
