#ifndef APEMOST_COMPAT_GSL_HISTOGRAM_H
#define APEMOST_COMPAT_GSL_HISTOGRAM_H
#include <stddef.h>
typedef struct {
    size_t n;
    double *range;
    double *bin;
} gsl_histogram;
#endif
