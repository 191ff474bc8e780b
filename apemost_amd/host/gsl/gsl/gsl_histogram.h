#ifndef APEMOST_COMPAT_GSL_HISTOGRAM_H
#define APEMOST_COMPAT_GSL_HISTOGRAM_H
#include <stddef.h>
#include <stdio.h>
/* the part of gsl_histogram the reference sources touch: bin i covers [range[i], range[i+1]) */
typedef struct {
    size_t n;
    double *range;
    double *bin;
} gsl_histogram;

gsl_histogram *gsl_histogram_alloc(size_t n);
void gsl_histogram_free(gsl_histogram *h);
int gsl_histogram_set_ranges_uniform(gsl_histogram *h, double xmin, double xmax);
int gsl_histogram_increment(gsl_histogram *h, double x);
double gsl_histogram_get(const gsl_histogram *h, size_t i);
int gsl_histogram_get_range(const gsl_histogram *h, size_t i, double *lower, double *upper);
double gsl_histogram_max(const gsl_histogram *h);
double gsl_histogram_min(const gsl_histogram *h);
size_t gsl_histogram_bins(const gsl_histogram *h);
double gsl_histogram_sum(const gsl_histogram *h);
double gsl_histogram_mean(const gsl_histogram *h);
double gsl_histogram_sigma(const gsl_histogram *h);
int gsl_histogram_scale(gsl_histogram *h, double scale);
int gsl_histogram_fprintf(FILE *stream, const gsl_histogram *h, const char *range_format, const char *bin_format);
#endif
