/* Histogram helpers of the reference API (reference src/histogram.h:22-50): used by the
 * reference's unit tests, its `analyse` phase and the command-line tools. */
#ifndef HISTOGRAM
#define HISTOGRAM

#include <gsl/gsl_histogram.h>
#include <gsl/gsl_vector.h>

/* nbins equal bins over [min, max]; the top edge is pushed out a little so that a value equal
 * to max still lands in the last bin */
gsl_histogram *create_hist(int nbins, double min, double max);
/* density of the entries of v over [min(v), max(v)] */
gsl_histogram *calc_hist(const gsl_vector *v, int nbins);
/* a file of n whitespace-separated columns: column i is added to hists[i] */
void append_to_hists(gsl_histogram **hists, unsigned int n, const char *filename);
/* per-column extremes of such a file: set / widen the two vectors */
void find_min_max(char *filename, gsl_vector *min, gsl_vector *max);
void update_min_max(char *filename, gsl_vector *min, gsl_vector *max);

#endif
