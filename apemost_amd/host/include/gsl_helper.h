#ifndef GSL_HELPER_H_
#define GSL_HELPER_H_

#include <gsl/gsl_vector.h>
#include <gsl/gsl_matrix.h>
#include <gsl/gsl_histogram.h>

gsl_vector *dup_vector(const gsl_vector *v); /* new copy, caller frees */
double calc_vector_sum(const gsl_vector *v);
void max_vector(gsl_vector *a, const gsl_vector *b); /* a := element-wise max(a, b) */
void min_vector(gsl_vector *a, const gsl_vector *b); /* a := element-wise min(a, b) */

#endif
