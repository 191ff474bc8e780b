#ifndef GSL_HELPER_H_
#define GSL_HELPER_H_

#include <gsl/gsl_vector.h>
#include <gsl/gsl_matrix.h>

gsl_vector *dup_vector(const gsl_vector *v); /* new copy, caller frees */
double calc_vector_sum(const gsl_vector *v);

#endif
