#ifndef APEMOST_COMPAT_GSL_MATRIX_H
#define APEMOST_COMPAT_GSL_MATRIX_H
#include <gsl/gsl_vector.h>

typedef struct {
    size_t size1;
    size_t size2;
    size_t tda;
    double *data;
    gsl_block *block;
    int owner;
} gsl_matrix;

gsl_matrix *gsl_matrix_alloc(const size_t n1, const size_t n2);
void gsl_matrix_free(gsl_matrix *m);
double gsl_matrix_get(const gsl_matrix *m, const size_t i, const size_t j);
void gsl_matrix_set(gsl_matrix *m, const size_t i, const size_t j, const double x);
void gsl_matrix_set_all(gsl_matrix *m, double x);
int gsl_matrix_fscanf(FILE *stream, gsl_matrix *m);
int gsl_matrix_get_col(gsl_vector *v, const gsl_matrix *m, const size_t j);
gsl_vector_const_view gsl_matrix_const_column(const gsl_matrix *m, const size_t j);
#endif
