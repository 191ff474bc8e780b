#ifndef APEMOST_COMPAT_GSL_VECTOR_H
#define APEMOST_COMPAT_GSL_VECTOR_H
#include <stddef.h>
#include <stdio.h>
#include <gsl/gsl_math.h>

typedef struct {
    size_t size;
    double *data;
} gsl_block;

typedef struct {
    size_t size;
    size_t stride;
    double *data;
    gsl_block *block;
    int owner;
} gsl_vector;

typedef struct {
    gsl_vector vector;
} gsl_vector_const_view;
typedef gsl_vector_const_view gsl_vector_view;

gsl_vector *gsl_vector_alloc(const size_t n);
gsl_vector *gsl_vector_calloc(const size_t n);
void gsl_vector_free(gsl_vector *v);
double gsl_vector_get(const gsl_vector *v, const size_t i);
void gsl_vector_set(gsl_vector *v, const size_t i, double x);
void gsl_vector_set_all(gsl_vector *v, double x);
void gsl_vector_set_zero(gsl_vector *v);
int gsl_vector_memcpy(gsl_vector *dest, const gsl_vector *src);
int gsl_vector_scale(gsl_vector *a, const double x);
int gsl_vector_add_constant(gsl_vector *a, const double x);
int gsl_vector_add(gsl_vector *a, const gsl_vector *b);
int gsl_vector_sub(gsl_vector *a, const gsl_vector *b);
int gsl_vector_mul(gsl_vector *a, const gsl_vector *b);
int gsl_vector_div(gsl_vector *a, const gsl_vector *b);
double gsl_vector_max(const gsl_vector *v);
double gsl_vector_min(const gsl_vector *v);
void gsl_vector_minmax(const gsl_vector *v, double *min_out, double *max_out);
int gsl_vector_fprintf(FILE *stream, const gsl_vector *v, const char *format);
#endif
