#include "gsl_helper.h"
#include "utils.h"

gsl_vector *dup_vector(const gsl_vector *v) {
    gsl_vector *r;
    assert(v != NULL && v->size > 0);
    r = gsl_vector_alloc(v->size);
    assert(r != NULL);
    gsl_vector_memcpy(r, v);
    return r;
}

double calc_vector_sum(const gsl_vector *v) {
    double s = 0;
    size_t i;
    for (i = 0; i < v->size; i++)
        s += gsl_vector_get(v, i);
    return s;
}

/* element-wise: a := max(a, b) / a := min(a, b) */
void max_vector(gsl_vector *a, const gsl_vector *b) {
    size_t i;
    assert(a->size == b->size);
    for (i = 0; a != b && i < a->size; i++)
        if (gsl_vector_get(b, i) > gsl_vector_get(a, i))
            gsl_vector_set(a, i, gsl_vector_get(b, i));
}

void min_vector(gsl_vector *a, const gsl_vector *b) {
    size_t i;
    assert(a->size == b->size);
    for (i = 0; a != b && i < a->size; i++)
        if (gsl_vector_get(b, i) < gsl_vector_get(a, i))
            gsl_vector_set(a, i, gsl_vector_get(b, i));
}
