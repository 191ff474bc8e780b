#include "debug.h"

void dump_vector(const gsl_vector *v) {
    size_t i;
    printf("Vector%ud[", (unsigned int)v->size);
    for (i = 0; i + 1 < v->size; i++)
        printf("%f;", gsl_vector_get(v, i));
    printf("%f]", gsl_vector_get(v, v->size - 1));
}

void dump_vectorln(const gsl_vector *v) {
    dump_vector(v);
    printf("\n");
}

void dump_mcmc(const mcmc *m) {
    unsigned int i;
    IFDEBUG {
        printf("\t\tn_par=%u; a/r=%lu/%lu prob/best=%f/%f iter=%lu\n", get_n_par(m), m->accept, m->reject,
               m->prob, m->prob_best, m->n_iter);
        for (i = 0; i < get_n_par(m); i++)
            printf("\t\t%s: accepts %lu rejects %lu\n", m->params_descr ? m->params_descr[i] : "?",
                   m->params_accepts[i], m->params_rejects[i]);
        dump_v("values", m->params);
        dump_v("best", m->params_best);
        dump_v("min", m->params_min);
        dump_v("max", m->params_max);
        dump_v("step-size", m->params_step);
    }
}
