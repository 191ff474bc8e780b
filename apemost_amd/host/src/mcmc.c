/* Chain objects: construction, destruction, sanity checks
 * (behaviour of reference src/mcmc.c:27-131). */
#include <string.h>
#include "mcmc.h"
#include "mcmc_internal.h"
#include "debug.h"
#include "apemost_bridge.h"

/* one host RNG shared by every chain of the process, created on first use */
static gsl_rng *shared_rng = NULL;

static void attach_rng(mcmc *m) {
    if (shared_rng == NULL) {
        gsl_rng_env_setup();
        shared_rng = gsl_rng_alloc(gsl_rng_default);
    }
    m->random = shared_rng;
}

mcmc *mcmc_init(const unsigned int n_pars) {
    mcmc *m = (mcmc *)mem_malloc(sizeof(mcmc));
    assert(m != NULL);
    m->n_par = n_pars;
    m->accept = 0;
    m->reject = 0;
    m->n_iter = 0;
    m->prob = -1e+10; /* "not evaluated yet": the first proposal always wins against it */
    m->prob_best = -1e+10;
    m->prior = 0;
    m->files = NULL;
    m->data = NULL;
    m->additional_data = NULL;
    attach_rng(m);
    m->params = gsl_vector_alloc(n_pars);
    m->params_best = gsl_vector_alloc(n_pars);
    m->params_step = gsl_vector_calloc(n_pars);
    m->params_min = gsl_vector_calloc(n_pars);
    m->params_max = gsl_vector_calloc(n_pars);
    m->params_accepts = (unsigned long *)mem_calloc(n_pars, sizeof(unsigned long));
    m->params_rejects = (unsigned long *)mem_calloc(n_pars, sizeof(unsigned long));
    m->params_descr = (const char **)mem_calloc(n_pars, sizeof(char *));
    assert(m->params && m->params_best && m->params_step && m->params_min && m->params_max);
    assert(m->params_accepts && m->params_rejects && m->params_descr);
    return m;
}

mcmc *mcmc_free(mcmc *m) {
    unsigned int i;
    apemost_chain_forget(m);
    mcmc_dump_close(m);
    if (shared_rng != NULL && m->random == shared_rng) {
        gsl_rng_free(shared_rng);
        shared_rng = NULL;
    }
    gsl_vector_free(m->params);
    gsl_vector_free(m->params_best);
    for (i = 0; i < m->n_par; i++)
        mem_free(m->params_descr[i]);
    mem_free(m->params_descr);
    mem_free(m->params_accepts);
    mem_free(m->params_rejects);
    gsl_vector_free(m->params_step);
    gsl_vector_free(m->params_min);
    gsl_vector_free(m->params_max);
    if (m->data != NULL)
        gsl_matrix_free((gsl_matrix *)m->data);
    mem_free(m);
    return NULL;
}

void mcmc_check(const mcmc *m) {
    (void)m;
    assert(m != NULL);
    assert(m->n_par > 0);
    assert(m->data != NULL);
    assert(m->data->size2 > 0);
    assert(m->params != NULL && m->params->size == m->n_par);
    assert(m->params_best != NULL && m->params_best->size == m->n_par);
    assert(m->params_step != NULL);
}
