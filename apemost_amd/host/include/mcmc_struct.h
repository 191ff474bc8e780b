/* One tempered Markov chain as host code sees it.  Field order and types are the
 * binary contract of APEMoST's `mcmc` (reference src/mcmc_struct.h:30-106): user
 * likelihoods read m->params, m->data, m->n_par and m->additional_data directly.
 * On the MI355X engine this struct is only the host mirror; the live state is the
 * structure-of-arrays block in HBM (include/apemost_hip.h). */
#ifndef MCMC_STRUCT_H_
#define MCMC_STRUCT_H_

#include <stdio.h>
#include <gsl/gsl_math.h>
#include <gsl/gsl_vector.h>
#include <gsl/gsl_matrix.h>
#include <gsl/gsl_rng.h>

typedef struct {
    unsigned int n_par;            /* number of model parameters */
    unsigned long accept;          /* accepted all-parameter steps */
    unsigned long reject;          /* rejected all-parameter steps */
    double prob;                   /* log-posterior of the latest evaluated point */
    double prior;                  /* log-prior part of prob */
    double prob_best;              /* best prob seen */
    gsl_rng *random;               /* host RNG handle (API compatibility) */
    gsl_vector *params;            /* current point, n_par */
    gsl_vector *params_best;       /* best point, n_par */
    FILE **files;                  /* per-parameter dump files or NULL */
    const char **params_descr;     /* parameter names */
    unsigned long *params_accepts; /* per-parameter accept counters */
    unsigned long *params_rejects; /* per-parameter reject counters */
    gsl_vector *params_step;       /* proposal widths */
    gsl_vector *params_min;        /* lower bounds */
    gsl_vector *params_max;        /* upper bounds */
    const gsl_matrix *data;        /* observations, shared by all chains */
    unsigned long n_iter;          /* samples appended */
    void *additional_data;         /* parallel_tempering_mcmc */
} mcmc;

#endif
