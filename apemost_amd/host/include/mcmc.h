/* Public chain API (same names and meaning as reference src/mcmc.h:61-173). */
#ifndef MCMC
#define MCMC

#include <stdio.h>
#include <stdlib.h>

/* number format of every dump file */
#define DUMP_FORMAT "%.15e"

#ifdef NOASSERT
#define assert(cond)
#else
#include <assert.h>
#endif

#include "mcmc_struct.h"

mcmc *mcmc_load(const char *filename, const char *datafilename);
mcmc *mcmc_load_params(const char *filename);
void mcmc_load_data(mcmc *m, const char *datafilename);
/* share the data matrix of m_orig (not copied, not owned) */
void mcmc_reuse_data(mcmc *m, const mcmc *m_orig);
/* returns NULL so callers can write m = mcmc_free(m) */
mcmc *mcmc_free(mcmc *m);
void mcmc_check(const mcmc *m);
/* write the current point to the dump files (if open) and count one sample */
void mcmc_append_current_parameters(mcmc *m);
void mcmc_dump_y_dat(mcmc *m, const gsl_vector *y_dat, const char *filename);
void mcmc_dump_flush(const mcmc *m);
void mcmc_dump_close(mcmc *m);
/* file names: <param name><suffix>-<index>.prob.dump */
void mcmc_open_dump_files(mcmc *m, const char *suffix, int index, char *mode);
void mcmc_dump_current(const mcmc *m);
void mcmc_dump_probabilities(const mcmc *m, int n_values, const char *suffix);
void mcmc_check_best(mcmc *m);

#include "markov_chain.h"
#include "mcmc_gettersetter.h"

/* ---- supplied by the application (one model per executable) ---- */
/* set_prob(m, log-prior + beta*log-likelihood) at m->params; old_values may be NULL */
void calc_model(mcmc *m, const gsl_vector *old_values);
/* same, when only parameter i changed */
void calc_model_for(mcmc *m, const unsigned int i, const double old_value);

#endif
