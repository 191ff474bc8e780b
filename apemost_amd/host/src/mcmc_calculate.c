#include "mcmc.h"
#include "mcmc_internal.h"

/* reference src/mcmc_calculate.c:30-41 */
void mcmc_append_current_parameters(mcmc *m) {
    mcmc_dump_current(m);
    m->n_iter++;
}

void mcmc_check_best(mcmc *m) {
    if (m->prob > m->prob_best) {
        m->prob_best = m->prob;
        set_params_best(m, m->params);
    }
}
