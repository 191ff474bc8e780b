/* Per-parameter visited-value files "<name><suffix>-<index>.prob.dump", one DUMP_FORMAT
 * value per line per step (formats of reference src/mcmc_dump.c:28-113). */
#include <string.h>
#include "mcmc.h"
#include "debug.h"

void mcmc_dump_y_dat(mcmc *m, const gsl_vector *y_dat, const char *filename) {
    FILE *f;
    size_t i, j;
#ifdef NODUMP
    return;
#endif
    assert(m->data->size1 == y_dat->size);
    f = fopen(filename, "w");
    assert(f != NULL);
    for (i = 0; i < m->data->size1; i++) {
        for (j = 0; j < m->data->size2; j++)
            fprintf(f, DUMP_FORMAT "\t", gsl_matrix_get(m->data, i, j));
        fprintf(f, DUMP_FORMAT "\n", gsl_vector_get(y_dat, i));
    }
    fclose(f);
}

void mcmc_open_dump_files(mcmc *m, const char *suffix, int index, char *mode) {
    unsigned int i;
    m->files = (FILE **)mem_calloc(m->n_par, sizeof(FILE *));
#ifdef NODUMP
    return;
#endif
    for (i = 0; i < get_n_par(m); i++) {
        char *name = (char *)mem_calloc(strlen(m->params_descr[i]) + strlen(suffix) + 32, sizeof(char));
        sprintf(name, "%s%s-%d.prob.dump", m->params_descr[i], suffix, index);
        m->files[i] = fopen(name, mode);
        assert(m->files[i] != NULL);
        mem_free(name);
    }
}

void mcmc_dump_current(const mcmc *m) {
    unsigned int i;
    if (m->files == NULL)
        return;
    for (i = 0; i < get_n_par(m); i++)
        if (m->files[i] != NULL)
            fprintf(m->files[i], DUMP_FORMAT "\n", gsl_vector_get(m->params, i));
}

void mcmc_dump_flush(const mcmc *m) {
    unsigned int i;
    if (m->files == NULL)
        return;
    for (i = 0; i < get_n_par(m); i++)
        if (m->files[i] != NULL)
            fflush(m->files[i]);
}

void mcmc_dump_close(mcmc *m) {
    unsigned int i;
    if (m->files == NULL)
        return;
    for (i = 0; i < get_n_par(m); i++)
        if (m->files[i] != NULL)
            fclose(m->files[i]);
    mem_free(m->files);
    m->files = NULL;
}

/* declared by the reference header but never defined there either */
void mcmc_dump_probabilities(const mcmc *m, int n_values, const char *suffix) {
    (void)m;
    (void)n_values;
    (void)suffix;
}
