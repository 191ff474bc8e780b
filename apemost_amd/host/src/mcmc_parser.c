/* `params` and `data` file readers (formats and error behaviour of reference
 * src/mcmc_parser.c:47-181: one "start min max name step" line per parameter, step < 0
 * meaning 10 % of the range; data = whitespace-separated numeric table). */
#include <string.h>
#include "mcmc.h"
#include "mcmc_internal.h"
#include "utils.h"
#include "debug.h"

#define MAX_LINE_LENGTH 256

static int read_parameter_line(mcmc *m, FILE *f, unsigned int i) {
    double start, lo, hi, step;
    char *name = (char *)mem_calloc(MAX_LINE_LENGTH, sizeof(char));
    int got = fscanf(f, "%lf\t%lf\t%lf\t%255s\t%lf\n", &start, &lo, &hi, name, &step);
    if (got != 5) {
        fprintf(stderr, "only %d fields matched.\n", got);
        return 1;
    }
    if (strlen(name) == 0) {
        fprintf(stderr, "description invalid: %s\n", name);
        return 1;
    }
    if (lo > hi) {
        fprintf(stderr, "min(%f) < max(%f)\n", lo, hi);
        return 1;
    }
    if (start > hi) {
        fprintf(stderr, "start(%f) > max(%f)\n", start, hi);
        return 1;
    }
    if (start < lo) {
        fprintf(stderr, "start(%f) < min(%f)\n", start, lo);
        return 1;
    }
    if (step < 0)
        step = (hi - lo) * 0.1;
    gsl_vector_set(m->params, i, start);
    gsl_vector_set(m->params_best, i, start);
    gsl_vector_set(m->params_min, i, lo);
    gsl_vector_set(m->params_max, i, hi);
    gsl_vector_set(m->params_step, i, step);
    m->params_descr[i] = name;
    return 0;
}

mcmc *mcmc_load_params(const char *filename) {
    const unsigned int n = countlines(filename);
    mcmc *m = mcmc_init(n);
    FILE *f = openfile(filename);
    unsigned int i;
    for (i = 0; i < n; i++) {
        if (read_parameter_line(m, f, i) != 0) {
            fprintf(stderr, "Line %u of %s is of incorrect format.\n", i + 1, filename);
            exit(1);
        }
    }
    fclose(f);
    return m;
}

void mcmc_load_data(mcmc *m, const char *datafilename) {
    const unsigned int rows = countlines(datafilename), cols = get_column_count(datafilename);
    gsl_matrix *data = gsl_matrix_alloc(rows, cols);
    FILE *f = openfile(datafilename);
    if (gsl_matrix_fscanf(f, data) != 0) {
        fprintf(stderr, "error reading input data. Perhaps inconsistent format?\n");
        fprintf(stderr, "tried to read %u x %u.\n", cols, rows);
        exit(3);
    }
    fclose(f);
    m->data = data;
    mcmc_check(m);
}

void mcmc_reuse_data(mcmc *m, const mcmc *m_orig) {
    assert(m_orig->data != NULL);
    m->data = m_orig->data;
    mcmc_check(m);
}

mcmc *mcmc_load(const char *filename, const char *datafilename) {
    mcmc *m = mcmc_load_params(filename);
    mcmc_load_data(m, datafilename);
    return m;
}
