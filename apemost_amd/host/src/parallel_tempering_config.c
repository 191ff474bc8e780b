/* Chain setup and the inter-phase files (formats of reference
 * src/parallel_tempering_config.c:28-202). */
#include <math.h>
#include <string.h>
#include "mcmc.h"
#include "parallel_tempering_config.h"
#include "apemost_bridge.h"
#include "debug.h"
#include "define_defaults.h"
#include "gsl_helper.h"
#include "utils.h"

unsigned int apemost_n_beta(void) {
    const char *e = getenv("APEMOST_N_BETA");
    if (e != NULL && atoi(e) > 0)
        return (unsigned int)atoi(e);
    return N_BETA;
}

void write_params_file(mcmc *m) {
    FILE *f = fopen(PARAMS_FILENAME "_suggested", "w");
    unsigned int i;
    if (f == NULL) {
        fprintf(stderr, "Could not write to file " PARAMS_FILENAME "_suggested\n");
        return;
    }
    for (i = 0; i < get_n_par(m); i++)
        fprintf(f, DUMP_FORMAT "\t" DUMP_FORMAT "\t" DUMP_FORMAT "\t%s\t" DUMP_FORMAT "\n",
                get_params_best_for(m, i), get_params_min_for(m, i), get_params_max_for(m, i),
                get_params_descr(m)[i], get_steps_for(m, i));
    fclose(f);
    printf("new suggested parameters file has been written\n");
}

void write_calibration_summary(mcmc **chains, unsigned int n_chains) {
    const double beta_0 = get_beta(chains[n_chains - 1]);
    const unsigned int n_pars = get_n_par(chains[0]);
    FILE *f = fopen("calibration_summary", "w");
    unsigned int i, j;
    if (f == NULL) {
        fprintf(stderr, "Could not write to file calibration_summary\n");
        return;
    }
    fprintf(f, "Summary of calibrations\n");
    fprintf(f, "\nBETA TABLE\n");
    fprintf(f, "Chain # | Calculated | Calibrated\n");
    for (i = 0; i < n_chains; i++)
        fprintf(f, "Chain %d | " DUMP_FORMAT " | %f\n", i, get_chain_beta(i, n_chains, beta_0), get_beta(chains[i]));
    fprintf(f, "\nSTEPWIDTH TABLE\n");
    fprintf(f, "Chain # | Calibrated stepwidths... \n");
    for (i = 0; i < n_chains; i++) {
        fprintf(f, "%d", i);
        for (j = 0; j < n_pars; j++)
            fprintf(f, "\t" DUMP_FORMAT, get_steps_for(chains[i], j));
        fprintf(f, "\n");
    }
    fprintf(f, "\nSTEPWIDTH ESTIMATE TABLE\n");
    fprintf(f, "If you find that the estimate deviates much or systematically from the ");
    fprintf(f, "calibrated stepwidths, please notify the authors.\n");
    fprintf(f, "Chain # | Calculated stepwidths... \n");
    for (i = 0; i < n_chains; i++) {
        const double scale = pow(get_beta(chains[i]), -0.5);
        fprintf(f, "%d", i);
        for (j = 0; j < n_pars; j++)
            fprintf(f, "\t" DUMP_FORMAT, get_steps_for(chains[0], j) * scale);
        fprintf(f, "\n");
    }
    fclose(f);
    printf("calibration summary has been written\n");
}

/* n_beta copies of the params file; chain 0 owns the data matrix, the others alias it */
mcmc **setup_chains() {
    const unsigned int n_beta = apemost_n_beta();
    mcmc **chains = (mcmc **)mem_calloc(n_beta, sizeof(mcmc *));
    unsigned int i;
    assert(chains != NULL);
    printf("Initializing %d chains ...\n", n_beta);
    for (i = 0; i < n_beta; i++) {
        parallel_tempering_mcmc *t;
        chains[i] = mcmc_load_params(PARAMS_FILENAME);
        if (i == 0)
            mcmc_load_data(chains[i], DATA_FILENAME);
        else
            mcmc_reuse_data(chains[i], chains[0]);
        mcmc_check(chains[i]);
        t = (parallel_tempering_mcmc *)mem_calloc(1, sizeof(parallel_tempering_mcmc));
        apemost_chain_place(chains[i], i);
        chains[i]->additional_data = t;
        set_beta(chains[i], 1.0);
    }
    return chains;
}

void read_calibration_file(mcmc **chains, unsigned int n_chains) {
    const unsigned int n_par = get_n_par(chains[0]);
    FILE *f = fopen(CALIBRATION_FILE, "r");
    unsigned int i, j;
    double v;
    if (f == NULL) {
        perror("could not read calibration file '" CALIBRATION_FILE "'");
        exit(1);
    }
    for (i = 0; i < n_chains; i++) {
        int ok = fscanf(f, "%lf", &v) == 1;
        if (ok)
            set_beta(chains[i], v);
        for (j = 0; ok && j < n_par; j++) {
            ok = fscanf(f, "%lf", &v) == 1;
            if (ok)
                set_steps_for(chains[i], v, j);
        }
        for (j = 0; ok && j < n_par; j++) {
            ok = fscanf(f, "%lf", &v) == 1;
            if (ok)
                set_params_for(chains[i], v, j);
        }
        if (!ok) {
            fprintf(stderr, "could not read %d chain calibrations. \nError with line %d.\n", n_chains, i + 1);
            exit(1);
        }
        set_params_best(chains[i], get_params(chains[i]));
    }
    fclose(f);
}

void write_calibrations_file(mcmc **chains, const unsigned int n_chains) {
    const unsigned int n_par = get_n_par(chains[0]);
    FILE *f = fopen(CALIBRATION_FILE, "w");
    unsigned int i, j;
    if (f == NULL) {
        perror("error writing to calibration results file");
        exit(1);
    }
    for (j = 0; j < n_chains; j++) {
        fprintf(f, DUMP_FORMAT, get_beta(chains[j]));
        for (i = 0; i < n_par; i++)
            fprintf(f, "\t" DUMP_FORMAT, get_steps_for(chains[j], i));
        for (i = 0; i < n_par; i++)
            fprintf(f, "\t" DUMP_FORMAT, get_params_for(chains[j], i));
        fprintf(f, "\n");
    }
    fclose(f);
    printf("wrote calibration results for %d chains to %s\n", n_chains, CALIBRATION_FILE);
}
