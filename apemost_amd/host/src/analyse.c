/* `analyse` phase entry points.  Post-processing of the dump files is disk-bound host work
 * outside the GPU path (next); the data-probability integral is provided, the
 * histogram writer is not yet. */
#include <math.h>
#include <string.h>
#include "mcmc.h"
#include "parallel_tempering.h"
#include "parallel_tempering_config.h"
#include "debug.h"

/* thermodynamic integration: mean of (prob - prior)/beta per chain from prob-chain<i>.dump,
 * integrated over beta with the rectangle rule (reference src/analyse.c:33-113) */
void analyse_data_probability() {
    const unsigned int n_beta = apemost_n_beta();
    mcmc **chains = setup_chains();
    double *mean = (double *)calloc(n_beta, sizeof(double));
    double evidence = 0, prev_beta = 0;
    unsigned int i;
    read_calibration_file(chains, n_beta);
    for (i = 0; i < n_beta; i++) {
        char name[100];
        FILE *f;
        double a, b, sum = 0;
        unsigned long n = 0;
        sprintf(name, "prob-chain%d.dump", i);
        f = fopen(name, "r");
        if (f == NULL) {
            fprintf(stderr, "could not read %s\n", name);
            exit(1);
        }
        while (fscanf(f, "%lf %lf", &a, &b) == 2) {
            sum += b;
            n++;
        }
        fclose(f);
        mean[i] = n ? sum / n / get_beta(chains[i]) : 0;
        printf("chain %u: beta = %f, <ln L> = %f (%lu samples)\n", i, get_beta(chains[i]), mean[i], n);
    }
    for (i = n_beta; i-- > 0;) { /* hottest (smallest beta) first */
        evidence += mean[i] * (get_beta(chains[i]) - prev_beta);
        prev_beta = get_beta(chains[i]);
    }
    printf("Model probability ln(p(D|M, I)): [about 10^%.0f] %f\n", evidence / log(10.0), evidence);
    free(mean);
}

void analyse_marginal_distributions() {
    fprintf(stderr, "analyse marginal: histogram post-processing is not part of this engine yet; "
                    "the dump files are in the reference's format and its tools read them.\n");
    exit(2);
}
