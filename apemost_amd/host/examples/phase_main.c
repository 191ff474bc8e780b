/* Minimal phase dispatcher for the host tests (the reference's apps/generic_main.c links
 * against the same API unchanged). */
#include <string.h>
#include "mcmc.h"
#include "parallel_tempering.h"
#include "define_defaults.h"

#ifndef MAX_ITERATIONS
#define MAX_ITERATIONS 0
#endif

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s calibrate_first|calibrate_rest|run|analyse\n", argv[0]);
        return 1;
    }
    if (strcmp(argv[1], "calibrate_first") == 0)
        calibrate_first();
    else if (strcmp(argv[1], "calibrate_rest") == 0)
        calibrate_rest();
    else if (strcmp(argv[1], "run") == 0)
        prepare_and_run_sampler(MAX_ITERATIONS, argc == 3 && strcmp(argv[2], "--append") == 0);
    else if (strcmp(argv[1], "analyse") == 0) {
        analyse_marginal_distributions();
        analyse_data_probability();
    }
    else
        return 1;
    return 0;
}
