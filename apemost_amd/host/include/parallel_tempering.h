/* Phases an application main calls (reference src/parallel_tempering.h:55-63). */
#ifndef PARALLEL_TEMPERING_H_
#define PARALLEL_TEMPERING_H_

#include "mcmc.h"
#include "parallel_tempering_beta.h"

#ifndef PRINT_PROB_INTERVAL
#define PRINT_PROB_INTERVAL 1000 /* iterations between progress/acceptance lines */
#endif
#define CALIBRATION_FILE "calibration_results"

void calibrate_first();
void prepare_and_run_sampler(unsigned long max_iterations, int append);
void calibrate_rest();
void analyse_marginal_distributions();
void analyse_data_probability();

#endif
