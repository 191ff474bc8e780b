#ifndef PARALLEL_TEMPERING_CONFIG_H_
#define PARALLEL_TEMPERING_CONFIG_H_

#include "mcmc.h"
#include "parallel_tempering_beta.h"

#ifndef CALIBRATION_FILE
#define CALIBRATION_FILE "calibration_results"
#endif

void write_params_file(mcmc *m);
void write_calibration_summary(mcmc **chains, unsigned int n_chains);
mcmc **setup_chains();
void read_calibration_file(mcmc **chains, unsigned int n_chains);
void write_calibrations_file(mcmc **chains, const unsigned int n_chains);
/* N_BETA, or the APEMOST_N_BETA environment override (ladders beyond the compile-time value) */
unsigned int apemost_n_beta(void);

#endif
