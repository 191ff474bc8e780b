#ifndef TEMPERING_INTERACTION_H_
#define TEMPERING_INTERACTION_H_

#include "mcmc.h"

/* one neighbour-swap attempt for the ladder (iter is informational) */
void tempering_interaction(mcmc **chains, unsigned int n_beta, unsigned long iter);

#endif
