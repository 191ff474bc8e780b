/* Internal glue between the host chain objects (mcmc**) and the device engine's C ABI
 * (include/apemost_hip.h).  Not part of the reference API. */
#ifndef APEMOST_BRIDGE_H_
#define APEMOST_BRIDGE_H_

#include "mcmc.h"
#include "apemost_hip.h"

typedef struct apemost_ladder apemost_ladder;

/* exits with a message unless rc == APEMOST_HIP_OK: there is no CPU fallback */
void apemost_hip_or_die(int rc, const char *what);
/* which device likelihood equals the linked calc_model(); exits if none does */
int apemost_detect_model(mcmc *m);
/* device twin of chains[0..n): detects the model, uploads data and state */
apemost_ladder *apemost_ladder_open(mcmc **chains, unsigned int n_chains);
void apemost_ladder_upload(apemost_ladder *l);
void apemost_ladder_download(apemost_ladder *l);
void apemost_ladder_close(apemost_ladder *l);
apemost_hip_sampler *apemost_ladder_sampler(apemost_ladder *l);
/* cached one-chain twin used by the single-chain API (markov_chain_step & co) */
apemost_ladder *apemost_single(mcmc *m);
/* number of tempering_interaction() calls so far in this process */
extern unsigned long apemost_swap_round;

#endif
