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
apemost_hip_sampler *apemost_ladder_sampler(apemost_ladder *l); /* shard 0 */
/* shards of the ladder (APEMOST_DEVICES=0,1,...: one per listed device; default one) */
#define APEMOST_MAX_SHARDS 16
unsigned int apemost_ladder_shards(const apemost_ladder *l);
apemost_hip_sampler *apemost_ladder_shard(apemost_ladder *l, unsigned int k);
unsigned int apemost_ladder_shard_first(const apemost_ladder *l, unsigned int k); /* k = shards: n_chains */
void apemost_ladder_calc_model(apemost_ladder *l, unsigned int first, unsigned int count);
int apemost_ladder_calibrate(apemost_ladder *l, unsigned int first, unsigned int count,
                             const apemost_hip_calib_config *c, int burn_in_only, int32_t *status);
/* calibration_progress.data from the readjustment log of the latest calibration on this sampler */
void apemost_write_calibration_progress(apemost_hip_sampler *s, unsigned int n_par);
void apemost_ladder_run(apemost_ladder *l, unsigned long n_rounds, unsigned int n_swap, double **d_samples);
/* cached one-chain twin used by the single-chain API (markov_chain_step & co) */
apemost_ladder *apemost_single(mcmc *m);
/* The engine's per-chain RNG address: which stream family the chain draws from (its position in
 * the ladder) and how many Metropolis updates it has made.  Kept beside the chain objects, keyed
 * by pointer, so that `mcmc` and `parallel_tempering_mcmc` keep the reference's layout.  A chain
 * seen for the first time gets tick 0 and the next free ladder position (creation order, so runs
 * are reproducible); setup_chains() assigns positions explicitly; mcmc_free() forgets the chain. */
typedef struct {
    unsigned long tick;     /* Metropolis updates performed so far */
    unsigned long chain_id; /* position in the ladder */
} apemost_chain_address_t;
apemost_chain_address_t *apemost_chain_address(const mcmc *m);
void apemost_chain_place(const mcmc *m, unsigned long chain_id); /* (re)register at a ladder position, tick 0 */
void apemost_chain_forget(const mcmc *m);
/* number of tempering_interaction() calls so far in this process */
extern unsigned long apemost_swap_round;

#endif
