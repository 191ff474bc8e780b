/* tempering_interaction() of the reference API (src/parallel_tempering_interaction.c:125-141)
 * on host chain objects: the ladder is mirrored to the device, one swap-only launch of the
 * round kernel applies the attempt, the ladder is read back.  The run phase never calls this;
 * there the swap is fused into the next round's launch. */
#include "parallel_tempering_interaction.h"
#include "parallel_tempering.h"
#include "apemost_bridge.h"

void tempering_interaction(mcmc **chains, unsigned int n_beta, unsigned long iter) {
    apemost_ladder *l;
    unsigned int k;
    (void)iter;
    assert(n_beta > 0);
    if (n_beta == 1)
        return;
    l = apemost_ladder_open(chains, n_beta);
    for (k = 0; k < apemost_ladder_shards(l); k++)
        apemost_hip_or_die(apemost_hip_set_round(apemost_ladder_shard(l, k), apemost_swap_round, 1), "set_round");
    apemost_ladder_run(l, 0, 1, NULL); /* no steps: just the pending swap attempt */
    apemost_swap_round++;
    apemost_ladder_download(l);
    apemost_ladder_close(l);
}
