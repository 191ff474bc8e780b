/*
 * apemost_hip.h -- C ABI of the MI355X (gfx950) parallel-tempering engine.
 *
 * This is the drop-in boundary for APEMoST's hot path: the body of
 * run_sampler() (src/parallel_tempering.c:347-419), i.e. per chain
 * markov_chain_step() (src/markov_chain.c:369-386) + mcmc_check_best()
 * (src/mcmc_calculate.c:35-41) + sample output, then tempering_interaction()
 * (src/parallel_tempering_interaction.c:125-141); and the calibration phases
 * built from the same step (src/markov_chain.c:34-79,
 * src/markov_chain_calibrate.c:1039-1204, src/parallel_tempering.c:78-207).
 *
 * Plain C: pointers, sizes, integers.  No torch / C++ types.  A reference
 * maintainer binds it directly from C (INTEGRATION.md shows the patch to
 * src/parallel_tempering.c); the Python host mirror binds it with ctypes.
 *
 * All chain state crosses the boundary as structure-of-arrays blocks
 * (apemost_hip_state_view) -- the flattened form of the reference's `mcmc`
 * struct (src/mcmc_struct.h:30-106) plus `parallel_tempering_mcmc`
 * (src/parallel_tempering_beta.h:65-76).
 *
 * Every function returns APEMOST_HIP_OK (0) or a negative error code;
 * apemost_hip_last_error() gives the message.  There is no CPU fallback: with
 * no usable HIP device every compute entry point fails with
 * APEMOST_HIP_ERR_NO_DEVICE.
 */
#ifndef APEMOST_HIP_H
#define APEMOST_HIP_H

#if !defined(__HIPCC_RTC__)
#include <stddef.h>
#include <stdint.h>
#else /* compiled by hiprtc (a user-supplied device model): no C library headers, its own fixed-width types */
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define APEMOST_HIP_ABI_VERSION 3

enum {
    APEMOST_HIP_OK = 0,
    APEMOST_HIP_ERR_INVALID = -1,   /* bad argument / shape mismatch */
    APEMOST_HIP_ERR_NO_DEVICE = -2, /* no HIP device, or not gfx950 */
    APEMOST_HIP_ERR_RUNTIME = -3,   /* a HIP call failed */
    APEMOST_HIP_ERR_UNSUPPORTED = -4,
    APEMOST_HIP_ERR_CALIBRATION = -5 /* calibration failed (reference: exit(1)) */
};

/* device-side likelihoods = re-implementations of the user plugins calc_model()
 * (src/mcmc.h:164) of the BASELINE apps */
enum {
    APEMOST_MODEL_SIMPLESIN = 0,  /* apps/simplesin.c:12-38   n_par = 4            */
    APEMOST_MODEL_PULSE = 1,      /* apps/pulse.c:12-54       n_par = 2 + 2*modes  */
    APEMOST_MODEL_PULSE_VROT = 2, /* apps/pulse_vrot.c:12-65  n_par = 7            */
    APEMOST_MODEL_SINE3 = 3,      /* 10-parameter 3-sinusoid model  */
    /* any other calc_model(): device source supplied by the user (apemost_device_model.h), named in
     * apemost_hip_config.device_model_source and compiled with hiprtc when the sampler is created;
     * any n_par; runs in the one-wave kernels */
    APEMOST_MODEL_USER = 4
};

/* BETA_ALIGNMENT choices, src/parallel_tempering_beta.c:53-83 */
enum {
    APEMOST_LADDER_CHEBYSHEV_BETA = 0,
    APEMOST_LADDER_EQUIDISTANT_BETA = 1,
    APEMOST_LADDER_EQUIDISTANT_TEMPERATURE = 2,
    APEMOST_LADDER_CHEBYSHEV_TEMPERATURE = 3,
    APEMOST_LADDER_EQUIDISTANT_STEPWIDTH = 4,
    APEMOST_LADDER_CHEBYSHEV_STEPWIDTH = 5,
    APEMOST_LADDER_HOT_CHAINS = 6
};

#define APEMOST_HIP_MAX_PAR 62            /* n_par + 2 lanes of one wavefront */
/* RNG addressing (rocRAND Philox4x32-10): stream (chain, slot) = subsequence chain*256+slot;
 * attempt q of parameter p's proposal at tick t = block (t<<24)|q of slot p, words 0,1;
 * accept uniform of tick t = word 0 of block t<<24 of slot n_par; swap attempt number r =
 * block r of the swap subsequence (word 0 pair choice, word 1 accept; with
 * APEMOST_HIP_FLAG_RANDOMSWAP words 1 and 2, word 0 being the swap_probability draw).
 * Non-Gaussian proposals read word 0 of the attempt's block. */
#define APEMOST_HIP_STREAMS_PER_CHAIN 256
#define APEMOST_HIP_TICK_SHIFT 24
#define APEMOST_HIP_SWAP_SUBSEQUENCE 0x8000000000000000ULL

typedef struct apemost_hip_sampler apemost_hip_sampler;

/* apemost_hip_config.flags */
enum {
    /* one round per launch: every swap attempt is the fused swap-in at the start of the next
     * launch; no workgroup ever waits for another one (the fallback the engine also takes by
     * itself when the grid cannot be co-resident, or after a failed in-launch hand-off) */
    APEMOST_HIP_FLAG_SINGLE_ROUND_LAUNCHES = 1,
    /* multi-round launches go through hipLaunchCooperativeKernel: the runtime guarantees (or
     * refuses) co-residency of the whole grid instead of the engine's occupancy estimate.  The
     * engine does so by itself where the grid fits the occupancy figure but not its estimate; a
     * refused launch is re-issued one round at a time, as are all later ones */
    APEMOST_HIP_FLAG_COOPERATIVE_LAUNCH = 2,
    /* stepping launches of workgroups with 8 likelihood waves use the classic two-phase step
     * (two barriers, the chain's wave alone between them) instead of the one-barrier kernel; same
     * chain either way (A/B comparisons, fallback) */
    APEMOST_HIP_FLAG_TWO_BARRIER_STEP = 4,
    /* the reference's compile-time variants (defaults: Gaussian proposal, decide_swap_now, no
     * adaptation).  At most one of the two proposal bits.  The proposal and swap variants run in
     * kernel instantiations of their own (the default kernels carry no test for them), built for
     * 1, 2, 4 and 8 waves per chain. */
    /* -DPROPOSAL_LOGISTIC: get_next_random_jump = gsl_ran_logistic(r, step),
     * src/mcmc_gettersetter.c:291-292 (uniform of word 0 of the attempt's block) */
    APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC = 8,
    /* -DPROPOSAL_UNIFORM: gsl_ran_flat(r, -step, step), src/mcmc_gettersetter.c:293-294 */
    APEMOST_HIP_FLAG_PROPOSAL_UNIFORM = 16,
    /* -DRANDOMSWAP: tempering_interaction() goes through
     * parallel_tempering_decide_swap_random(chains, n_beta, 1)
     * (src/parallel_tempering_interaction.c:47-64, 130-131): one more uniform ahead of the pair
     * choice -- swap attempt r reads words 0 (swap_probability), 1 (pair), 2 (accept) of block r */
    APEMOST_HIP_FLAG_RANDOMSWAP = 32,
    /* -DADAPT: adapt() after the steps of every round of a stepping launch
     * (src/parallel_tempering.c:282-301, 404) nudges the chain's step widths by 0.99 or 1/0.99
     * towards apemost_hip_config.adapt_target */
    APEMOST_HIP_FLAG_ADAPT = 64,
    /* test hook: every cooperative launch is treated as refused by the runtime, so that the path a
     * real refusal takes (the launch re-issued round by round, single-round launches from then on)
     * can be exercised on a machine where the runtime never refuses */
    APEMOST_HIP_FLAG_TEST_REFUSE_COOPERATIVE = 128,
    /* test hook for the other designed failure: inside a multi-round launch the lower chain of the
     * pair of swap attempt 3 does not publish its record, and every bounded wait of the launch gives
     * up after a few thousand polls instead of eight million -- the partner's wait runs out, the
     * launch's error word is raised, apemost_hip_synchronize reports it, and the sampler issues one
     * round per launch from then on */
    APEMOST_HIP_FLAG_TEST_WITHHOLD_PUBLISH = 256,
    /* -DRWM: adapt()'s other block (src/parallel_tempering.c:268-281; it does not compile in the reference:
     * `markov_chain_step(chains[i], 0)` passes two arguments to a function of one).  As restated here: after
     * the steps of every round and before its swap attempt every chain keeps its log-posterior, takes ONE
     * more markov_chain_step -- accept / reject counters and the RNG tick move, but no mcmc_check follows it:
     * no sample row, n_iter and the best point stay what the round's own steps left -- and
     * rmw_adapt_stepwidth (src/markov_chain.c:342-367) moves every step width by
     * U / sqrt(n_iter) * (min(1, exp(prob - prob_old)) - adapt_target) * (max - min), clamped to
     * [1e-7, 1e6] * (max - min); U = word p % 4 of block (tick << 24) | (1 + p / 4) of the accept slot.
     * Every round is a launch of its own (as with APEMOST_HIP_FLAG_ADAPT, which runs after it). */
    APEMOST_HIP_FLAG_RWM = 512
};

typedef struct {
    int32_t abi_version;     /* APEMOST_HIP_ABI_VERSION */
    int32_t device;          /* HIP device ordinal */
    int32_t model;           /* APEMOST_MODEL_* */
    int32_t n_par;           /* get_n_par(), src/mcmc_gettersetter.c:174-183 */
    int32_t n_chains;        /* chains resident on this device (a shard of the ladder) */
    int32_t n_data;          /* m->data->size1 */
    int32_t n_cols;          /* m->data->size2 (>= 2) */
    int32_t waves_per_chain; /* likelihood wavefronts per chain: 1, 2, 4, 6 or 8; 0 = choose */
    int32_t lds_policy;      /* data vector staged in LDS: 0 = choose, 1 = always (if it fits), 2 = never */
    int32_t flags;           /* APEMOST_HIP_FLAG_* bits, 0 = defaults */
    int64_t chain_offset;    /* ladder index of local chain 0 */
    int64_t n_chains_global; /* N_BETA, src/define_defaults.h:24-31 */
    uint64_t seed;           /* rocRAND Philox4x32-10 seed (role of GSL_RNG_SEED) */
    double sigma;            /* SIGMA, apps/simplesin.c:8-10 */
    double hmin;             /* HMIN, apps/pulse.c:8-10 */
    uint64_t circular_params; /* bit p set: parameter p wraps around [min,max] instead of being
                               * redrawn (-DCIRCULAR_PARAMS, src/markov_chain.c:241-265) */
    double adapt_target;      /* TARGET_ACCEPTANCE_RATE (src/define_defaults.h:77-79) for
                               * APEMOST_HIP_FLAG_ADAPT; 0 = the reference's default 0.5 */
    const char *device_model_source; /* APEMOST_MODEL_USER: path of the device source of the likelihood
                                      * (include/apemost_device_model.h); NULL otherwise */
} apemost_hip_config;

/* Host-side structure-of-arrays view of n_chains chains; any pointer may be NULL
 * (that field is skipped).  Shapes: [n_chains][n_par] or [n_chains]. */
typedef struct {
    double *params;           /* m->params */
    double *params_best;      /* m->params_best */
    double *step;             /* m->params_step */
    double *pmin;             /* m->params_min */
    double *pmax;             /* m->params_max */
    uint64_t *params_accepts; /* m->params_accepts */
    uint64_t *params_rejects; /* m->params_rejects */
    double *beta;             /* parallel_tempering_mcmc.beta */
    double *prob;             /* m->prob */
    double *prior;            /* m->prior */
    double *prob_best;        /* m->prob_best */
    uint64_t *accept;         /* m->accept */
    uint64_t *reject;         /* m->reject */
    uint64_t *n_iter;         /* m->n_iter */
    uint64_t *swapcount;      /* parallel_tempering_mcmc.swapcount */
    uint64_t *ticks;          /* [n_chains] Metropolis updates performed = RNG address of the next one */
} apemost_hip_state_view;

/* calibration knobs: src/define_defaults.h:24-86, src/markov_chain.h:25-32 */
typedef struct {
    uint32_t burn_in_iterations; /* BURN_IN_ITERATIONS */
    uint32_t iter_limit;         /* ITER_LIMIT */
    uint32_t iter_readjust;      /* ITER_READJUST */
    int32_t no_rescaling_limit;  /* NO_RESCALING_LIMIT */
    double rat_limit;            /* desired_acceptance_rate argument */
    double target_global;        /* TARGET_ACCEPTANCE_RATE */
    double max_ar_deviation;     /* MAX_AR_DEVIATION */
    double mul;                  /* MUL */
    double adjust_step;          /* DEFAULT_ADJUST_STEP */
    int32_t progress_chain;      /* local chain whose readjustments are logged for calibration_progress.data
                                  * (src/markov_chain_calibrate.c:1143-1146; the reference reopens that file
                                  * "w" for every chain, so the last chain calibrated is the one whose lines
                                  * survive), or -1 */
    int32_t reserved;            /* 0 */
} apemost_hip_calib_config;

/* ---- environment ---------------------------------------------------------- */
const char *apemost_hip_last_error(void);
int apemost_hip_abi_version(void);
int apemost_hip_device_count(int *count);
/* name[] receives the gcnArchName; fails unless it is a gfx950 part */
int apemost_hip_device_info(int device, char *name, size_t name_len, int *compute_units,
                            uint64_t *hbm_bytes);

/* ---- lifetime ------------------------------------------------------------- */
int apemost_hip_create(const apemost_hip_config *cfg, apemost_hip_sampler **out);
int apemost_hip_destroy(apemost_hip_sampler *s);
int apemost_hip_synchronize(apemost_hip_sampler *s);
/* the HIP stream (hipStream_t) every launch of this sampler goes to */
int apemost_hip_stream(apemost_hip_sampler *s, void **stream);
int apemost_hip_waves_per_chain(apemost_hip_sampler *s, int *waves, int *data_in_lds);
/* how this sampler's stepping launches are issued as things stand: the one-barrier kernel (1; 2: in its form
 * with a helper wavefront, the models with a prior on ladders of at most one chain per CU) or the
 * two-phase one (0), multi-round launches through hipLaunchCooperativeKernel or plain, and how many
 * rounds one launch may hold (1: the grid is not resident, a cooperative launch was refused, or a
 * hand-off timed out) */
int apemost_hip_launch_policy(apemost_hip_sampler *s, int32_t *one_barrier, int32_t *cooperative, int32_t *max_rounds);
/* APEMOST_MODEL_USER: seconds hiprtc took to compile this sampler's device model (the kernels of the
 * user's likelihood for 1, 2, 4 and 8 waves per chain); 0 when the process had compiled the same source
 * before.  Fails for the built-in models. */
int apemost_hip_user_model_compile_seconds(apemost_hip_sampler *s, double *seconds);
/* move the shard along the ladder (single-chain API of the C host layer: the chain's ladder
 * position selects its RNG streams); offset + n_chains must stay <= n_chains_global */
int apemost_hip_set_chain_offset(apemost_hip_sampler *s, int64_t chain_offset);

/* ---- data and state (mcmc_load_data / setup_chains / read_calibration_file) */
/* row-major [n_data][n_cols] host matrix, as gsl_matrix stores it */
int apemost_hip_set_data(apemost_hip_sampler *s, const double *data_rowmajor);
int apemost_hip_set_state(apemost_hip_sampler *s, const apemost_hip_state_view *v);
int apemost_hip_get_state(apemost_hip_sampler *s, const apemost_hip_state_view *v);
/* position of the swap stream = number of tempering_interaction() calls so far */
int apemost_hip_set_round(apemost_hip_sampler *s, uint64_t round, int swap_pending);
int apemost_hip_get_round(apemost_hip_sampler *s, uint64_t *round, int *swap_pending);

/* ---- the hot path --------------------------------------------------------- */
/* calc_model() for local chains [first, first+count) at their current params: prob,
 * prior updated on device (src/parallel_tempering.c:88,147,185 call sites);
 * count < 0 means every chain from `first` on */
int apemost_hip_calc_model(apemost_hip_sampler *s, int32_t first, int32_t count);

/* calc_model() at arbitrary points (apps/eval_main.c:52-66): n points, params
 * [n][n_par], beta [n]; results to host arrays prob[n], prior[n] */
int apemost_hip_loglike(apemost_hip_sampler *s, int32_t n, const double *params, const double *beta,
                        double *prob, double *prior);

/* One launch of the round kernel: optionally apply the pending swap attempt
 * (tempering_interaction for swap-stream position `round`), then n_steps x
 * {markov_chain_step, mcmc_check_best, n_iter++, sample row}.  d_samples is a
 * DEVICE pointer to [n_steps][n_chains][n_par+2] doubles (params.., prob,
 * prob-prior; the rows the reference prints to <name>-chain-<i>.prob.dump and
 * prob-chain<i>.dump) or NULL.  Asynchronous on the sampler's stream.
 * Sharded ladders call apemost_hip_edge_* around it; whole ladders use
 * apemost_hip_run. */
int apemost_hip_launch_round(apemost_hip_sampler *s, uint32_t n_steps, int apply_swap,
                             double *d_samples);

/* Several rounds in one launch: [pending swap] steps [swap] steps ... (n_rounds x n_steps steps,
 * n_rounds-1 swap attempts exchanged between workgroups inside the launch).  d_samples:
 * DEVICE [n_rounds*n_steps][n_chains][n_par+2] or NULL.  n_rounds may not exceed
 * apemost_hip_max_rounds_per_launch() (1 when the grid cannot be fully resident), and on a sharded
 * ladder none of the in-launch swap attempts may pick a pair that straddles a shard edge. */
int apemost_hip_launch_rounds(apemost_hip_sampler *s, uint32_t n_rounds, uint32_t n_steps, int apply_swap,
                              double *d_samples);
int apemost_hip_max_rounds_per_launch(apemost_hip_sampler *s, int32_t *max_rounds);

/* n_steps x markov_chain_step_for(m, param) (src/markov_chain.c:317-333) for every resident
 * chain: only parameter `param` is proposed, only its counters move */
int apemost_hip_launch_round_for(apemost_hip_sampler *s, uint32_t n_steps, int32_t param, double *d_samples);

/* run_sampler() for a ladder that lives entirely on this device: n_rounds x
 * {n_swap steps, swap attempt}; the last swap is applied before returning
 * control (still asynchronous).  d_samples: DEVICE [n_rounds*n_swap][n_chains][n_par+2] or NULL */
int apemost_hip_run(apemost_hip_sampler *s, uint64_t n_rounds, uint32_t n_swap, double *d_samples);

/* device buffers for sample rows, for hosts without their own device allocator (the C host
 * layer): [n_steps][n_chains][n_par+2] doubles */
int apemost_hip_samples_alloc(apemost_hip_sampler *s, uint64_t n_steps, double **d_samples);
int apemost_hip_samples_read(apemost_hip_sampler *s, const double *d_samples, uint64_t n_steps, double *host);
int apemost_hip_samples_free(apemost_hip_sampler *s, double *d_samples);

/* The same read without stalling the sampler: the copy is queued behind everything launched so far
 * but runs on a second stream, so launches issued after this call overlap with it (double-buffered
 * sample sinks: the device fills buffer B while buffer A drains to the host).  host_samples should
 * be pinned memory (apemost_hip_host_alloc).  counters, if not NULL, receives [2][n_chains]
 * uint64: m->accept then m->reject of every chain as they stand after the launches issued so far
 * (what the reference prints to acceptance_rate.dump, src/parallel_tempering.c:320-326), without a
 * host synchronisation of the sampler's stream.  apemost_hip_samples_wait blocks until the latest
 * such read has landed (and reports a void launch like apemost_hip_samples_read). */
int apemost_hip_samples_read_async(apemost_hip_sampler *s, const double *d_samples, uint64_t n_steps,
                                   double *host_samples, uint64_t *counters);
/* The same with the rows packed ON THE DEVICE into what the sink will write, so that only that crosses
 * PCIe and the host writes the pinned buffer as it is: of the n_steps rows the steps skip, skip + thin,
 * ... are kept (*n_kept of them);
 *   layout 0: per kept step the parameter vectors of chains 0 .. n_param_chains-1, then
 *             (prob, prob - prior) of every chain  -- the record of the C host's binary sink;
 *   layout 1: the kept rows themselves, [n_kept][n_chains][n_par+2].
 * d_packed: DEVICE scratch for the packed batch (apemost_hip_samples_alloc sizes fit), host_packed:
 * pinned; the wait as for apemost_hip_samples_read_async; counters, if not NULL, receives [2][n_chains]
 * uint64 as there and BEHIND them n_par doubles: chain 0's parameter vector after the batch's last step
 * (the reference's progress line prints chain 0's current point, src/parallel_tempering.c:309-318; a
 * packed batch may have kept an older step, or none) -- (2 n_chains + n_par) * 8 bytes. */
int apemost_hip_samples_pack_read_async(apemost_hip_sampler *s, const double *d_samples, uint64_t n_steps,
                                        uint64_t skip, uint64_t thin, int32_t n_param_chains, int32_t layout,
                                        double *d_packed, double *host_packed, uint64_t *counters, uint64_t *n_kept);
int apemost_hip_samples_wait(apemost_hip_sampler *s);
/* page-locked host memory for those reads */
int apemost_hip_host_alloc(size_t bytes, void **p);
int apemost_hip_host_free(void *p);

/* pair index `a` that tempering_interaction() will pick at swap-stream position
 * `round` (parallel_tempering_decide_swap_now, interaction.c:87-97); -1 if n_global==1 */
int64_t apemost_hip_swap_pair(uint64_t seed, uint64_t round, int64_t n_chains_global);
/* the same for this sampler's ladder and swap schedule (with APEMOST_HIP_FLAG_RANDOMSWAP the pair
 * comes from word 1 of the block) */
int64_t apemost_hip_sampler_swap_pair(const apemost_hip_sampler *s, uint64_t round);
/* how many of the swap attempts first_round, first_round + 1, ... (at most max_rounds) pick a pair
 * that lies inside this sampler's shard or outside it altogether, i.e. stops at the first pair that
 * straddles one of the shard's edges: the rounds a sharded ladder may put into one launch */
int64_t apemost_hip_rounds_within_shard(const apemost_hip_sampler *s, uint64_t first_round, int64_t max_rounds);

/* sharded ladders: the swap partner across a shard edge.  side 0 = lower
 * neighbour (chain_offset-1), 1 = upper neighbour.  A record is
 * apemost_hip_edge_doubles(n_par) doubles: beta, prob, prob_best,
 * params[n_par], params_best[n_par].  d_buf is a DEVICE pointer (what RCCL
 * sends/receives).  export packs this shard's edge chain; import fills the
 * halo slot the next launch_round(apply_swap=1) reads. */
int32_t apemost_hip_edge_doubles(int32_t n_par);
int apemost_hip_edge_export(apemost_hip_sampler *s, int side, double *d_buf);
int apemost_hip_edge_import(apemost_hip_sampler *s, int side, const double *d_buf);

/* ---- one process, several devices (or several shards on one device) --------------------------
 * The same block-partitioned ladder as the one-process-per-GPU driver, for hosts that
 * stay a single process (the C host layer with APEMOST_DEVICES=0,1,...): shards[j] holds chains
 * [offset_j, offset_j + n_j) of the same ladder (same seed, same n_chains_global, offsets
 * ascending and contiguous), each on its own device and stream.
 *
 * apemost_hip_edge_exchange: the pending swap attempt picked the pair that straddles lower|upper:
 * both shards export their edge record, the records cross with hipMemcpyPeerAsync (xGMI between
 * devices, a device copy within one), and land in the halo rows the next launch reads -- all
 * stream-ordered with events, no host synchronisation.
 *
 * apemost_hip_run_shards: run_sampler() over all shards in lock step: rounds are batched into
 * multi-round launches up to the next swap attempt that needs a neighbour's record, exactly as
 * apemost_hip_run does on one device; d_samples[j] (may be NULL) receives shard j's rows,
 * [n_rounds*n_swap][n_j][n_par+2].  The result is bit-identical to the whole ladder on one
 * sampler. */
int apemost_hip_edge_exchange(apemost_hip_sampler *lower, apemost_hip_sampler *upper);
int apemost_hip_run_shards(apemost_hip_sampler **shards, int32_t n_shards, uint64_t n_rounds, uint32_t n_swap,
                           double **d_samples);

/* ---- calibration ---------------------------------------------------------- */
void apemost_hip_calib_defaults(apemost_hip_calib_config *c);
/* markov_chain_calibrate() (burn_in + calibrate_orig) on device for local chains
 * [first, first+count); status[count] (host) receives 0 / 1 step too large /
 * 2 iteration limit per chain, iters[count] the calibrate_orig sweep counts.
 * With burn_in_only != 0 only burn_in() runs (-DSKIP_CALIBRATE_ALLCHAINS). */
int apemost_hip_calibrate_chains(apemost_hip_sampler *s, int32_t first, int32_t count,
                                 const apemost_hip_calib_config *c, int burn_in_only,
                                 int32_t *status, uint64_t *iters);

/* The same in pieces.  The calibration runs as a sequence of launches ("segments"): a chain's trip
 * count is data-dependent, so every segment ends after a bounded number of likelihood evaluations per
 * chain, the chains that are done drop out, and the survivors are launched again -- with more
 * wavefronts per chain as they get fewer.  Results do not depend on where the segments are cut.
 *   begin  launches the first segment (asynchronous);
 *   poll   never blocks: if the segment in flight has ended, collects it and launches the next one;
 *          *active = chains still calibrating.  Hosts that drive several devices from one thread poll
 *          them in turn, and a host with a SIGINT handler polls between looks at its flag;
 *   cancel no further segments: end then returns with status -1 for the chains that were not done
 *          (their state is a consistent point of their calibration);
 *   end    polls and waits until no chain is left, then hands out status/iters of the chains begin named. */
int apemost_hip_calibrate_begin(apemost_hip_sampler *s, int32_t first, int32_t count,
                                const apemost_hip_calib_config *c, int burn_in_only);
int apemost_hip_calibrate_poll(apemost_hip_sampler *s, int32_t *active);
int apemost_hip_calibrate_cancel(apemost_hip_sampler *s);
/* several samplers (one per device) calibrating at once, driven by one thread: returns when a segment
 * of one of them has ended and been followed up, or none has chains left; *active_total = chains
 * still calibrating over all of them */
int apemost_hip_calibrate_wait_any(apemost_hip_sampler **samplers, int32_t n, int32_t *active_total);
int apemost_hip_calibrate_end(apemost_hip_sampler *s, int32_t *status, uint64_t *iters);
/* the readjustment log of apemost_hip_calib_config.progress_chain after calibrate_end: row k =
 * { iter, then per parameter (normalised step width, acceptance rate) } as the reference prints them
 * after its k-th readjustment; rows [capacity_rows][1 + 2 n_par], *n_rows = rows that exist */
int apemost_hip_calibrate_progress(apemost_hip_sampler *s, double *rows, int32_t capacity_rows, int32_t *n_rows);
/* the latest calibration in numbers: segments launched, likelihood evaluations of all its chains, and
 * launches and wall seconds per workgroup shape (index = likelihood wavefronts per chain) */
int apemost_hip_calibrate_stats(apemost_hip_sampler *s, uint64_t *segments, uint64_t *evaluations,
                                uint64_t launches_by_waves[9], double seconds_by_waves[9]);

/* ---- test hooks: device RNG conformance ------------------------------------ */
/* n raw 32-bit outputs of rocRAND philox4x32_10 (seed, subsequence, offset) */
int apemost_hip_rng_raw(int device, uint64_t seed, uint64_t subsequence, uint64_t offset, int32_t n,
                        uint32_t *out);
/* attempts q0..q0+n-1 of the proposal stream (chain, slot) at `tick`, exactly as the
 * step kernel evaluates them: valid[i] says whether the polar pair is usable, and then
 * the N(0,sigma) variate is (sigma*y[i])*s[i]; accept_log_u = ln(uniform) of the accept
 * test of that tick when slot == n_par */
int apemost_hip_rng_attempts(int device, uint64_t seed, uint64_t chain, int32_t slot, uint64_t tick,
                             uint64_t q0, int32_t n, double *y, double *s, int32_t *valid,
                             double *accept_log_u);

/* ---- timing ---------------------------------------------------------------- */
/* HIP events on the sampler's stream: begin/end bracket a region; elapsed ms and
 * the number of round-kernel launches inside it */
int apemost_hip_timer_begin(apemost_hip_sampler *s);
int apemost_hip_timer_end(apemost_hip_sampler *s, float *elapsed_ms, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* APEMOST_HIP_H */
