/*
 * apemost_oracle.h -- CPU restatement of APEMoST's parallel-tempering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under apemost_amd/, include/ or the host
 * library may include, link or call this.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and only as the checker / the timed
 * CPU baseline -- never as the product path.
 *
 * Pinning status (see oracle/README.md):
 *   pinned   : mt19937 stream (GSL rng KATs), simplesin likelihood (manual eval
 *              KAT), params/data parser fixtures, gsl_sf_log KATs, mod_double KATs.
 *   unpinned : sampler trajectories, calibration results, swap sequences -- the
 *              reference holds no fixture for them and cannot be built here
 *              (GSL absent).  "parity unpinned" for those.
 *
 * Every function cites the reference file:line (relative to the APEMoST tree)
 * whose behaviour it restates.
 */
#ifndef APEMOST_ORACLE_H
#define APEMOST_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- models (apps/<model>.c) ------------------------------------------- */
enum {
    ORC_MODEL_SIMPLESIN = 0,  /* apps/simplesin.c:12-38, n_par = 4          */
    ORC_MODEL_PULSE = 1,      /* apps/pulse.c:12-54, n_par = 2 + 2*modes    */
    ORC_MODEL_PULSE_VROT = 2, /* apps/pulse_vrot.c:12-65, n_par = 7         */
    ORC_MODEL_SINE3 = 3,      /* own 10-parameter model, SURVEY N3           */
    /* the reference's other examples, for the engine's user-supplied device models */
    ORC_MODEL_SINE2 = 4,      /* apps/simplesin2.c:12-34, n_par = 2          */
    ORC_MODEL_NORMAL = 5,     /* apps/normal.c:8-34, n_par = 1               */
    ORC_MODEL_BERNOULLI = 6   /* apps/bernoulli_example.c:10-49, n_par = n_cols */
};

/* ---- beta ladders (src/parallel_tempering_beta.c:53-83) ---------------- */
enum {
    ORC_LADDER_CHEBYSHEV_BETA = 0,
    ORC_LADDER_EQUIDISTANT_BETA = 1,
    ORC_LADDER_EQUIDISTANT_TEMPERATURE = 2,
    ORC_LADDER_CHEBYSHEV_TEMPERATURE = 3,
    ORC_LADDER_EQUIDISTANT_STEPWIDTH = 4,
    ORC_LADDER_CHEBYSHEV_STEPWIDTH = 5,
    ORC_LADDER_HOT_CHAINS = 6
};

/* ---- RNG ----------------------------------------------------------------
 * ORC_RNG_GLOBAL_MT : the reference's single process-global gsl mt19937
 *                     (src/mcmc.c:27-35), consumed in program order.
 * ORC_RNG_STREAMS   : counter-based Philox4x32-10 streams addressed the way the
 *                     device engine addresses rocRAND: stream (chain c, slot s)
 *                     is subsequence c*256+s.  Every Metropolis update of a
 *                     chain has a tick t (0,1,2,.. per chain).  Attempt q of
 *                     the proposal of parameter p at tick t is Philox block
 *                     (t<<24)|q of slot p: words 0,1 are the polar pair; an
 *                     attempt fails if a word is 0, the polar test rejects, or
 *                     the proposal leaves [min,max]; the first successful q
 *                     wins (same law as the reference's redraw loops).  The
 *                     accept uniform of tick t is word 0 of block t<<24 of slot
 *                     n_par.  The swap stream is subsequence 2^63, block = round
 *                     (word 0 pair choice, word 1 accept).
 */
enum { ORC_RNG_GLOBAL_MT = 0, ORC_RNG_STREAMS = 1 };

/* proposal distribution: default gsl_ran_gaussian, -DPROPOSAL_LOGISTIC gsl_ran_logistic(sigma),
 * -DPROPOSAL_UNIFORM gsl_ran_flat(-sigma, sigma)  (src/mcmc_gettersetter.c:290-305) */
enum { ORC_PROPOSAL_GAUSSIAN = 0, ORC_PROPOSAL_LOGISTIC = 1, ORC_PROPOSAL_UNIFORM = 2 };

typedef struct {
    uint32_t mt[624];
    int mti;
} orc_mt19937;

void orc_mt_seed(orc_mt19937 *g, unsigned long seed);
uint32_t orc_mt_next(orc_mt19937 *g);

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* n-th 32-bit output of rocRAND's philox4x32_10 engine (seed, subsequence, offset=n) */
uint32_t orc_philox_at(uint64_t seed, uint64_t subsequence, uint64_t n);

#define ORC_STREAMS_PER_CHAIN 256
#define ORC_SWAP_SUBSEQUENCE 0x8000000000000000ULL
#define ORC_TICK_SHIFT 24

typedef struct {
    int kind;            /* ORC_RNG_* */
    orc_mt19937 mt;      /* GLOBAL_MT state */
    uint64_t seed;       /* STREAMS: philox key */
    uint64_t *ticks;     /* STREAMS: [n_chain] Metropolis updates performed so far */
    uint64_t round;      /* STREAMS: swap-stream position */
    uint64_t draws;      /* statistics: total 32-bit draws consumed */
} orc_rng;

/* ---- ladder state: structure of arrays, same layout as the device ------- */
typedef struct {
    int n_chain;            /* chains held here */
    int n_par;
    int model;
    int n_data, n_cols;
    int64_t chain_offset;   /* global index of local chain 0 (sharded ladders) */
    const double *data;     /* row-major [n_data][n_cols], shared by all chains */
    double sigma;           /* SIGMA (simplesin.c:8-10), default 0.5 */
    double hmin;            /* HMIN (pulse.c:8-10), default 1e-6 */
    uint64_t circular;      /* bit p: parameter p is in -DCIRCULAR_PARAMS (markov_chain.h:34-46) */
    /* per chain */
    double *params;         /* [n_chain][n_par] */
    double *params_best;    /* [n_chain][n_par] */
    double *step;           /* [n_chain][n_par] */
    double *pmin;           /* [n_chain][n_par] */
    double *pmax;           /* [n_chain][n_par] */
    uint64_t *params_accepts; /* [n_chain][n_par] */
    uint64_t *params_rejects; /* [n_chain][n_par] */
    double *beta;           /* [n_chain] */
    double *prob;           /* [n_chain] */
    double *prior;          /* [n_chain] */
    double *prob_best;      /* [n_chain] */
    uint64_t *accept;       /* [n_chain] */
    uint64_t *reject;       /* [n_chain] */
    uint64_t *n_iter;       /* [n_chain] */
    uint64_t *swapcount;    /* [n_chain] */
    /* compile-time variants of the reference, 0 = its defaults */
    int proposal;           /* ORC_PROPOSAL_*: get_next_random_jump, src/mcmc_gettersetter.c:290-305 */
    int randomswap;         /* -DRANDOMSWAP, src/parallel_tempering_interaction.c:130-131 */
    int adapt;              /* -DADAPT, src/parallel_tempering.c:282-301 */
    double adapt_target;    /* TARGET_ACCEPTANCE_RATE, src/define_defaults.h:77-79 */
    int rwm;                /* -DRWM, src/parallel_tempering.c:268-281 + src/markov_chain.c:342-367 */
} orc_state;

/* calibration knobs (src/define_defaults.h:24-86, src/markov_chain.h:25-32) */
typedef struct {
    unsigned int burn_in_iterations; /* BURN_IN_ITERATIONS 10000 */
    double rat_limit;                /* desired_acceptance_rate argument, 0.5 */
    double target_global;            /* TARGET_ACCEPTANCE_RATE macro, 0.5 */
    double max_ar_deviation;         /* MAX_AR_DEVIATION 0.01 */
    unsigned int iter_limit;         /* ITER_LIMIT 100000 */
    double mul;                      /* MUL 0.85 */
    double adjust_step;              /* DEFAULT_ADJUST_STEP 0.5 */
    unsigned int iter_readjust;      /* ITER_READJUST 200 */
    int no_rescaling_limit;          /* NO_RESCALING_LIMIT 15 */
} orc_calib_cfg;

void orc_calib_defaults(orc_calib_cfg *c);

/* status codes of orc_calibrate (the reference calls exit(1) instead) */
enum { ORC_CALIB_OK = 0, ORC_CALIB_STEP_TOO_LARGE = 1, ORC_CALIB_ITER_LIMIT = 2 };

/* ---- API ------------------------------------------------------------------ */
/* GLOBAL_MT only: gsl_rng_uniform / gsl_ran_gaussian on the global stream */
double orc_uniform(orc_rng *r);
double orc_gaussian(orc_rng *r, double sigma);
/* STREAMS: attempt q of (global chain, slot) at tick t.  Returns 1 if the polar
 * pair is usable; then the N(0,sigma) variate is (sigma*y)*s_out */
int orc_gaussian_attempt(uint64_t seed, uint64_t chain_global, int slot, uint64_t tick, uint64_t q,
                         double *y_out, double *s_out);
/* get_next_random_jump for the three proposal laws: on the global stream, and as attempt q of a
 * tick-addressed stream (returns 0 if the law rejects the attempt's uniforms) */
double orc_jump(orc_rng *r, double sigma, int proposal);
int orc_jump_attempt(uint64_t seed, uint64_t chain_global, int slot, uint64_t tick, uint64_t q,
                     int proposal, double sigma, double *jump_out);
double orc_accept_log_uniform(uint64_t seed, uint64_t chain_global, int n_par, uint64_t tick);

double orc_loglike(int model, int n_par, const double *params, const double *data,
                   int n_data, int n_cols, double beta, double sigma, double hmin,
                   double *prior_out);
void orc_calc_model(orc_state *s, int chain);

int orc_check_accept(double prob_old, double prob_new, orc_rng *r, const orc_state *s,
                     int chain, int *drew);
void orc_markov_chain_step(orc_state *s, orc_rng *r, int chain);
void orc_markov_chain_step_for(orc_state *s, orc_rng *r, int chain, int p);
void orc_check_best(orc_state *s, int chain);
void orc_restart_from_best(orc_state *s, int chain);
void orc_reset_accept_rejects(orc_state *s, int chain);

double orc_ladder_beta(int kind, unsigned int i, unsigned int n_beta, double beta_0);
double orc_get_chain_beta(int kind, unsigned int i, unsigned int n_beta, double beta_0);
double orc_calc_beta_0(const orc_state *s, int chain, const double *stepwidth_factors);

/* returns the swapped pair index a, or -1; if `trace` is non-NULL it receives
 * {a, r, c} of the attempt */
int orc_tempering_interaction(orc_state *s, orc_rng *r, double *trace);
int orc_swap_decision(double a_beta, double b_beta, double a_prob, double b_prob,
                      double log_u, double *r_out);
int orc_swap_pair_index(double u, int n_beta);

/* Sharded ladders (the engine's multi-GPU layout): this state holds chains
 * [chain_offset, chain_offset+n_chain) of an n_global ladder.  Applies swap attempt number
 * r->round to the local chains; halo_lo / halo_hi are the edge records (beta, prob, prob_best,
 * params[n_par], params_best[n_par]) of chains chain_offset-1 and chain_offset+n_chain, needed
 * only when the chosen pair straddles that edge (else may be NULL).  Returns the pair index a. */
int orc_tempering_interaction_shard(orc_state *s, orc_rng *r, int64_t n_global, const double *halo_lo,
                                    const double *halo_hi, int *swapped_out);
/* n_steps x {markov_chain_step, check_best, n_iter++} for every local chain, no swap */
void orc_run_steps(orc_state *s, orc_rng *r, unsigned int n_steps, double *samples, int n_threads);

/* -DADAPT step-width nudging of one chain at the end of a round (orc_run_sampler calls it for
 * every chain when s->adapt is set) */
void orc_adapt(orc_state *s, int chain);
/* adapt() with -DRWM for one chain: one more markov_chain_step, then rmw_adapt_stepwidth with the log-posterior
 * the chain had before it (src/parallel_tempering.c:275-280, src/markov_chain.c:342-367) */
void orc_rwm(orc_state *s, orc_rng *r, int chain);
/* rmw_adapt_stepwidth's n_par uniforms in STREAMS mode: word p % 4 of block (tick << 24) | (1 + p / 4) of the
 * accept slot, tick = the extra step's */
double orc_rwm_uniform(uint64_t seed, uint64_t chain_global, int n_par, uint64_t tick, int p);

/* samples: [n_rounds*n_swap][n_chain][n_par+2] = params.., prob, prob-prior ; may be NULL.
 * n_threads > 1 is only meaningful with ORC_RNG_STREAMS. */
void orc_run_sampler(orc_state *s, orc_rng *r, uint64_t n_rounds, unsigned int n_swap,
                     double *samples, int n_threads);

void orc_burn_in(orc_state *s, orc_rng *r, int chain, unsigned int burn_in_iterations);
/* where orc_calibrate_orig writes the reference's calibration_progress.data
 * (src/markov_chain_calibrate.c:1052, 1141-1146); NULL = nowhere (default).  The string is kept by
 * pointer. */
void orc_set_progress_path(const char *path);
int orc_calibrate_orig(orc_state *s, orc_rng *r, int chain, const orc_calib_cfg *cfg,
                       uint64_t *iters_out);
int orc_markov_chain_calibrate(orc_state *s, orc_rng *r, int chain, const orc_calib_cfg *cfg,
                               uint64_t *iters_out);
int orc_calibrate_first(orc_state *s, orc_rng *r, const orc_calib_cfg *cfg);
int orc_calibrate_rest(orc_state *s, orc_rng *r, const orc_calib_cfg *cfg, int ladder_kind,
                       double beta_0, int skip_calibrate_allchains, int n_threads,
                       double *beta_0_out, double *stepwidth_factors_out);

double orc_mod_double(double x, double y);

#ifdef __cplusplus
}
#endif
#endif
