/* Bridge: host chain objects <-> device engine (see apemost_bridge.h). */
#include <math.h>
#include <string.h>
#include "apemost_bridge.h"
#include "parallel_tempering_beta.h"
#include "markov_chain.h"
#include "debug.h"

#ifndef SIGMA
#define SIGMA 0.5
#endif
#ifndef HMIN
#define HMIN 1e-6
#endif

unsigned long apemost_swap_round = 0;

/* A ladder on the device(s): one sampler per shard.  With APEMOST_DEVICES=0,1,... the ladder is
 * block-partitioned over those devices (chain i on shard floor(i*G/n)); a device named
 * twice carries two shards (how the tests run it on one GPU).  Without it: one shard on
 * APEMOST_DEVICE (default 0). */
struct apemost_ladder {
    mcmc **chains;
    unsigned int n, n_par;
    unsigned int n_shards;
    apemost_hip_sampler *s[APEMOST_MAX_SHARDS];
    unsigned int lo[APEMOST_MAX_SHARDS + 1]; /* shard k holds chains [lo[k], lo[k+1]) */
    apemost_hip_state_view v; /* host staging arrays for the whole ladder, structure of arrays */
};

void apemost_hip_or_die(int rc, const char *what) {
    if (rc == APEMOST_HIP_OK)
        return;
    fflush(stdout);
    fprintf(stderr, "APEMoST MI355X engine: %s failed (%d): %s\n", what, rc, apemost_hip_last_error());
    fprintf(stderr, "This build has no CPU path for the sampler; a gfx950 GPU is required.\n");
    exit(1);
}

static unsigned long env_seed(void) {
    const char *s = getenv("APEMOST_SEED");
    if (s == NULL)
        s = getenv("GSL_RNG_SEED");
    return s ? strtoul(s, NULL, 0) : 0;
}

static parallel_tempering_mcmc *pt(const mcmc *m) { return (parallel_tempering_mcmc *)m->additional_data; }

/* ---- side table: mcmc pointer -> RNG address (open addressing, linear probing, tombstones) ---- */
struct address_slot {
    const mcmc *key; /* NULL = never used, TOMBSTONE = forgotten */
    apemost_chain_address_t a;
};
static struct address_slot *address_table = NULL;
static size_t address_cap = 0, address_used = 0; /* used counts live keys and tombstones */
static unsigned long address_next_id = 0;
static const mcmc *const TOMBSTONE = (const mcmc *)&address_table;

static size_t address_hash(const mcmc *m, size_t cap) {
    size_t x = (size_t)m;
    x ^= x >> 17;
    x *= (size_t)0x9E3779B97F4A7C15ul;
    x ^= x >> 29;
    return x & (cap - 1);
}

static struct address_slot *address_find(const mcmc *m, int create) {
    size_t i;
    struct address_slot *grave = NULL;
    if (address_cap == 0 || (create && 2 * (address_used + 1) > address_cap)) {
        /* grow (or start): re-insert the live entries, dropping tombstones */
        const size_t old_cap = address_cap, cap = old_cap ? 2 * old_cap : 64;
        struct address_slot *old = address_table;
        if (!create && old_cap == 0)
            return NULL;
        address_table = (struct address_slot *)calloc(cap, sizeof(struct address_slot));
        address_cap = cap;
        address_used = 0;
        for (i = 0; i < old_cap; i++)
            if (old[i].key != NULL && old[i].key != TOMBSTONE) {
                size_t j = address_hash(old[i].key, cap);
                while (address_table[j].key != NULL)
                    j = (j + 1) & (cap - 1);
                address_table[j] = old[i];
                address_used++;
            }
        free(old);
    }
    for (i = address_hash(m, address_cap);; i = (i + 1) & (address_cap - 1)) {
        struct address_slot *s = &address_table[i];
        if (s->key == m)
            return s;
        if (s->key == TOMBSTONE && grave == NULL)
            grave = s;
        if (s->key == NULL) {
            if (!create)
                return NULL;
            if (grave != NULL)
                s = grave;
            else
                address_used++;
            s->key = m;
            s->a.tick = 0;
            s->a.chain_id = address_next_id++;
            return s;
        }
    }
}

apemost_chain_address_t *apemost_chain_address(const mcmc *m) { return &address_find(m, 1)->a; }

void apemost_chain_place(const mcmc *m, unsigned long chain_id) {
    apemost_chain_address_t *a = apemost_chain_address(m);
    a->tick = 0;
    a->chain_id = chain_id;
    if (chain_id >= address_next_id)
        address_next_id = chain_id + 1;
}

void apemost_chain_forget(const mcmc *m) {
    struct address_slot *s = address_find(m, 0);
    if (s != NULL)
        s->key = TOMBSTONE;
}

/* -DCIRCULAR_PARAMS=1,2,...: 1-based indices of the parameters that wrap around their range
 * (reference src/markov_chain.h:34-46); the default single 0 means none */
static uint64_t circular_mask(void) {
    static const unsigned int circular[] = {CIRCULAR_PARAMS, 0};
    uint64_t mask = 0;
    unsigned int j;
    for (j = 0; circular[j] != 0; j++)
        mask |= (uint64_t)1 << (circular[j] - 1);
    return mask;
}

static int default_device(void) {
    const char *dev = getenv("APEMOST_DEVICE");
    return dev ? atoi(dev) : 0;
}

/* run_ladder: the sampler of a phase that goes through run_sampler().  The reference calls adapt()
 * (-DADAPT) from run_sampler's loop only (src/parallel_tempering.c:404): the one-chain twin behind
 * markov_chain_step() & co and the sampler that identifies the likelihood never adapt. */
static apemost_hip_sampler *create_sampler(const mcmc *m, int model, unsigned int n_chains, long chain_offset,
                                           long n_global, int device, int run_ladder) {
    apemost_hip_config cfg;
    apemost_hip_sampler *s = NULL;
    const char *waves = getenv("APEMOST_WAVES"), *flags = getenv("APEMOST_FLAGS");
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = APEMOST_HIP_ABI_VERSION;
    cfg.device = device;
    cfg.flags = flags ? atoi(flags) : 0;
    cfg.model = model;
    cfg.n_par = (int)m->n_par;
    cfg.n_chains = (int)n_chains;
    cfg.n_data = (int)m->data->size1;
    cfg.n_cols = (int)m->data->size2;
    cfg.waves_per_chain = waves ? atoi(waves) : 0;
    cfg.lds_policy = getenv("APEMOST_LDS") ? atoi(getenv("APEMOST_LDS")) : 0;
    cfg.chain_offset = chain_offset;
    cfg.n_chains_global = n_global;
    cfg.seed = env_seed();
    cfg.sigma = SIGMA;
    cfg.hmin = HMIN;
    cfg.circular_params = circular_mask();
    /* the reference's compile-time variants of the proposal, the swap schedule and the run loop */
#ifdef PROPOSAL_LOGISTIC
    cfg.flags |= APEMOST_HIP_FLAG_PROPOSAL_LOGISTIC;
#endif
#ifdef PROPOSAL_UNIFORM
    cfg.flags |= APEMOST_HIP_FLAG_PROPOSAL_UNIFORM;
#endif
#ifdef RANDOMSWAP
    cfg.flags |= APEMOST_HIP_FLAG_RANDOMSWAP;
#endif
#ifdef ADAPT
    if (run_ladder)
        cfg.flags |= APEMOST_HIP_FLAG_ADAPT;
#endif
#ifdef RWM
    if (run_ladder)
        cfg.flags |= APEMOST_HIP_FLAG_RWM;
#endif
    (void)run_ladder;
    cfg.adapt_target = TARGET_ACCEPTANCE_RATE;
    cfg.device_model_source = model == APEMOST_MODEL_USER ? getenv("APEMOST_DEVICE_MODEL_SRC") : NULL;
    apemost_hip_or_die(apemost_hip_create(&cfg, &s), "apemost_hip_create");
    if (m->data->tda != m->data->size2) {
        fprintf(stderr, "data matrix must be contiguous\n");
        exit(1);
    }
    apemost_hip_or_die(apemost_hip_set_data(s, m->data->data), "apemost_hip_set_data");
    return s;
}

static int model_fits(int model, unsigned int n_par) {
    switch (model) {
    case APEMOST_MODEL_SIMPLESIN:
        return n_par == 4;
    case APEMOST_MODEL_SINE3:
        return n_par == 10;
    case APEMOST_MODEL_PULSE:
        return n_par >= 4 && (n_par - 2) % 2 == 0;
    case APEMOST_MODEL_PULSE_VROT:
        return n_par == 7;
    case APEMOST_MODEL_USER:
        return getenv("APEMOST_DEVICE_MODEL_SRC") != NULL;
    }
    return 0;
}

static const char *model_name(int model) {
    static const char *names[] = {"simplesin", "pulse", "pulse_vrot", "sine3", "user"};
    return (model >= 0 && model < 5) ? names[model] : "?";
}

/* The user's likelihood is host C and cannot run on the GPU; the engine carries device
 * re-implementations of the BASELINE models, and takes any other likelihood as device source
 * (APEMOST_DEVICE_MODEL_SRC=<file>, include/apemost_device_model.h; examples/device_models/ has the
 * reference's apps/simplesin2.c, normal.c and bernoulli_example.c).  Find the device likelihood that
 * reproduces the linked calc_model() -- or the functions registered with set_function -- at a
 * handful of points inside the prior box, or stop: a device model is never taken on trust. */
#define DETECT_POINTS 6
int apemost_detect_model(mcmc *m) {
    static int cached = -1;
    static unsigned int cached_npar = 0;
    const unsigned int n = m->n_par;
    const char *forced = getenv("APEMOST_DEVICE_MODEL");
    double *pts, *beta, host_prob[DETECT_POINTS], host_prior[DETECT_POINTS], dev_prob[DETECT_POINTS],
        dev_prior[DETECT_POINTS];
    gsl_vector *saved;
    double saved_prob, saved_prior;
    const double saved_beta = pt(m) ? pt(m)->beta : 1.0;
    unsigned int j, p;
    int model, found = -1;

    if (cached >= 0 && cached_npar == n)
        return cached;
    pts = (double *)malloc(sizeof(double) * DETECT_POINTS * n);
    beta = (double *)malloc(sizeof(double) * DETECT_POINTS);
    saved = gsl_vector_alloc(n);
    gsl_vector_memcpy(saved, m->params);
    saved_prob = m->prob;
    saved_prior = m->prior;
    for (j = 0; j < DETECT_POINTS; j++) {
        for (p = 0; p < n; p++) {
            const double lo = gsl_vector_get(m->params_min, p), hi = gsl_vector_get(m->params_max, p);
            double frac = (j + 1) * 0.6180339887498949 * (p + 1) + 0.137 * p;
            frac -= floor(frac);
            pts[j * n + p] = lo + (hi - lo) * (0.05 + 0.9 * frac);
            gsl_vector_set(m->params, p, pts[j * n + p]);
        }
        /* every other point at a second temperature: a plugin that tempers differently from the
         * device model (e.g. the library flavour's beta * (prior + loglike), quirk Q6) must not pass
         * because the chain happens to sit at beta = 1 */
        beta[j] = pt(m) ? ((j & 1) ? 0.37 * saved_beta : saved_beta) : 1.0;
        if (pt(m))
            pt(m)->beta = beta[j];
        calc_model(m, NULL);
        host_prob[j] = m->prob;
        host_prior[j] = m->prior;
    }
    if (pt(m))
        pt(m)->beta = saved_beta;
    gsl_vector_memcpy(m->params, saved);
    gsl_vector_free(saved);
    m->prob = saved_prob;
    m->prior = saved_prior;

    for (model = 0; model < 5 && found < 0; model++) {
        apemost_hip_sampler *s;
        int ok = 1;
        if (!model_fits(model, n))
            continue;
        if (getenv("APEMOST_DEVICE_MODEL_SRC") != NULL && model != APEMOST_MODEL_USER)
            continue; /* the user names the device likelihood: that one is checked, no other */
        if (forced && strcmp(forced, model_name(model)) != 0)
            continue;
        s = create_sampler(m, model, 1, 0, 1, default_device(), 0);
        apemost_hip_or_die(apemost_hip_loglike(s, DETECT_POINTS, pts, beta, dev_prob, dev_prior),
                           "apemost_hip_loglike");
        apemost_hip_destroy(s);
        for (j = 0; j < DETECT_POINTS; j++) {
            const double scale = fabs(host_prob[j]) > 1 ? fabs(host_prob[j]) : 1;
            if (!(fabs(dev_prob[j] - host_prob[j]) <= 1e-9 * scale))
                ok = 0;
            if (model == APEMOST_MODEL_PULSE || model == APEMOST_MODEL_PULSE_VROT || model == APEMOST_MODEL_USER)
                if (!(fabs(dev_prior[j] - host_prior[j]) <= 1e-9 * (fabs(host_prior[j]) + 1)))
                    ok = 0;
        }
        if (ok)
            found = model;
    }
    free(pts);
    free(beta);
    if (found < 0) {
        fprintf(stderr,
                "APEMoST MI355X engine: the linked calc_model() (%u parameters) matches %s\n"
                "(SIGMA=%g HMIN=%g).  Device models must reproduce the host plugin to 1e-9; refusing to sample a\n"
                "different posterior.  A likelihood other than the built-in ones is given as device source:\n"
                "APEMOST_DEVICE_MODEL_SRC=<file> (include/apemost_device_model.h).\n",
                n, getenv("APEMOST_DEVICE_MODEL_SRC") ? "not the device model of APEMOST_DEVICE_MODEL_SRC"
                                                      : "none of the device likelihoods (simplesin, pulse, pulse_vrot, sine3)",
                (double)SIGMA, (double)HMIN);
        exit(1);
    }
    IFDEBUG printf("device likelihood: %s\n", model_name(found));
    cached = found;
    cached_npar = n;
    return found;
}

static void alloc_view(apemost_ladder *l) {
    const size_t n = l->n, np = l->n_par;
    apemost_hip_state_view *v = &l->v;
    v->params = (double *)calloc(n * np, sizeof(double));
    v->params_best = (double *)calloc(n * np, sizeof(double));
    v->step = (double *)calloc(n * np, sizeof(double));
    v->pmin = (double *)calloc(n * np, sizeof(double));
    v->pmax = (double *)calloc(n * np, sizeof(double));
    v->params_accepts = (uint64_t *)calloc(n * np, sizeof(uint64_t));
    v->params_rejects = (uint64_t *)calloc(n * np, sizeof(uint64_t));
    v->beta = (double *)calloc(n, sizeof(double));
    v->prob = (double *)calloc(n, sizeof(double));
    v->prior = (double *)calloc(n, sizeof(double));
    v->prob_best = (double *)calloc(n, sizeof(double));
    v->accept = (uint64_t *)calloc(n, sizeof(uint64_t));
    v->reject = (uint64_t *)calloc(n, sizeof(uint64_t));
    v->n_iter = (uint64_t *)calloc(n, sizeof(uint64_t));
    v->swapcount = (uint64_t *)calloc(n, sizeof(uint64_t));
    v->ticks = (uint64_t *)calloc(n, sizeof(uint64_t));
}

static void free_view(apemost_hip_state_view *v) {
    free(v->params);
    free(v->params_best);
    free(v->step);
    free(v->pmin);
    free(v->pmax);
    free(v->params_accepts);
    free(v->params_rejects);
    free(v->beta);
    free(v->prob);
    free(v->prior);
    free(v->prob_best);
    free(v->accept);
    free(v->reject);
    free(v->n_iter);
    free(v->swapcount);
    free(v->ticks);
}

/* the staging arrays of shard k: the same arrays, offset to its first chain */
static apemost_hip_state_view shard_view(const apemost_ladder *l, unsigned int k) {
    const size_t c = l->lo[k], np = l->n_par;
    apemost_hip_state_view v = l->v;
    v.params += c * np;
    v.params_best += c * np;
    v.step += c * np;
    v.pmin += c * np;
    v.pmax += c * np;
    v.params_accepts += c * np;
    v.params_rejects += c * np;
    v.beta += c;
    v.prob += c;
    v.prior += c;
    v.prob_best += c;
    v.accept += c;
    v.reject += c;
    v.n_iter += c;
    v.swapcount += c;
    v.ticks += c;
    return v;
}

void apemost_ladder_upload(apemost_ladder *l) {
    const unsigned int np = l->n_par;
    unsigned int c, p;
    for (c = 0; c < l->n; c++) {
        const mcmc *m = l->chains[c];
        for (p = 0; p < np; p++) {
            const size_t k = (size_t)c * np + p;
            l->v.params[k] = gsl_vector_get(m->params, p);
            l->v.params_best[k] = gsl_vector_get(m->params_best, p);
            l->v.step[k] = gsl_vector_get(m->params_step, p);
            l->v.pmin[k] = gsl_vector_get(m->params_min, p);
            l->v.pmax[k] = gsl_vector_get(m->params_max, p);
            l->v.params_accepts[k] = m->params_accepts[p];
            l->v.params_rejects[k] = m->params_rejects[p];
        }
        l->v.beta[c] = pt(m) ? pt(m)->beta : 1.0;
        l->v.swapcount[c] = pt(m) ? pt(m)->swapcount : 0;
        l->v.ticks[c] = apemost_chain_address(m)->tick;
        l->v.prob[c] = m->prob;
        l->v.prior[c] = m->prior;
        l->v.prob_best[c] = m->prob_best;
        l->v.accept[c] = m->accept;
        l->v.reject[c] = m->reject;
        l->v.n_iter[c] = m->n_iter;
    }
    for (c = 0; c < l->n_shards; c++) {
        const apemost_hip_state_view part = shard_view(l, c);
        apemost_hip_or_die(apemost_hip_set_state(l->s[c], &part), "apemost_hip_set_state");
    }
}

void apemost_ladder_download(apemost_ladder *l) {
    const unsigned int np = l->n_par;
    unsigned int c, p;
    for (c = 0; c < l->n_shards; c++) {
        const apemost_hip_state_view part = shard_view(l, c);
        apemost_hip_or_die(apemost_hip_get_state(l->s[c], &part), "apemost_hip_get_state");
    }
    for (c = 0; c < l->n; c++) {
        mcmc *m = l->chains[c];
        for (p = 0; p < np; p++) {
            const size_t k = (size_t)c * np + p;
            gsl_vector_set(m->params, p, l->v.params[k]);
            gsl_vector_set(m->params_best, p, l->v.params_best[k]);
            gsl_vector_set(m->params_step, p, l->v.step[k]);
            m->params_accepts[p] = (unsigned long)l->v.params_accepts[k];
            m->params_rejects[p] = (unsigned long)l->v.params_rejects[k];
        }
        if (pt(m))
            pt(m)->swapcount = (unsigned long)l->v.swapcount[c];
        apemost_chain_address(m)->tick = (unsigned long)l->v.ticks[c];
        m->prob = l->v.prob[c];
        m->prior = l->v.prior[c];
        m->prob_best = l->v.prob_best[c];
        m->accept = (unsigned long)l->v.accept[c];
        m->reject = (unsigned long)l->v.reject[c];
        m->n_iter = (unsigned long)l->v.n_iter[c];
    }
}

/* APEMOST_DEVICES=0,1,...: device ordinal of every shard, in ladder order */
static unsigned int parse_devices(int *devices) {
    const char *spec = getenv("APEMOST_DEVICES");
    unsigned int n = 0;
    while (spec != NULL && *spec != 0 && n < APEMOST_MAX_SHARDS) {
        devices[n++] = atoi(spec);
        spec = strchr(spec, ',');
        if (spec != NULL)
            spec++;
    }
    if (n == 0)
        devices[n++] = default_device();
    return n;
}

apemost_ladder *apemost_ladder_open(mcmc **chains, unsigned int n_chains) {
    apemost_ladder *l = (apemost_ladder *)calloc(1, sizeof(apemost_ladder));
    const int model = apemost_detect_model(chains[0]);
    int devices[APEMOST_MAX_SHARDS];
    unsigned int k, shards = parse_devices(devices);
    l->chains = chains;
    l->n = n_chains;
    l->n_par = chains[0]->n_par;
    /* every shard at least two chains (shard 0 calibrates chains 0 and 1 by itself) */
    while (shards > 1 && n_chains < 2 * shards)
        shards--;
    l->n_shards = shards;
    for (k = 0; k <= shards; k++)
        l->lo[k] = (unsigned int)(((unsigned long)k * n_chains + shards - 1) / shards);
    for (k = 0; k < shards; k++)
        l->s[k] = create_sampler(chains[0], model, l->lo[k + 1] - l->lo[k], (long)l->lo[k], (long)n_chains,
                                 devices[k], 1);
    alloc_view(l);
    apemost_ladder_upload(l);
    for (k = 0; k < shards; k++)
        apemost_hip_or_die(apemost_hip_set_round(l->s[k], apemost_swap_round, 0), "apemost_hip_set_round");
    return l;
}

void apemost_ladder_close(apemost_ladder *l) {
    unsigned int k;
    if (l == NULL)
        return;
    for (k = 0; k < l->n_shards; k++)
        apemost_hip_destroy(l->s[k]);
    free_view(&l->v);
    free(l);
}

apemost_hip_sampler *apemost_ladder_sampler(apemost_ladder *l) { return l->s[0]; }
unsigned int apemost_ladder_shards(const apemost_ladder *l) { return l->n_shards; }
apemost_hip_sampler *apemost_ladder_shard(apemost_ladder *l, unsigned int k) { return l->s[k]; }
unsigned int apemost_ladder_shard_first(const apemost_ladder *l, unsigned int k) { return l->lo[k]; }

/* calc_model() for chains [first, first+count) of the ladder, wherever they live */
void apemost_ladder_calc_model(apemost_ladder *l, unsigned int first, unsigned int count) {
    unsigned int k;
    for (k = 0; k < l->n_shards; k++) {
        const unsigned int a = first > l->lo[k] ? first : l->lo[k];
        const unsigned int b = first + count < l->lo[k + 1] ? first + count : l->lo[k + 1];
        if (a < b)
            apemost_hip_or_die(apemost_hip_calc_model(l->s[k], (int)(a - l->lo[k]), (int)(b - a)), "calc_model");
    }
}

/* calibration_progress.data as the reference leaves it (src/markov_chain_calibrate.c:1052, 1141-1146):
 * opened "w" by every chain's calibration, one line per parameter and readjustment; here written
 * from the log the device kept for the chain named in apemost_hip_calib_config.progress_chain */
void apemost_write_calibration_progress(apemost_hip_sampler *s, unsigned int n_par) {
    int32_t n_rows = 0, k;
    unsigned int i;
    double *rows;
    FILE *f;
    apemost_hip_or_die(apemost_hip_calibrate_progress(s, NULL, 0, &n_rows), "calibration_progress.data");
    rows = (double *)malloc(((size_t)n_rows + 1) * (1 + 2 * n_par) * sizeof(double));
    if (n_rows > 0)
        apemost_hip_or_die(apemost_hip_calibrate_progress(s, rows, n_rows, &n_rows), "calibration_progress.data");
    f = fopen("calibration_progress.data", "w");
    if (f != NULL) {
        for (k = 0; k < n_rows; k++) {
            const double *r = rows + (size_t)k * (1 + 2 * n_par);
            for (i = 0; i < n_par; i++)
                fprintf(f, "%d\t%lu\t%f\t%f\t%f\n", (int)i, (unsigned long)r[0], r[1 + 2 * i], r[2 + 2 * i], -1.);
        }
        fclose(f);
    }
    free(rows);
}

/* markov_chain_calibrate() (or burn_in only) for chains [first, first+count): every shard's first
 * segment is launched before anything is awaited, then the shards are polled in turn, so the devices
 * calibrate concurrently from the first launch to the last.  status[count] per chain; returns the
 * first non-zero ABI return code (APEMOST_HIP_ERR_CALIBRATION, ...) or 0.  The readjustments of the
 * last chain of the range go to calibration_progress.data: in the reference's (single-threaded)
 * order that chain's calibration is the last one to open the file. */
int apemost_ladder_calibrate(apemost_ladder *l, unsigned int first, unsigned int count,
                             const apemost_hip_calib_config *c, int burn_in_only, int32_t *status) {
    unsigned int k;
    int rc = APEMOST_HIP_OK;
    const unsigned int last = first + count - 1;
    apemost_hip_sampler *logger = NULL;
    for (k = 0; k < l->n_shards; k++) {
        const unsigned int a = first > l->lo[k] ? first : l->lo[k];
        const unsigned int b = first + count < l->lo[k + 1] ? first + count : l->lo[k + 1];
        if (a < b) {
            apemost_hip_calib_config ck = *c;
            ck.progress_chain = -1;
            if (!burn_in_only && last >= a && last < b) {
                ck.progress_chain = (int32_t)(last - l->lo[k]);
                logger = l->s[k];
            }
            apemost_hip_or_die(apemost_hip_calibrate_begin(l->s[k], (int)(a - l->lo[k]), (int)(b - a), &ck, burn_in_only),
                               "markov_chain_calibrate");
        }
    }
    if (l->n_shards > 1) { /* keep every device fed until all are done */
        int32_t active = 1;
        while (active > 0)
            apemost_hip_or_die(apemost_hip_calibrate_wait_any(l->s, (int32_t)l->n_shards, &active), "markov_chain_calibrate");
    }
    for (k = 0; k < l->n_shards; k++) {
        const unsigned int a = first > l->lo[k] ? first : l->lo[k];
        const unsigned int b = first + count < l->lo[k + 1] ? first + count : l->lo[k + 1];
        if (a < b) {
            const int r = apemost_hip_calibrate_end(l->s[k], status ? status + (a - first) : NULL, NULL);
            if (r != APEMOST_HIP_OK && r != APEMOST_HIP_ERR_CALIBRATION)
                apemost_hip_or_die(r, "markov_chain_calibrate");
            if (r != APEMOST_HIP_OK && rc == APEMOST_HIP_OK)
                rc = r;
        }
    }
    if (logger != NULL)
        apemost_write_calibration_progress(logger, l->n_par);
    return rc;
}

/* n_rounds x {n_swap steps per chain, one swap attempt} on the whole ladder (asynchronous);
 * d_samples[k]: device rows of shard k or NULL.  n_rounds = 0 with a pending swap attempt applies
 * just that attempt. */
void apemost_ladder_run(apemost_ladder *l, unsigned long n_rounds, unsigned int n_swap, double **d_samples) {
    apemost_hip_or_die(apemost_hip_run_shards(l->s, (int)l->n_shards, n_rounds, n_swap, d_samples), "run_sampler");
}

/* one-chain twin for the single-chain API; rebuilt when the data matrix or n_par changes */
#define SINGLE_LADDER_SPAN 1048576L
apemost_ladder *apemost_single(mcmc *m) {
    static apemost_ladder *cache = NULL;
    static const gsl_matrix *cache_data = NULL;
    static mcmc *slot[1];
    if (cache == NULL || cache_data != m->data || cache->n_par != m->n_par) {
        const int model = apemost_detect_model(m);
        apemost_ladder_close(cache);
        cache = (apemost_ladder *)calloc(1, sizeof(apemost_ladder));
        cache->chains = slot;
        cache->n = 1;
        cache->n_par = m->n_par;
        cache->n_shards = 1;
        cache->lo[0] = 0;
        cache->lo[1] = 1;
        cache->s[0] = create_sampler(m, model, 1, 0, SINGLE_LADDER_SPAN, default_device(), 0);
        alloc_view(cache);
        cache_data = m->data;
    }
    slot[0] = m;
    apemost_hip_or_die(apemost_hip_set_chain_offset(cache->s[0], (long)(apemost_chain_address(m)->chain_id % SINGLE_LADDER_SPAN)),
                       "apemost_hip_set_chain_offset");
    return cache;
}
