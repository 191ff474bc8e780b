#ifndef APEMOST_COMPAT_GSL_RNG_H
#define APEMOST_COMPAT_GSL_RNG_H
#include <gsl/gsl_math.h>

typedef struct {
    const char *name;
    unsigned long max;
    unsigned long min;
} gsl_rng_type;

typedef struct {
    const gsl_rng_type *type;
    void *state;
} gsl_rng;

extern const gsl_rng_type *gsl_rng_mt19937;
extern const gsl_rng_type *gsl_rng_default;
extern unsigned long gsl_rng_default_seed;

/* reads GSL_RNG_TYPE / GSL_RNG_SEED like GSL; only mt19937 is provided */
const gsl_rng_type *gsl_rng_env_setup(void);
gsl_rng *gsl_rng_alloc(const gsl_rng_type *T);
void gsl_rng_free(gsl_rng *r);
void gsl_rng_set(const gsl_rng *r, unsigned long seed);
unsigned long gsl_rng_get(const gsl_rng *r);
double gsl_rng_uniform(const gsl_rng *r);
double gsl_rng_uniform_pos(const gsl_rng *r);
#endif
