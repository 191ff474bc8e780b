/* Implementation of the minimal GSL-compatible surface in gsl/ (own code).
 * mt19937: Matsumoto & Nishimura's generator with the 2002 seeding, GSL's conventions
 * (seed 0 -> 4357, uniform = get/2^32).  gsl_ran_gaussian: polar Box-Muller, second
 * variate discarded. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <gsl/gsl_matrix.h>
#include <gsl/gsl_histogram.h>
#include <math.h>
#include <gsl/gsl_randist.h>
#include <gsl/gsl_sf.h>

const char *gsl_strerror(const int gsl_errno) {
    switch (gsl_errno) {
    case GSL_SUCCESS:
        return "success";
    case GSL_EDOM:
        return "input domain error";
    case GSL_EINVAL:
        return "invalid argument supplied by user";
    default:
        return "error";
    }
}

void gsl_error(const char *reason, const char *file, int line, int gsl_errno) {
    fflush(stdout);
    fprintf(stderr, "gsl: %s:%d: ERROR: %s\n", file, line, reason);
    fprintf(stderr, "Default GSL error handler invoked (%s).\n", gsl_strerror(gsl_errno));
    fflush(stderr);
    abort();
}

/* ---- vectors ---- */
static gsl_vector *vector_new(const size_t n, int zero) {
    gsl_vector *v;
    if (n == 0)
        gsl_error("vector length n must be positive integer", __FILE__, __LINE__, GSL_EINVAL);
    v = (gsl_vector *)malloc(sizeof(gsl_vector));
    v->block = (gsl_block *)malloc(sizeof(gsl_block));
    v->block->size = n;
    v->block->data = (double *)(zero ? calloc(n, sizeof(double)) : malloc(n * sizeof(double)));
    v->size = n;
    v->stride = 1;
    v->data = v->block->data;
    v->owner = 1;
    return v;
}
gsl_vector *gsl_vector_alloc(const size_t n) { return vector_new(n, 0); }
gsl_vector *gsl_vector_calloc(const size_t n) { return vector_new(n, 1); }
void gsl_vector_free(gsl_vector *v) {
    if (!v)
        return;
    if (v->owner && v->block) {
        free(v->block->data);
        free(v->block);
    }
    free(v);
}
double gsl_vector_get(const gsl_vector *v, const size_t i) {
    if (i >= v->size)
        gsl_error("index out of range", __FILE__, __LINE__, GSL_EINVAL);
    return v->data[i * v->stride];
}
void gsl_vector_set(gsl_vector *v, const size_t i, double x) {
    if (i >= v->size)
        gsl_error("index out of range", __FILE__, __LINE__, GSL_EINVAL);
    v->data[i * v->stride] = x;
}
void gsl_vector_set_all(gsl_vector *v, double x) {
    size_t i;
    for (i = 0; i < v->size; i++)
        v->data[i * v->stride] = x;
}
void gsl_vector_set_zero(gsl_vector *v) { gsl_vector_set_all(v, 0.0); }
static void same_length(const gsl_vector *a, const gsl_vector *b) {
    if (a->size != b->size)
        gsl_error("vectors must have same length", __FILE__, __LINE__, GSL_EINVAL);
}
int gsl_vector_memcpy(gsl_vector *dest, const gsl_vector *src) {
    size_t i;
    same_length(dest, src);
    for (i = 0; i < src->size; i++)
        dest->data[i * dest->stride] = src->data[i * src->stride];
    return GSL_SUCCESS;
}
int gsl_vector_scale(gsl_vector *a, const double x) {
    size_t i;
    for (i = 0; i < a->size; i++)
        a->data[i * a->stride] *= x;
    return GSL_SUCCESS;
}
int gsl_vector_add_constant(gsl_vector *a, const double x) {
    size_t i;
    for (i = 0; i < a->size; i++)
        a->data[i * a->stride] += x;
    return GSL_SUCCESS;
}
#define ELEMENTWISE(NAME, OP)                                                                    \
    int NAME(gsl_vector *a, const gsl_vector *b) {                                               \
        size_t i;                                                                                \
        same_length(a, b);                                                                       \
        for (i = 0; i < a->size; i++)                                                            \
            a->data[i * a->stride] OP b->data[i * b->stride];                                    \
        return GSL_SUCCESS;                                                                      \
    }
ELEMENTWISE(gsl_vector_add, +=)
ELEMENTWISE(gsl_vector_sub, -=)
ELEMENTWISE(gsl_vector_mul, *=)
ELEMENTWISE(gsl_vector_div, /=)
void gsl_vector_minmax(const gsl_vector *v, double *min_out, double *max_out) {
    double lo = v->data[0], hi = v->data[0];
    size_t i;
    for (i = 1; i < v->size; i++) {
        double x = v->data[i * v->stride];
        if (x < lo)
            lo = x;
        if (x > hi)
            hi = x;
    }
    *min_out = lo;
    *max_out = hi;
}
double gsl_vector_max(const gsl_vector *v) {
    double lo, hi;
    gsl_vector_minmax(v, &lo, &hi);
    return hi;
}
double gsl_vector_min(const gsl_vector *v) {
    double lo, hi;
    gsl_vector_minmax(v, &lo, &hi);
    return lo;
}
int gsl_vector_fprintf(FILE *stream, const gsl_vector *v, const char *format) {
    size_t i;
    for (i = 0; i < v->size; i++) {
        if (fprintf(stream, format, v->data[i * v->stride]) < 0 || putc('\n', stream) == EOF)
            return GSL_FAILURE;
    }
    return GSL_SUCCESS;
}

/* ---- matrices ---- */
gsl_matrix *gsl_matrix_alloc(const size_t n1, const size_t n2) {
    gsl_matrix *m;
    if (n1 == 0 || n2 == 0)
        gsl_error("matrix dimensions must be positive integers", __FILE__, __LINE__, GSL_EINVAL);
    m = (gsl_matrix *)malloc(sizeof(gsl_matrix));
    m->block = (gsl_block *)malloc(sizeof(gsl_block));
    m->block->size = n1 * n2;
    m->block->data = (double *)malloc(n1 * n2 * sizeof(double));
    m->size1 = n1;
    m->size2 = n2;
    m->tda = n2;
    m->data = m->block->data;
    m->owner = 1;
    return m;
}
void gsl_matrix_free(gsl_matrix *m) {
    if (!m)
        return;
    if (m->owner && m->block) {
        free(m->block->data);
        free(m->block);
    }
    free(m);
}
double gsl_matrix_get(const gsl_matrix *m, const size_t i, const size_t j) {
    if (i >= m->size1 || j >= m->size2)
        gsl_error("index out of range", __FILE__, __LINE__, GSL_EINVAL);
    return m->data[i * m->tda + j];
}
void gsl_matrix_set(gsl_matrix *m, const size_t i, const size_t j, const double x) {
    if (i >= m->size1 || j >= m->size2)
        gsl_error("index out of range", __FILE__, __LINE__, GSL_EINVAL);
    m->data[i * m->tda + j] = x;
}
void gsl_matrix_set_all(gsl_matrix *m, double x) {
    size_t i, j;
    for (i = 0; i < m->size1; i++)
        for (j = 0; j < m->size2; j++)
            m->data[i * m->tda + j] = x;
}
int gsl_matrix_fscanf(FILE *stream, gsl_matrix *m) {
    size_t i, j;
    for (i = 0; i < m->size1; i++)
        for (j = 0; j < m->size2; j++)
            if (fscanf(stream, "%lf", &m->data[i * m->tda + j]) != 1)
                return GSL_FAILURE;
    return GSL_SUCCESS;
}
int gsl_matrix_get_col(gsl_vector *v, const gsl_matrix *m, const size_t j) {
    size_t i;
    if (j >= m->size2 || v->size != m->size1)
        gsl_error("column index or vector length mismatch", __FILE__, __LINE__, GSL_EINVAL);
    for (i = 0; i < m->size1; i++)
        v->data[i * v->stride] = m->data[i * m->tda + j];
    return GSL_SUCCESS;
}
gsl_vector_const_view gsl_matrix_const_column(const gsl_matrix *m, const size_t j) {
    gsl_vector_const_view view;
    if (j >= m->size2)
        gsl_error("column index is out of range", __FILE__, __LINE__, GSL_EINVAL);
    view.vector.size = m->size1;
    view.vector.stride = m->tda;
    view.vector.data = m->data + j;
    view.vector.block = m->block;
    view.vector.owner = 0;
    return view;
}

/* ---- mt19937 ---- */
#define MT_N 624
#define MT_M 397
typedef struct {
    unsigned long mt[MT_N];
    int mti;
} mt_state;
static const gsl_rng_type mt19937_type = {"mt19937", 0xffffffffUL, 0};
const gsl_rng_type *gsl_rng_mt19937 = &mt19937_type;
const gsl_rng_type *gsl_rng_default = &mt19937_type;
unsigned long gsl_rng_default_seed = 0;

const gsl_rng_type *gsl_rng_env_setup(void) {
    const char *t = getenv("GSL_RNG_TYPE"), *s = getenv("GSL_RNG_SEED");
    if (t && strcmp(t, "mt19937") != 0) {
        fprintf(stderr, "GSL_RNG_TYPE=%s not available in the compat layer (only mt19937)\n", t);
        exit(1);
    }
    gsl_rng_default_seed = s ? strtoul(s, NULL, 0) : 0;
    if (s)
        fprintf(stderr, "GSL_RNG_SEED=%lu\n", gsl_rng_default_seed);
    gsl_rng_default = gsl_rng_mt19937;
    return gsl_rng_default;
}
void gsl_rng_set(const gsl_rng *r, unsigned long seed) {
    mt_state *s = (mt_state *)r->state;
    int i;
    if (seed == 0)
        seed = 4357;
    s->mt[0] = seed & 0xffffffffUL;
    for (i = 1; i < MT_N; i++)
        s->mt[i] = (1812433253UL * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (unsigned long)i) & 0xffffffffUL;
    s->mti = MT_N;
}
gsl_rng *gsl_rng_alloc(const gsl_rng_type *T) {
    gsl_rng *r = (gsl_rng *)malloc(sizeof(gsl_rng));
    r->type = T;
    r->state = malloc(sizeof(mt_state));
    gsl_rng_set(r, gsl_rng_default_seed);
    return r;
}
void gsl_rng_free(gsl_rng *r) {
    if (!r)
        return;
    free(r->state);
    free(r);
}
unsigned long gsl_rng_get(const gsl_rng *r) {
    mt_state *s = (mt_state *)r->state;
    unsigned long k;
    if (s->mti >= MT_N) {
        int kk;
        unsigned long y;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (s->mt[kk] & 0x80000000UL) | (s->mt[kk + 1] & 0x7fffffffUL);
            s->mt[kk] = s->mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
        }
        for (; kk < MT_N - 1; kk++) {
            y = (s->mt[kk] & 0x80000000UL) | (s->mt[kk + 1] & 0x7fffffffUL);
            s->mt[kk] = s->mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
        }
        y = (s->mt[MT_N - 1] & 0x80000000UL) | (s->mt[0] & 0x7fffffffUL);
        s->mt[MT_N - 1] = s->mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1UL) ? 0x9908b0dfUL : 0UL);
        s->mti = 0;
    }
    k = s->mt[s->mti++];
    k ^= (k >> 11);
    k ^= (k << 7) & 0x9d2c5680UL;
    k ^= (k << 15) & 0xefc60000UL;
    k ^= (k >> 18);
    return k & 0xffffffffUL;
}
double gsl_rng_uniform(const gsl_rng *r) { return gsl_rng_get(r) / 4294967296.0; }
double gsl_rng_uniform_pos(const gsl_rng *r) {
    double x;
    do {
        x = gsl_rng_uniform(r);
    } while (x == 0);
    return x;
}

/* ---- distributions ---- */
double gsl_ran_gaussian(const gsl_rng *r, const double sigma) {
    double x, y, r2;
    do {
        x = -1 + 2 * gsl_rng_uniform_pos(r);
        y = -1 + 2 * gsl_rng_uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0);
    return sigma * y * sqrt(-2.0 * log(r2) / r2);
}
double gsl_ran_logistic(const gsl_rng *r, const double a) {
    double x;
    do {
        x = gsl_rng_uniform_pos(r);
    } while (x == 1);
    return a * log(x / (1 - x));
}
double gsl_ran_flat(const gsl_rng *r, const double a, const double b) {
    double u = gsl_rng_uniform(r);
    return a * (1 - u) + b * u;
}

/* ---- special functions ---- */
double gsl_sf_log(const double x) {
    if (x <= 0.0)
        gsl_error("domain error", __FILE__, __LINE__, GSL_EDOM);
    return log(x);
}
double gsl_sf_sin(const double x) { return sin(x); }
double gsl_sf_cos(const double x) { return cos(x); }
double gsl_sf_exp(const double x) { return exp(x); }

/* ---- histograms: bin i covers [range[i], range[i+1]) ---- */
gsl_histogram *gsl_histogram_alloc(size_t n) {
    gsl_histogram *h;
    if (n == 0)
        GSL_ERROR_NULL("histogram length n must be positive integer", GSL_EDOM);
    h = (gsl_histogram *)malloc(sizeof(gsl_histogram));
    h->range = h ? (double *)calloc(n + 1, sizeof(double)) : NULL;
    h->bin = h ? (double *)calloc(n, sizeof(double)) : NULL;
    if (h == NULL || h->range == NULL || h->bin == NULL)
        GSL_ERROR_NULL("failed to allocate space for histogram", GSL_ENOMEM);
    h->n = n;
    return h;
}

void gsl_histogram_free(gsl_histogram *h) {
    if (h == NULL)
        return;
    free(h->range);
    free(h->bin);
    free(h);
}

int gsl_histogram_set_ranges_uniform(gsl_histogram *h, double xmin, double xmax) {
    const size_t n = h->n;
    size_t i;
    if (xmin >= xmax)
        GSL_ERROR("xmin must be less than xmax", GSL_EINVAL);
    for (i = 0; i <= n; i++) {
        const double f1 = (double)(n - i) / (double)n, f2 = (double)i / (double)n;
        h->range[i] = f1 * xmin + f2 * xmax;
    }
    for (i = 0; i < n; i++)
        h->bin[i] = 0;
    return GSL_SUCCESS;
}

int gsl_histogram_increment(gsl_histogram *h, double x) {
    size_t lo = 0, hi = h->n;
    if (!(x >= h->range[0]) || !(x < h->range[h->n]))
        return GSL_EDOM; /* outside: silently ignored, as GSL does */
    while (hi - lo > 1) { /* range[lo] <= x < range[hi] */
        const size_t mid = (lo + hi) / 2;
        if (x >= h->range[mid])
            lo = mid;
        else
            hi = mid;
    }
    h->bin[lo] += 1;
    return GSL_SUCCESS;
}

double gsl_histogram_get(const gsl_histogram *h, size_t i) {
    if (i >= h->n)
        GSL_ERROR_VAL("index lies outside valid range of 0 .. n - 1", GSL_EDOM, 0);
    return h->bin[i];
}

int gsl_histogram_get_range(const gsl_histogram *h, size_t i, double *lower, double *upper) {
    if (i >= h->n)
        GSL_ERROR("index lies outside valid range of 0 .. n - 1", GSL_EDOM);
    *lower = h->range[i];
    *upper = h->range[i + 1];
    return GSL_SUCCESS;
}

double gsl_histogram_max(const gsl_histogram *h) { return h->range[h->n]; }
double gsl_histogram_min(const gsl_histogram *h) { return h->range[0]; }
size_t gsl_histogram_bins(const gsl_histogram *h) { return h->n; }

double gsl_histogram_sum(const gsl_histogram *h) {
    double s = 0;
    size_t i;
    for (i = 0; i < h->n; i++)
        s += h->bin[i];
    return s;
}

/* weighted mean / spread of the bin centres (negative bins count as empty) */
double gsl_histogram_mean(const gsl_histogram *h) {
    double mean = 0, w = 0;
    size_t i;
    for (i = 0; i < h->n; i++) {
        const double x = (h->range[i + 1] + h->range[i]) / 2, b = h->bin[i];
        if (b > 0) {
            w += b;
            mean += (x - mean) * (b / w);
        }
    }
    return mean;
}

double gsl_histogram_sigma(const gsl_histogram *h) {
    const double mean = gsl_histogram_mean(h);
    double var = 0, w = 0;
    size_t i;
    for (i = 0; i < h->n; i++) {
        const double d = (h->range[i + 1] + h->range[i]) / 2 - mean, b = h->bin[i];
        if (b > 0) {
            w += b;
            var += (d * d - var) * (b / w);
        }
    }
    return sqrt(var);
}

int gsl_histogram_scale(gsl_histogram *h, double scale) {
    size_t i;
    for (i = 0; i < h->n; i++)
        h->bin[i] *= scale;
    return GSL_SUCCESS;
}

int gsl_histogram_fprintf(FILE *stream, const gsl_histogram *h, const char *range_format, const char *bin_format) {
    size_t i;
    for (i = 0; i < h->n; i++) {
        if (fprintf(stream, range_format, h->range[i]) < 0 || putc(' ', stream) == EOF ||
            fprintf(stream, range_format, h->range[i + 1]) < 0 || putc(' ', stream) == EOF ||
            fprintf(stream, bin_format, h->bin[i]) < 0 || putc('\n', stream) == EOF)
            GSL_ERROR("fprintf failed", GSL_EFAILED);
    }
    return GSL_SUCCESS;
}
