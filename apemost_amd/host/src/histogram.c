/* Histogram helpers (contract of reference src/histogram.c:34-157; quirks kept where the
 * reference's own unit test pins them, tests/tests.c:57-76). */
#include <stdio.h>
#include <stdlib.h>
#include <gsl/gsl_histogram.h>
#include <gsl/gsl_vector.h>
#include "gsl_helper.h"
#include "utils.h"
#include "debug.h"
#include "histogram.h"

gsl_histogram *create_hist(int nbins, double min, double max) {
    gsl_histogram *h = gsl_histogram_alloc((size_t)nbins);
    gsl_histogram_set_ranges_uniform(h, min, max);
    h->range[h->n] += (max - min) / 10000; /* max itself belongs to the last bin */
    return h;
}

/* As in the reference, the bin count is the LENGTH of v (the nbins argument only enters a value
 * that is never used), the top edge is widened by one whole unit, and the counts are divided by
 * the sum of the VALUES of v: tests/tests.c expects bounds [0, 4) and 2/3, 0, 1/3 for (3, 0, 0). */
gsl_histogram *calc_hist(const gsl_vector *v, int nbins) {
    gsl_histogram *h = gsl_histogram_alloc(v->size);
    double lo, hi, total = 0;
    size_t i;
    (void)nbins;
    gsl_vector_minmax(v, &lo, &hi);
    require(gsl_histogram_set_ranges_uniform(h, lo, hi));
    h->range[h->n] += 1;
    for (i = 0; i < v->size; i++) {
        const double x = gsl_vector_get(v, i);
        total += x;
        require(gsl_histogram_increment(h, x));
    }
    require(gsl_histogram_scale(h, 1 / total));
    return h;
}

/* walks a file of n columns and hands every value to `visit(column, value, ctx)`; a short last
 * line ends the walk, anything that is not a number stops the program like the reference does */
static unsigned long walk_columns(const char *filename, unsigned int n,
                                  void (*visit)(unsigned int, double, void *), void *ctx) {
    FILE *f = openfile(filename);
    unsigned long line = 0;
    int more = 1;
    while (more) {
        unsigned int i;
        for (i = 0; i < n; i++) {
            double x;
            if (fscanf(f, "%lf", &x) != 1) {
                if (!feof(f)) {
                    fprintf(stderr, "field could not be read: %d, line %lu in %s\n", i + 1, line + 1, filename);
                    exit(1);
                }
                more = 0;
                break;
            }
            visit(i, x, ctx);
        }
        if (more)
            line++;
    }
    fclose(f);
    return line;
}

static void visit_increment(unsigned int column, double x, void *ctx) {
    gsl_histogram_increment(((gsl_histogram **)ctx)[column], x);
}

void append_to_hists(gsl_histogram **hists, unsigned int n, const char *filename) {
    walk_columns(filename, n, visit_increment, hists);
}

struct extremes {
    gsl_vector *min, *max;
    unsigned long seen; /* values visited so far: the first row initialises */
};

static void visit_extremes(unsigned int column, double x, void *ctx) {
    struct extremes *e = (struct extremes *)ctx;
    if (e->seen < e->min->size) {
        gsl_vector_set(e->min, column, x);
        gsl_vector_set(e->max, column, x);
    } else {
        if (x < gsl_vector_get(e->min, column))
            gsl_vector_set(e->min, column, x);
        if (x > gsl_vector_get(e->max, column))
            gsl_vector_set(e->max, column, x);
    }
    e->seen++;
}

void find_min_max(char *filename, gsl_vector *min, gsl_vector *max) {
    struct extremes e;
    assert(min->size == max->size);
    e.min = min;
    e.max = max;
    e.seen = 0;
    walk_columns(filename, (unsigned int)min->size, visit_extremes, &e);
    if (e.seen < min->size) {
        fprintf(stderr, "field could not be read: %lu, line 1 in %s\n", e.seen + 1, filename);
        exit(1);
    }
}

/* NB reference behaviour (src/histogram.c:147-157): the file's extremes NARROW the given box
 * from inside -- min becomes the larger of the two minima, max the smaller of the two maxima */
void update_min_max(char *filename, gsl_vector *min, gsl_vector *max) {
    gsl_vector *file_min = dup_vector(min), *file_max = dup_vector(max);
    find_min_max(filename, file_min, file_max);
    max_vector(min, file_min);
    min_vector(max, file_max);
    gsl_vector_free(file_min);
    gsl_vector_free(file_max);
}
