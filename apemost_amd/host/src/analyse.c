/* `analyse` phase: post-processing of the run phase's dump files on the host (disk-bound work,
 * outside the GPU path).  Same inputs, outputs and formulas as reference
 * src/analyse.c:33-285 and its histogram helper (src/histogram.c:33-42): the thermodynamic
 * integral over beta of the mean log-likelihood, and per-parameter marginal histograms with a
 * batch-means Monte-Carlo error.  The reference's n_beta < 100 limit does not apply. */
#include <math.h>
#include <string.h>
#include "mcmc.h"
#include "parallel_tempering.h"
#include "parallel_tempering_config.h"
#include "utils.h"
#include "debug.h"

#ifndef NBINS
#define NBINS 200
#endif
#ifndef GNUPLOT_STYLE
#define GNUPLOT_STYLE "with histeps"
#endif

/* ln p(D|M) = integral over beta of <ln L>_beta: per chain the mean of column 2 of
 * prob-chain<i>.dump (prob - prior = beta * ln L) divided by beta, rectangle rule from the
 * hottest chain down to beta = 1 */
void analyse_data_probability() {
    const unsigned int n_beta = apemost_n_beta();
    mcmc **chains = setup_chains();
    double *mean = (double *)calloc(n_beta, sizeof(double));
    double logprob = 0, previous_beta = 0;
    unsigned int i, j;
    read_calibration_file(chains, n_beta);
    for (i = 0; i < n_beta; i++) {
        char name[100];
        FILE *f;
        double total, part, sum = 0;
        unsigned long n = 0;
        sprintf(name, "prob-chain%d.dump", i);
        printf("reading probabilities of chain %d\r", i);
        fflush(stdout);
        f = fopen(name, "r");
        if (f == NULL) {
            fprintf(stderr, "calculating data probability failed: file %s not found\n", name);
            return;
        }
        while (fscanf(f, "%le\t%le", &total, &part) == 2) {
            sum += part;
            n++;
        }
        fclose(f);
        if (n == 0) {
            fprintf(stderr, "calculating data probability failed: no data points found in %s\n", name);
            return;
        }
        mean[i] = sum / get_beta(chains[i]) / n;
    }
    for (j = n_beta; j-- > 0;) {
        assert(get_beta(chains[j]) > previous_beta);
        logprob += mean[j] * (get_beta(chains[j]) - previous_beta);
        previous_beta = get_beta(chains[j]);
    }
    printf("Model probability ln(p(D|M, I)): [about 10^%.0f] %.5f\n"
           "\nTable to compare support against other models (Jeffrey):\n"
           " other model ln(p(D|M,I)) | supporting evidence for this model\n"
           " --------------------------------- \n"
           "        >  %04.1f \tnegative (supports other model)\n"
           "  %04.1f .. %04.1f \tBarely worth mentioning\n"
           "  %04.1f .. %04.1f \tSubstantial\n"
           "  %04.1f .. %04.1f \tStrong\n"
           "  %04.1f .. %04.1f \tVery strong\n"
           "        <  %04.1f \tDecisive\n",
           logprob / log(10.0), logprob, logprob, logprob, logprob - log(3.0), logprob - log(3.0),
           logprob - log(10.0), logprob - log(10.0), logprob - log(30.0), logprob - log(30.0),
           logprob - log(100.0), logprob - log(100.0));
    printf("\nbe careful.\n");
    free(mean);
}

/* spread of the batch means (batches of `batchsize` consecutive samples) around the mean */
static double batch_means_error(double mean, const char *filename, unsigned long batchsize) {
    FILE *f = openfile(filename);
    double v, batchsum = 0, errorsum = 0;
    unsigned long n = 0;
    int nbatches = 0;
    while (fscanf(f, "%lf", &v) == 1) {
        n++;
        batchsum += v;
        if (n % batchsize == batchsize - 1) {
            const double d = batchsum / batchsize - mean;
            errorsum += d * d;
            batchsum = 0;
            nbatches++;
        }
    }
    fclose(f);
    return sqrt(errorsum / nbatches);
}

/* NBINS-bin density of one parameter's visited values (chain 0) over [min, max] of the prior box
 * (or of the data with -DHISTOGRAMS_MINMAX); the top edge is widened by 1e-4 of the range so the
 * maximum falls into the last bin.  Output: "<name>.histogram", lines "lower upper density". */
static void marginal_distribution(mcmc **chains, unsigned int param, int find_minmax) {
    const char *name = get_params_descr(chains[0])[param];
    double lo = get_params_min_for(chains[0], param), hi = get_params_max_for(chains[0], param);
    double bins[NBINS], edges[NBINS + 1], v, total = 0, mean = 0, var = 0, width, err;
    char in_name[300], out_name[300];
    FILE *f;
    int b;
    sprintf(in_name, "%s-chain-%d.prob.dump", name, 0);
    sprintf(out_name, "%s.histogram", name);
    if (get_column_count(in_name) != 1) {
        fprintf(stderr, "number of columns different in file %s\n", in_name);
        exit(1);
    }
    if (find_minmax) {
        int first = 1;
        f = openfile(in_name);
        while (fscanf(f, "%lf", &v) == 1) {
            if (first || v < lo)
                lo = v;
            if (first || v > hi)
                hi = v;
            first = 0;
        }
        fclose(f);
    }
    for (b = 0; b <= NBINS; b++)
        edges[b] = lo + (hi - lo) * b / NBINS;
    edges[NBINS] += (hi - lo) / 10000;
    memset(bins, 0, sizeof bins);
    printf("reading values: chain %3d parameter %s   \r", 0, name);
    fflush(stdout);
    f = openfile(in_name);
    while (fscanf(f, "%lf", &v) == 1) {
        if (v < edges[0] || v >= edges[NBINS])
            continue;
        b = (int)((v - lo) / (hi - lo) * NBINS);
        if (b >= NBINS)
            b = NBINS - 1;
        while (b > 0 && v < edges[b])
            b--;
        while (b < NBINS - 1 && v >= edges[b + 1])
            b++;
        bins[b] += 1;
        total += 1;
    }
    fclose(f);
    width = (hi - lo) / NBINS;
    f = fopen(out_name, "w");
    assert(f != NULL);
    for (b = 0; b < NBINS; b++) {
        bins[b] *= width / total; /* the reference's scaling: (max-min)/nbins/iterations */
        fprintf(f, DUMP_FORMAT " " DUMP_FORMAT " " DUMP_FORMAT "\n", edges[b], edges[b + 1], bins[b]);
    }
    fclose(f);
    {
        double wsum = 0;
        for (b = 0; b < NBINS; b++) { /* histogram mean and sigma from the bin centres */
            const double centre = 0.5 * (edges[b] + edges[b + 1]);
            wsum += bins[b];
            mean += bins[b] * centre;
        }
        mean /= wsum;
        for (b = 0; b < NBINS; b++) {
            const double d = 0.5 * (edges[b] + edges[b + 1]) - mean;
            var += bins[b] * d * d;
        }
        var /= wsum;
    }
    err = batch_means_error(mean, in_name, (unsigned long)sqrt(total));
    printf("mcmc error estimate of %s: %f %s\n", name, err, (err > sqrt(var) * 0.01 ? "** high!" : " (ok)"));
    printf("Note: Include a error estimate in your publication!\n");
}

void analyse_marginal_distributions() {
    const unsigned int n_beta = apemost_n_beta();
    mcmc **chains = setup_chains();
    const unsigned int n_par = get_n_par(chains[0]);
    int find_minmax = 0;
    unsigned int i;
    FILE *plot;
    read_calibration_file(chains, n_beta);
#ifdef HISTOGRAMS_MINMAX
    find_minmax = 1;
#endif
    for (i = 0; i < n_par; i++)
        marginal_distribution(chains, i, find_minmax);
    plot = fopen("marginal_distributions.gnuplot", "w");
    assert(plot != NULL);
    fprintf(plot, "# set terminal png size %d,%d; set output \"marginal_distributions.png\"\n", 600, 300 * n_par);
    fprintf(plot, "set multiplot\n");
    fprintf(plot, "set size 1,%f\n", 1. / n_par);
    for (i = 0; i < n_par; i++) {
        fprintf(plot, "set origin 0,%f\n", (n_par - i - 1) * 1. / n_par);
        fprintf(plot, "plot \"%s.histogram\" u 1:3 title \"%s\" " GNUPLOT_STYLE "\n", get_params_descr(chains[0])[i],
                get_params_descr(chains[0])[i]);
    }
    fprintf(plot, "unset multiplot\n");
    fclose(plot);
}
