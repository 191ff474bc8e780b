/* Device likelihood of the reference's apps/bernoulli_example.c: logistic regression.  Data column 0
 * is the 0/1 outcome, columns 1.. are the regressors; parameter 0 is the intercept
 * (apps/bernoulli_example.c:10-49, SIGMA = 2 is fixed in that file).
 *   prior    = sum_{j >= 1} -(param_j / SIGMA)^2 / 2                        (:21-24)
 *   term(i)  = log(p_i) or log(1 - p_i), p_i = logistic(eta_i),  eta_i = param_0 + sum_j x_ij param_j   (:29-44)
 *   finish() = prior + get_beta(m) * sum                                    (:47) */
#include "apemost_device_model.h"

__device__ double apemost_user_term(const apemost_model_ctx *ctx, int i) {
    double eta = ctx->params[0], p;
    for (int j = 1; j < ctx->n_par; j++)
        eta += APEMOST_DATA(ctx, i, j) * ctx->params[j];
    if (eta > 0)
        p = 1 / (1 + exp(-eta));
    else
        p = exp(eta) / (1 + exp(eta));
    return APEMOST_DATA(ctx, i, 0) == 0 ? log(1 - p) : log(p);
}

__device__ double apemost_user_finish(const apemost_model_ctx *ctx, double sum, double beta, double *prior) {
    double pr = 0;
    for (int j = 1; j < ctx->n_cols; j++) {
        const double t = ctx->params[j] / 2;
        pr += -(t * t) / 2;
    }
    *prior = pr;
    return pr + beta * sum;
}
