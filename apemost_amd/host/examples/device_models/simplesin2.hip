/* Device likelihood of the reference's apps/simplesin2.c (two parameters: amplitude, frequency; the
 * phase is fixed at 0.3312), for APEMOST_DEVICE_MODEL_SRC / apemost_hip_config.device_model_source.
 * Restates apps/simplesin2.c:12-34 in the form include/apemost_device_model.h asks for:
 *   term(i)  = (param0 * sin(2 pi (param1 * x_i + 0.3312)) - y_i)^2        (:12-16, :27-30)
 *   finish() = get_beta(m) * square_sum / (-2 * SIGMA * SIGMA)             (:31)
 * The reference calls gsl_sf_sin; this uses the device math library's sin (<= 1 ulp). */
#include "apemost_device_model.h"

__device__ double apemost_user_term(const apemost_model_ctx *ctx, int i) {
    const double x = APEMOST_DATA(ctx, i, 0);
    const double y = ctx->params[0] * sin(2.0 * 3.14159265358979323846 * (ctx->params[1] * x + 0.3312)) - APEMOST_DATA(ctx, i, 1);
    return y * y;
}

__device__ double apemost_user_finish(const apemost_model_ctx *ctx, double sum, double beta, double *prior) {
    (void)prior; /* set_prior is never called: m->prior stays what it was */
    return beta * sum / (-2 * ctx->sigma * ctx->sigma);
}
