/* Minimal GSL-compatible surface for builds on machines without GSL (the MI355X image has
 * none).  Own code, C90-clean (compiles under the reference's `-ansi -pedantic`).  Only what
 * APEMoST's engine API and its example applications touch; field layout of gsl_vector /
 * gsl_matrix follows GSL's public structs because applications read ->size, ->size1, ->data.
 * When a real GSL is installed, build with USE_SYSTEM_GSL=1 and this directory is not used. */
#ifndef APEMOST_COMPAT_GSL_MATH_H
#define APEMOST_COMPAT_GSL_MATH_H
#include <limits.h>
#include <math.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846264338328
#endif
#ifndef M_E
#define M_E 2.71828182845904523536028747135
#endif
#define GSL_SUCCESS 0
#define GSL_FAILURE (-1)
#define GSL_EDOM 1
#define GSL_EINVAL 4
#define GSL_EFAILED 5
#define GSL_ENOMEM 8
/* report through gsl_error (prints and aborts by default), then return like GSL's macros */
#define GSL_ERROR_VAL(reason, gsl_errno, value)                                                   \
    do {                                                                                          \
        gsl_error(reason, __FILE__, __LINE__, gsl_errno);                                         \
        return value;                                                                             \
    } while (0)
#define GSL_ERROR(reason, gsl_errno) GSL_ERROR_VAL(reason, gsl_errno, gsl_errno)
#define GSL_ERROR_NULL(reason, gsl_errno) GSL_ERROR_VAL(reason, gsl_errno, 0)
#define GSL_POSINF (1.0 / 0.0 * 1.0)
#define GSL_MAX(a, b) ((a) > (b) ? (a) : (b))
#define GSL_MIN(a, b) ((a) < (b) ? (a) : (b))
const char *gsl_strerror(const int gsl_errno);
/* default GSL behaviour: print and abort */
void gsl_error(const char *reason, const char *file, int line, int gsl_errno);
#endif
