/* Compile-time knobs and their defaults (values of reference src/define_defaults.h:24-86);
 * override with -D... exactly as with the reference. */
#ifndef DEFINE_DEFAULTS_H_
#define DEFINE_DEFAULTS_H_

#ifndef N_BETA
#define N_BETA 20 /* chains in the ladder */
#endif
#ifndef BETA_0
#define BETA_0 -0.001 /* hottest beta; < 0: derive from the calibrated step widths */
#endif
#ifndef BURN_IN_ITERATIONS
#define BURN_IN_ITERATIONS 10000
#endif
#ifndef ITER_LIMIT
#define ITER_LIMIT 100000
#endif
#ifndef MUL
#define MUL 0.85 /* step-width rescale factor */
#endif
#ifndef N_SWAP
#define N_SWAP -30 /* steps between swap attempts; < 0: 2000/N_BETA */
#endif
#ifndef PARAMS_FILENAME
#define PARAMS_FILENAME "params"
#endif
#ifndef DATA_FILENAME
#define DATA_FILENAME "data"
#endif
#ifndef TARGET_ACCEPTANCE_RATE
#define TARGET_ACCEPTANCE_RATE 0.50
#endif
#ifndef MAX_AR_DEVIATION
#define MAX_AR_DEVIATION 0.01
#endif

#endif
