/* Debug print macros with the reference's names (src/debug.h:34-104). */
#ifndef DEBUGHELPER
#define DEBUGHELPER

#include <stdio.h>

#ifdef DEBUG
#define IFDEBUG if (1)
#else
#define IFDEBUG if (0)
#endif
#ifdef SEGV
#define IFSEGV if (1)
#else
#define IFSEGV if (0)
#endif
#ifdef WAIT
#define IFWAIT if (1)
#else
#define IFWAIT if (0)
#endif
#ifdef VERBOSE
#define IFVERBOSE if (1)
#else
#define IFVERBOSE if (0)
#endif

#include "memory.h"
#include "mcmc.h"

#define STRINGIFY(x) #x
#define TOSTRING(x) STRINGIFY(x)
#define AT __FILE__ ":" TOSTRING(__LINE__)

#define APEMOST_DBG(fmt, str, var)                                                               \
    IFDEBUG {                                                                                    \
        printf("\tDEBUG[%s]: %s: " fmt "\n", AT, str, var);                                      \
        fflush(NULL);                                                                            \
    }
#define debug(str)                                                                               \
    IFDEBUG {                                                                                    \
        printf("\tDEBUG[%s]: %s\n", AT, str);                                                    \
        fflush(NULL);                                                                            \
    }
#define dump_i(str, var) APEMOST_DBG("%i", str, var)
#define dump_ui(str, var) APEMOST_DBG("%u", str, var)
#define dump_d(str, var) APEMOST_DBG("%f", str, var)
#define dump_l(str, var) APEMOST_DBG("%ld", str, var)
#define dump_ul(str, var) APEMOST_DBG("%lu", str, var)
#define dump_size(str, var) APEMOST_DBG("%lu", str, (unsigned long)var)
#define dump_s(str, var) APEMOST_DBG("%s", str, var)
#define dump_p(str, var) APEMOST_DBG("%p", str, var)
#define dump_i_s(str, index, var)                                                                \
    IFDEBUG {                                                                                    \
        printf("\tDEBUG[%s]: %s[%i]: %s\n", AT, str, index, var);                                \
        fflush(NULL);                                                                            \
    }
#define dump_v(str, v)                                                                           \
    IFDEBUG {                                                                                    \
        printf("\tDEBUG[%s]: %s: ", AT, str);                                                    \
        dump_vectorln(v);                                                                        \
        fflush(NULL);                                                                            \
    }

void dump_mcmc(const mcmc *m);
void dump_vector(const gsl_vector *v);
void dump_vectorln(const gsl_vector *v);

#define require(x) (x)

#endif
