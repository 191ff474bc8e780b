/* Allocation macros with the reference's names (src/memory.h:40-56).  Plain malloc; the
 * optional Boehm collector of the reference is not supported (define nothing). */
#ifndef MEMORY_H_
#define MEMORY_H_

#include <stdlib.h>
#include "debug.h"

#define FREEMSG(x) IFSEGV dump_p("about to free", (void *)x);
#define mem_malloc(x) malloc(x)
#define mem_calloc(n, x) calloc(n, x)
#define mem_realloc(p, x) realloc(p, x)
#define mem_free(x)                                                                              \
    {                                                                                            \
        FREEMSG(x);                                                                              \
        free((void *)x);                                                                         \
    }

#endif
