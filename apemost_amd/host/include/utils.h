#ifndef MCMC_UTILS
#define MCMC_UTILS

#include <stdio.h>
#include <stdlib.h>

#ifdef NOASSERT
#define assert(cond)
#else
#include <assert.h>
#endif

FILE *openfile(const char *filename); /* exits on failure */
unsigned int countlines(const char *filename);
unsigned int get_column_count(const char *filename); /* tokens on the first line */

#endif
