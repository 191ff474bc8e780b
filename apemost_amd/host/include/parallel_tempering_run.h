#ifndef _PARALLEL_TEMPERING_RUN
#define _PARALLEL_TEMPERING_RUN

extern int run;      /* cleared by SIGINT: finish the current batch and stop */
extern int dumpflag; /* set by SIGUSR1/2: report at the next opportunity */

int get_duration();
void register_signal_handlers();
long unsigned int get_ticks_per_second();

#endif
