/* Signal semantics of the run phase (reference src/parallel_tempering_run.c:25-59):
 * SIGINT finishes the current batch and stops, SIGUSR1/2 request a report. */
#include <signal.h>
#include <time.h>
#include "mcmc.h"
#include "parallel_tempering_run.h"

int run = 1;
int dumpflag = 0;

static void on_sigint(int signalnr) {
    printf("\nreceived Ctrl-C (%d). Stopping ... (please be patient)\n\n", signalnr);
    run = 0;
}

static void on_sigusr(int signalnr) {
    printf("\nreceived SIGUSR (%d). Will dump at next opportunity.\n\n", signalnr);
    signal(SIGUSR1, on_sigusr);
    signal(SIGUSR2, on_sigusr);
    dumpflag = 1;
}

void register_signal_handlers() {
    signal(SIGINT, on_sigint);
    signal(SIGUSR1, on_sigusr);
    signal(SIGUSR2, on_sigusr);
}

int get_duration() {
    static clock_t last = 0;
    clock_t before = last;
    last = clock();
    return (int)(last - before);
}

long unsigned int get_ticks_per_second() { return CLOCKS_PER_SEC; }
