/* tests/c/exit_midstream.c -- TEST INFRASTRUCTURE.  radiod ends through exit() from a signal handler's closedown() (src/main.c) without deleting a single filter:
 * the front-end thread is still writing blocks and a thousand channel threads sit in execute_filter_output() when the process runs its exit handlers.  The library
 * under it must let that happen: no hang in a destructor, no crash.  This program does exactly that on the drop-in: a front-end thread at wall-clock pace, `nch`
 * channel threads, exit(0) from the main thread after `ms` milliseconds.      usage: exit_midstream [nch] [ms] */
#include <complex.h>
#include <execinfo.h>
#include <signal.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "ka9q_filter_abi.h"

static struct filter_in Master;
enum { L = 259200, M = 64801 };                 /* 12.96 MS/s real, 20 ms blocks */

static void on_segv(int sig) { void *bt[48]; int n = backtrace(bt, 48); fprintf(stderr, "exit_midstream: signal %d\n", sig); backtrace_symbols_fd(bt, n, 2); _exit(70); }
static void *front_end(void *arg) {
  (void)arg;
  unsigned seed = 1;
  for (;;) {
    float *w = Master.input_write_pointer.r;
    for (int i = 0; i < L; i++) { seed = seed * 1664525u + 1013904223u; w[i] = (float)(int)(seed >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    write_rfilter(&Master, NULL, L);
    usleep(20000);
  }
  return NULL;
}
static void *channel(void *arg) {
  int const k = (int)(long)arg;
  struct filter_out out;
  memset(&out, 0, sizeof out);
  if (create_filter_output(&out, &Master, 240, COMPLEX) != 0) { fprintf(stderr, "create_filter_output failed\n"); _exit(3); }
  set_filter(&out, -0.4, 0.4, 11.0);
  for (;;) execute_filter_output(&out, 1000 + 37 * k);
  return NULL;
}
int main(int argc, char **argv) {
  int const nch = argc > 1 ? atoi(argv[1]) : 256, ms = argc > 2 ? atoi(argv[2]) : 500;
  signal(SIGSEGV, on_segv); signal(SIGABRT, on_segv);
  if (create_filter_input(&Master, L, M, REAL) != 0) { fprintf(stderr, "create_filter_input failed\n"); return 2; }
  pthread_t t;
  for (long k = 0; k < nch; k++) pthread_create(&t, NULL, channel, (void *)k);
  pthread_create(&t, NULL, front_end, NULL);
  usleep(1000 * ms);
  fprintf(stderr, "exit_midstream: %u blocks in, leaving through exit()\n", Master.next_jobnum);
  exit(0);
}
