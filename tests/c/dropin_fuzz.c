/* tests/c/dropin_fuzz.c -- TEST INFRASTRUCTURE.  Random traffic against the filter.h drop-in from many threads at once, for
 * the sanitizer runs on the CPU stub engine (tests/test_dropin_stub.py): every worker owns a few slaves and, for a fixed number of
 * steps, picks at random among what radiod's channel threads do to them -- create (COMPLEX / REAL / SPECTRUM, several sizes),
 * set_filter, execute_filter_output with a new or the old shift, flip isb, delete and re-create -- while the main thread keeps
 * writing blocks into the master until every worker is done.  Nothing is compared: the run must end, every call must return 0 (or
 * the documented -1), and ThreadSanitizer / AddressSanitizer / LeakSanitizer must stay silent.
 *   usage: dropin_fuzz <workers> <steps> <seed> */
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "ka9q_filter_abi.h"

static struct filter_in Master;
static int L = 25920, M = 6481;
static _Atomic int Done, Failed;
static int Steps;

static uint64_t rnd(uint64_t *s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }

#define NSL 3
static void *worker(void *a) {
  uint64_t s = 0x9E3779B97F4A7C15ull ^ ((uint64_t)(uintptr_t)a * 0xD1B54A32D192ED03ull);
  struct filter_out sl[NSL]; bool live[NSL] = {false, false, false}; int shift[NSL] = {0, 0, 0};
  memset(sl, 0, sizeof sl);
  static const int olens[4] = {240, 480, 160, 960};
  /* an optional private second filter (radiod's filter2, src/radio.c:1572-1594): a pooled inline master shared-nothing per worker, but
     the POOL behind it is shared by all of them */
  struct filter_in f2in; struct filter_out f2out; bool f2 = false;
  memset(&f2in, 0, sizeof f2in); memset(&f2out, 0, sizeof f2out);
  for (int step = 0; step < Steps && !atomic_load(&Failed); step++) {
    int const k = (int)(rnd(&s) % NSL);
    unsigned const op = (unsigned)(rnd(&s) % 100);
    if (!live[k]) {
      unsigned const kind = (unsigned)(rnd(&s) % 10);
      enum filtertype const t = kind == 0 ? SPECTRUM : (kind == 1 ? REAL : COMPLEX);
      int const olen = t == SPECTRUM ? 0 : olens[rnd(&s) % 4];
      memset(&sl[k], 0, sizeof sl[k]);
      if (create_filter_output(&sl[k], &Master, olen, t) != 0) { fprintf(stderr, "create_filter_output(%d,%d) failed\n", olen, (int)t); atomic_store(&Failed, 1); break; }
      if (t != SPECTRUM) {
        double const lo = -0.4 + 0.3 * (double)(rnd(&s) % 100) / 100.0;
        if (set_filter(&sl[k], lo, lo + 0.1 + 0.3 * (double)(rnd(&s) % 100) / 100.0, 3.0 + (double)(rnd(&s) % 8)) != 0) { fprintf(stderr, "set_filter failed\n"); atomic_store(&Failed, 1); break; }
      }
      live[k] = true; shift[k] = (int)(rnd(&s) % 24000) - 12000;
      continue;
    }
    if (op < 70) {                                   /* the common case: take the next block, sometimes retuned */
      if (op < 15) shift[k] = (int)(rnd(&s) % 40000) - 20000;
      if (execute_filter_output(&sl[k], shift[k]) != 0) { fprintf(stderr, "execute_filter_output failed\n"); atomic_store(&Failed, 1); break; }
      if (f2 && sl[k].out_type == COMPLEX && sl[k].olen == 240) {           /* through the second filter, as radio.c:1503-1513 does */
        int const r = write_cfilter(&f2in, sl[k].output.c, 240);
        if (r < 0) { fprintf(stderr, "filter2 write_cfilter failed\n"); atomic_store(&Failed, 1); break; }
        if (r > 0 && execute_filter_output(&f2out, 0) != 0) { fprintf(stderr, "filter2 execute_filter_output failed\n"); atomic_store(&Failed, 1); break; }
      }
    } else if (op < 74) {                              /* filter2 comes and goes */
      if (!f2) {
        memset(&f2in, 0, sizeof f2in); memset(&f2out, 0, sizeof f2out);
        if (create_filter_input(&f2in, 240, 273, COMPLEX) != 0) { fprintf(stderr, "filter2 create_filter_input failed\n"); atomic_store(&Failed, 1); break; }
        f2in.perform_inline = true;
        if (create_filter_output(&f2out, &f2in, 240, COMPLEX) != 0) { fprintf(stderr, "filter2 create_filter_output failed\n"); atomic_store(&Failed, 1); break; }
        if (set_filter(&f2out, -0.1, 0.1, 7.0) != 0) { fprintf(stderr, "filter2 set_filter failed\n"); atomic_store(&Failed, 1); break; }
        f2 = true;
      } else { delete_filter_output(&f2out); delete_filter_input(&f2in); f2 = false; }
    } else if (op < 80) {
      if (sl[k].out_type != SPECTRUM) {
        double const lo = -0.45 + 0.4 * (double)(rnd(&s) % 100) / 100.0;
        if (set_filter(&sl[k], lo, lo + 0.05 + 0.4 * (double)(rnd(&s) % 100) / 100.0, 2.0 + (double)(rnd(&s) % 10)) != 0) { fprintf(stderr, "set_filter failed\n"); atomic_store(&Failed, 1); break; }
      }
    } else if (op < 86) {
      /* callers flip it directly (src/radio.c:1586); a byte store cannot tear, and the drop-in reads it with a relaxed atomic load --
         written here as an atomic store so that the race detector watches the drop-in, not this line */
      if (sl[k].out_type == COMPLEX) __atomic_store_n((unsigned char *)&sl[k].isb, (unsigned char)!sl[k].isb, __ATOMIC_RELAXED);
    } else if (op < 94) {
      delete_filter_output(&sl[k]); live[k] = false;
    } else {
      usleep((useconds_t)(rnd(&s) % 3000));                            /* fall behind: laps and drops are part of the contract */
    }
  }
  for (int k = 0; k < NSL; k++) if (live[k]) delete_filter_output(&sl[k]);
  if (f2) { delete_filter_output(&f2out); delete_filter_input(&f2in); }
  atomic_fetch_add(&Done, 1);
  return NULL;
}

int main(int argc, char **argv) {
  int const nw = argc > 1 ? atoi(argv[1]) : 16;
  Steps = argc > 2 ? atoi(argv[2]) : 200;
  uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 0) : 1;
  memset(&Master, 0, sizeof Master);
  if (create_filter_input(&Master, L, M, REAL) != 0) { fprintf(stderr, "create_filter_input failed\n"); return 3; }
  struct notch_state *notch = calloc(2, sizeof *notch);
  notch[0].bin = 7; notch[0].alpha = 0.02; notch[1].bin = 0; notch[1].alpha = 0.01;        /* the list ends with bin 0 (src/filter.c:466) */
  Master.notches = notch;
  pthread_t *th = calloc((size_t)nw, sizeof *th);
  for (int i = 0; i < nw; i++) pthread_create(&th[i], NULL, worker, (void *)(uintptr_t)(seed * 1000 + (uint64_t)i + 1));
  float *blk = malloc(sizeof(float) * (size_t)L);
  uint64_t s = seed;
  long blocks = 0;
  while (atomic_load(&Done) < nw && blocks < 100000) {
    for (int i = 0; i < L; i++) blk[i] = (float)((double)(rnd(&s) % 2001) / 1000.0 - 1.0) * 0.05f;
    if (write_rfilter(&Master, blk, L) < 0) { fprintf(stderr, "write_rfilter failed\n"); atomic_store(&Failed, 1); break; }
    blocks++;
    usleep(300);
  }
  for (int i = 0; i < nw; i++) pthread_join(th[i], NULL);
  printf("blocks %ld workers %d steps %d failed %d\n", blocks, nw, Steps, atomic_load(&Failed));
  delete_filter_input(&Master);
  free(th); free(blk); free(notch);
  return atomic_load(&Failed) ? 2 : 0;
}
