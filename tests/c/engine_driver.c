/* tests/c/engine_driver.c -- TEST INFRASTRUCTURE.  Exercises the threading inside the engine (libchz_hip.so's C ABI) for the
 * ThreadSanitizer run of tests/test_engine_emulated.py: blocks pipelined over 4 lanes from 1, 2 or 4 issuing threads
 * (CHZ_ENQ_THREADS), the notch hand-over between them, a plain bank and a tuned bank with noise estimate and demodulators (the
 * in-order demodulator stream with its own hand-over), retunes / response swaps / tuning changes between runs, and a pool of inline
 * masters executed from two caller threads at once.  Nothing numeric is checked here (the parity tests do that); the run must end
 * with every call succeeding and the race detector silent. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "chz_engine.h"

#define OK(call) do { if ((call) != 0) { fprintf(stderr, "%s: %s\n", #call, chz_last_error()); exit(2); } } while (0)

static chz_mini *Pool;
static void *mini_thread(void *a) {
  int const base = (int)(long)a;
  enum { N = 512, L2 = 240, K = 8 };
  float *win[K], *out[K]; int inst[K];
  for (int i = 0; i < K; i++) {
    inst[i] = base + i;
    win[i] = calloc((size_t)2 * N, sizeof(float)); out[i] = calloc((size_t)2 * L2, sizeof(float));
    for (int k = 0; k < 2 * N; k++) win[i][k] = (float)((k * 7 + i) % 13) * 0.01f;
  }
  for (int r = 0; r < 5; r++) OK(chz_mini_execute(Pool, K, inst, (const float *const *)win, NULL, NULL, out));
  for (int i = 0; i < K; i++) { free(win[i]); free(out[i]); }
  return NULL;
}

int main(void) {
  chz_engine *e = NULL;
  int const L = 25920, M = 6481, P = 300, olen = 240, nch = 8;
  if (getenv("CHZ_ENQ_THREADS")) OK(chz_set_option("enq_threads", getenv("CHZ_ENQ_THREADS")));      /* (options are an API, not environment variables) */
  OK(chz_engine_create(&e, L, M, CHZ_REAL, 0, NULL, 8));
  int const plain = chz_bank_create(e, P, olen, nch), tuned = chz_bank_create(e, P, olen, nch);
  if (plain < 0 || tuned < 0) { fprintf(stderr, "bank: %s\n", chz_last_error()); return 2; }
  float *resp = calloc((size_t)nch * P * 2, sizeof(float));
  for (int i = 0; i < nch * P; i++) resp[2 * i] = 1.0f / P;
  int *shifts = calloc((size_t)nch, sizeof(int)); double *freq = calloc((size_t)nch, sizeof(double));
  for (int i = 0; i < nch; i++) { shifts[i] = 100 + 37 * i; freq[i] = -3.3 / 12000.0; }
  OK(chz_bank_set_responses(e, plain, 0, nch, resp)); OK(chz_bank_set_shifts(e, plain, 0, nch, shifts)); OK(chz_bank_set_active(e, plain, nch));
  OK(chz_bank_set_responses(e, tuned, 0, nch, resp)); OK(chz_bank_set_tuning(e, tuned, 0, 0, nch, shifts, freq, NULL)); OK(chz_bank_set_active(e, tuned, nch));
  OK(chz_bank_enable_noise(e, tuned, 1.296e6));
  OK(chz_bank_set_pcm_stride(e, tuned, 2 * olen));
  chz_demod_params dp; memset(&dp, 0, sizeof dp);
  dp.channels = 1; dp.agc = 1; dp.encoding = CHZ_PCM_S16BE; dp.squelch_tail = 1; dp.tuned = 1; dp.kind = CHZ_DEMOD_LINEAR;
  dp.samprate = 12000; dp.headroom = 0.18; dp.threshold = 0.18; dp.recovery_rate = 10; dp.hangtime = 1.1; dp.bandwidth = 2950; dp.squelch_open = 6.3; dp.squelch_close = 5;
  dp.gain = 300;
  chz_demod_params *dps = calloc((size_t)nch, sizeof *dps);
  for (int i = 0; i < nch; i++) { dps[i] = dp; if (i % 4 == 1) { dps[i].kind = CHZ_DEMOD_FM; dps[i].bandwidth = 8000; dps[i].gain = 1; } }
  OK(chz_bank_set_demod(e, tuned, 0, 0, nch, dps, 0.02));
  int nb[2] = {5, 0}; OK(chz_set_notches(e, nb, 2, 0.01));
  float *x = calloc((size_t)8 * L, sizeof(float));
  for (long i = 0; i < 8L * L; i++) x[i] = ((float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f) * 0.05f;
  OK(chz_input_write(e, x, 8L * L - (M - 1))); OK(chz_input_write(e, x + (8L * L - (M - 1)), M - 1));

  OK(chz_mini_create(&Pool, 240, 273, 16, 0));
  for (int i = 0; i < 16; i++) { if (chz_mini_add(Pool) != i) { fprintf(stderr, "mini_add\n"); return 2; } }
  float *r2 = calloc((size_t)2 * 512, sizeof(float)); for (int i = 0; i < 512; i++) r2[2 * i] = 1.0f / 512;
  for (int i = 0; i < 16; i++) OK(chz_mini_set_response(Pool, i, r2));
  pthread_t mt[2];
  pthread_create(&mt[0], NULL, mini_thread, (void *)0L); pthread_create(&mt[1], NULL, mini_thread, (void *)8L);

  chz_timing t;
  unsigned job = 0;
  void *pcm = calloc((size_t)nch, (size_t)2 * olen); chz_demod_status *st = calloc((size_t)nch, sizeof *st);
  for (int it = 0; it < 4; it++) {
    OK(chz_run_blocks(e, job, 8, 0, it == 3, &t));
    job += 8;
    for (int i = 0; i < 5; i++) shifts[(it * 5 + i) % nch] += 3;
    int const c0 = (it * 5) % (nch - 5 > 0 ? nch - 5 : 1);
    OK(chz_bank_set_shifts(e, plain, c0, 5, shifts + c0));
    OK(chz_bank_set_responses(e, plain, it % nch, 1, resp));
    OK(chz_bank_set_tuning(e, tuned, job, c0, 5, shifts + c0, freq + c0, NULL));
    OK(chz_bank_read_pcm(e, tuned, (int)((job - 1) % 4), 0, nch, pcm, st));
    OK(chz_engine_check(e));
  }
  pthread_join(mt[0], NULL); pthread_join(mt[1], NULL);
  printf("driver ok blocks %u\n", job);
  chz_mini_destroy(Pool);
  chz_engine_destroy(e);
  free(resp); free(shifts); free(freq); free(dps); free(x); free(r2); free(pcm); free(st);
  return 0;
}
