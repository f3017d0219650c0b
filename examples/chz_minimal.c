/* examples/chz_minimal.c -- the C ABI of include/chz_engine.h from plain C, no Python, no torch.
 *
 * One 129.6 MS/s real master (L = 2,592,000, M = 648,001), one 12 kHz channel (P = 300, olen = 240) tuned to a
 * bin-centred carrier.  Known answer: x[n] = a cos(2 pi k0 n / N) -> X[k0] = a N / 2; with a response whose
 * DC bin is sqrt(2)/N (what set_filter's gain normalisation produces for a real master, src/filter.c:1024-1028) every
 * output sample of the block has magnitude a / sqrt(2).
 *
 *   gcc -O2 -I include examples/chz_minimal.c -L ka9q-radio_amd -lchz_hip -Wl,-rpath,$PWD/ka9q-radio_amd -lm -o chz_minimal
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "chz_engine.h"

#define CHECK(call) do { if ((call) < 0) { fprintf(stderr, "%s: %s\n", #call, chz_last_error()); return 1; } } while (0)

int main(void) {
  const int L = 2592000, M = 648001, N = L + M - 1, P = 300, olen = 240;
  const int k0 = 250000;                       /* 10.000 MHz: bin-centred (40 Hz bins) */
  const double a = 0.1;
  chz_engine *e = NULL;
  CHECK(chz_engine_create(&e, L, M, CHZ_REAL, 0, NULL, 8));
  chz_info info;
  CHECK(chz_engine_info(e, &info));
  printf("plan: %s\n", info.plan);

  int bank = chz_bank_create(e, P, olen, 1);
  if (bank < 0) { fprintf(stderr, "%s\n", chz_last_error()); return 1; }
  /* a flat response over the whole channel with set_filter's gain: H[k] = sqrt(2)/N */
  float *resp = calloc((size_t)2 * P, sizeof *resp);
  for (int k = 0; k < P; k++) resp[2 * k] = (float)(sqrt(2.0) / N);
  CHECK(chz_bank_set_responses(e, bank, 0, 1, resp));
  CHECK(chz_bank_set_shifts(e, bank, 0, 1, &k0));
  CHECK(chz_bank_set_active(e, bank, 1));

  float *x = malloc(sizeof *x * (size_t)L);
  float *out = malloc(sizeof *out * 2 * (size_t)olen);
  double worst = 0;
  for (unsigned job = 0; job < 3; job++) {
    for (int n = 0; n < L; n++) {
      /* phase-continuous across blocks: absolute sample index job*L + n (k0*n reduced mod N in integers) */
      const long long idx = ((long long)job * L + n) % N * (long long)k0 % N;
      x[n] = (float)(a * cos(2.0 * M_PI * (double)idx / N));
    }
    CHECK(chz_input_write(e, x, L));
    CHECK(chz_step(e, job));                   /* forward transform + every bank */
    CHECK(chz_bank_read(e, bank, 0, 1, out));
    if (job == 0) continue;                    /* the first window starts with M-1 zeros: not a steady tone yet */
    for (int n = 0; n < olen; n++) {
      const double mag = hypot(out[2 * n], out[2 * n + 1]);
      const double err = fabs(mag - a / sqrt(2.0)) / (a / sqrt(2.0));
      if (err > worst) worst = err;
    }
  }
  printf("bin-centred carrier: |y| = a/sqrt(2) within %.2e (relative)\n", worst);
  chz_engine_destroy(e);
  free(resp); free(x); free(out);
  return worst < 1e-4 ? 0 : 2;
}
