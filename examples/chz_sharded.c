/* examples/chz_sharded.c -- BASELINE config 4 from plain C: one process per GPU, rank 0 owns the front end and the forward
 * transform, the block spectrum travels through RCCL (behind the engine's C ABI) on the slot's own HIP stream, every rank runs
 * its own 24 kHz channels.  No Python, no torch; the ranks meet through a rendezvous file.
 *
 *   gcc -O2 -I include examples/chz_sharded.c -L ka9q-radio_amd -lchz_hip -Wl,-rpath,$PWD/ka9q-radio_amd -lm -o chz_sharded
 *   for r in 0 1 2 3 4 5 6 7; do ./chz_sharded $r 8 /tmp/chz_id $r & done; wait        # args: rank world idfile device
 *
 * Every rank tunes its first channel to the same bin-centred carrier, so the known answer of chz_minimal.c holds on every GPU:
 * |y| = a / sqrt(2).  On ranks other than 0 that can only come out right if the spectrum really arrived. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "chz_engine.h"

#define CHECK(call) do { if ((call) < 0) { fprintf(stderr, "rank %d: %s: %s\n", rank, #call, chz_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s rank world idfile [device]\n", argv[0]); return 1; }
  const int rank = atoi(argv[1]), world = atoi(argv[2]), device = argc > 4 ? atoi(argv[4]) : rank;
  const int L = 2592000, M = 648001, N = L + M - 1, P = 600, olen = 480, nch = 1024;
  const int k0 = 250000;                       /* 10.000 MHz, bin-centred */
  const double a = 0.1;
  chz_engine *e = NULL;
  chz_comm *comm = NULL;
  CHECK(chz_engine_create(&e, L, M, CHZ_REAL, device, NULL, 8));
  CHECK(chz_comm_create_file(&comm, rank, world, argv[3], device, 120.0));

  int bank = chz_bank_create(e, P, olen, nch);
  if (bank < 0) { fprintf(stderr, "%s\n", chz_last_error()); return 1; }
  float *resp = calloc((size_t)2 * P * nch, sizeof *resp);
  int *shifts = malloc(sizeof *shifts * nch);
  for (int c = 0; c < nch; c++) {
    for (int k = 0; k < P; k++) resp[((size_t)c * P + k) * 2] = (float)(sqrt(2.0) / N);
    /* config 4's raster: 0.5 MHz + i * 7.8 kHz, rank r owns channels [r*nch, (r+1)*nch); channel 0 of every rank -> the carrier */
    shifts[c] = c == 0 ? k0 : (int)lrint((0.5e6 + ((double)rank * nch + c) * 7.8e3) / (129.6e6 / N));
  }
  CHECK(chz_bank_set_responses(e, bank, 0, nch, resp));
  CHECK(chz_bank_set_shifts(e, bank, 0, nch, shifts));
  CHECK(chz_bank_set_active(e, bank, nch));

  float *x = malloc(sizeof *x * (size_t)L);
  float *out = malloc(sizeof *out * 2 * (size_t)olen);
  double worst = 0;
  chz_timing t;
  for (unsigned job = 0; job < 3; job++) {
    if (rank == 0) {                           /* only the root has a front end */
      for (int n = 0; n < L; n++) {
        const long long idx = ((long long)job * L + n) % N * (long long)k0 % N;
        x[n] = (float)(a * cos(2.0 * M_PI * (double)idx / N));
      }
      CHECK(chz_input_write(e, x, L));
    }
    CHECK(chz_run_blocks_sharded(e, comm, 0, 0, NULL, NULL, job, 1, &t));       /* forward on the root, ncclBroadcast, own channels */
    CHECK(chz_bank_read(e, bank, 0, 1, out));
    if (job == 0) continue;
    for (int n = 0; n < olen; n++) {
      const double mag = hypot(out[2 * n], out[2 * n + 1]);
      const double err = fabs(mag - a / sqrt(2.0)) / (a / sqrt(2.0));
      if (err > worst) worst = err;
    }
  }
  double v = worst;
  CHECK(chz_comm_allreduce_max(comm, &v, 1));
  printf("rank %d of %d: |y| = a/sqrt(2) within %.2e here, %.2e on the worst rank; last block %.3f ms\n", rank, world, worst, v, t.total_ms);
  chz_comm_destroy(comm);
  chz_engine_destroy(e);
  free(resp); free(shifts); free(x); free(out);
  return v < 1e-4 ? 0 : 2;
}
