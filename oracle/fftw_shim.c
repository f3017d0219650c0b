/* oracle/fftw_shim.c -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 *
 * Implements the FFTW3 single-precision entry points that the reference's
 * src/filter.c leaves unresolved (fftwf_plan_dft_1d, _r2c_1d, _c2r_1d,
 * fftwf_execute, _execute_dft, _execute_dft_r2c, fftwf_destroy_plan,
 * fftwf_plan_with_nthreads, fftwf_init_threads, fftwf_import_system_wisdom,
 * fftwf_import_wisdom_from_filename, fftwf_version) on top of oracle/dft.c, so
 * that oracle/_ref/libka9q_ref.so = the reference's own filter.c, compiled
 * unmodified, with only the FFT butterflies supplied by this project.
 *
 * FFTW3 itself is a third-party dependency that is neither vendored in
 * /root/reference nor installed in this image (debian/control:7,
 * src/Makefile:290; version unpinned, docs/FFTW3.md:137-140).
 *
 * Precision is chosen per process with oracle_fft_set_precision():
 *   0 (default) float64 arithmetic, one rounding to float32  -> parity oracle
 *   1           float32 arithmetic                           -> CPU timing baseline
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <complex.h>
#include "shims/fftw3.h"
#include "dft.h"

const char fftwf_version[] = "oracle-dft-shim (not FFTW; float64 mixed-radix DFT by definition)";

static int Precision = ODFT_F64;
static int Plan_threads;      /* fftwf_plan_with_nthreads, below */
void oracle_fft_set_precision(int p) { Precision = p ? ODFT_F32 : ODFT_F64; }
int oracle_fft_get_precision(void) { return Precision; }

enum kind { K_C2C, K_R2C, K_C2R };

struct fftwf_plan_s {
  enum kind kind;
  int n;
  int sign;
  void *in, *out;     /* arrays given at planning time (used by fftwf_execute) */
  odft_plan *dft;
};

static fftwf_plan mkplan(enum kind k, int n, int sign, void *in, void *out) {
  if (n < 1) return NULL;
  struct fftwf_plan_s *p = calloc(1, sizeof *p);
  if (!p) return NULL;
  p->kind = k; p->n = n; p->sign = sign; p->in = in; p->out = out;
  p->dft = odft_create(n, Precision);
  if (!p->dft) { free(p); return NULL; }
  odft_set_threads(p->dft, Plan_threads);
  odft_warm(p->dft, k != K_C2C && k != K_C2R);
  return p;
}

/* FFTW_WISDOM_ONLY is honoured as "wisdom present": returning NULL would only
   make the reference retry with FFTW_ESTIMATE and log (src/filter.c:106-119). */
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags) {
  (void)flags; return mkplan(K_C2C, n, sign, in, out);
}
fftwf_plan fftwf_plan_dft_r2c_1d(int n, float *in, fftwf_complex *out, unsigned flags) {
  (void)flags; return mkplan(K_R2C, n, FFTW_FORWARD, in, out);
}
fftwf_plan fftwf_plan_dft_c2r_1d(int n, fftwf_complex *in, float *out, unsigned flags) {
  (void)flags; return mkplan(K_C2R, n, FFTW_BACKWARD, in, out);
}

void fftwf_execute_dft(const fftwf_plan p, fftwf_complex *in, fftwf_complex *out) {
  odft_c2c(p->dft, (const float *)in, (float *)out, p->sign);
}
void fftwf_execute_dft_r2c(const fftwf_plan p, float *in, fftwf_complex *out) {
  odft_r2c(p->dft, in, (float *)out);
}
void fftwf_execute_dft_c2r(const fftwf_plan p, fftwf_complex *in, float *out) {
  odft_c2r(p->dft, (const float *)in, out);
}
void fftwf_execute(const fftwf_plan p) {
  switch (p->kind) {
  case K_C2C: fftwf_execute_dft(p, p->in, p->out); break;
  case K_R2C: fftwf_execute_dft_r2c(p, p->in, p->out); break;
  case K_C2R: fftwf_execute_dft_c2r(p, p->in, p->out); break;
  }
}
void fftwf_destroy_plan(fftwf_plan p) {
  if (!p) return;
  odft_destroy(p->dft);
  free(p);
}
int fftwf_init_threads(void) { return 1; }
/* plans made after this call run on `nthreads` threads (FFTW's contract; the reference passes fft-internal-threads, src/filter.c:131-133):
   honoured by the float32 timing path's long transform (oracle/dft.c), ignored by the float64 parity oracle */
void fftwf_plan_with_nthreads(int nthreads) { Plan_threads = nthreads < 1 ? 1 : nthreads; }
int fftwf_import_system_wisdom(void) { return 1; }
int fftwf_import_wisdom_from_filename(const char *filename) { (void)filename; return 1; }
void *fftwf_malloc(size_t n) { void *p = NULL; return posix_memalign(&p, 64, n ? n : 64) == 0 ? p : NULL; }
float *fftwf_alloc_real(size_t n) { return fftwf_malloc(n * sizeof(float)); }
fftwf_complex *fftwf_alloc_complex(size_t n) { return fftwf_malloc(n * sizeof(fftwf_complex)); }
void fftwf_free(void *p) { free(p); }
