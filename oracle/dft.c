/* oracle/dft.c -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 * See dft.h / dft_core.h.  Restates the DFT definition FFTW3 implements for the
 * reference's call sites src/filter.c:106,127,148,505,508,573,582,914,1007,1030.
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <alloca.h>
#include "dft.h"

#define DFT_REAL double
#define DFT_NAME(x) d64_##x
#include "dft_core.h"
#undef DFT_REAL
#undef DFT_NAME

#define DFT_REAL float
#define DFT_NAME(x) f32_##x
#include "dft_core.h"
#undef DFT_REAL
#undef DFT_NAME
#include <pthread.h>
#include "dft_batch.h"

/* Cache-friendly "four-step" path for the float32 CPU baseline: a long transform of length
   h = n1*n2 becomes n2 column transforms of length n1 (gathered 8 columns at a time so every cache
   line fetched is used), a twiddle, and n1 contiguous row transforms of length n2 written back
   transposed in blocks of 8.  Only the timing leg uses it; the float64 oracle stays on the plain
   recursion. */
struct four_step {
  int h, n1, n2;
  f32_plan *p1, *p2;
  double *wstep;     /* W_h^{j2} (re,im), j2 = 0..n2-1: the per-column twiddle step */
};

struct odft_plan {
  int n;
  int precision;
  int threads;        /* float32 timing path only: threads of the long four-step transform (odft_set_threads) */
  struct four_step *fs;   /* for the length-n/2 complex transform behind the float32 r2c, when n is large */
  d64_plan *full64;   /* length n   (lazily built) */
  d64_plan *half64;   /* length n/2 (lazily built, even n only) */
  f32_plan *full32;
  f32_plan *half32;
  double *rtw;        /* W_n^k, k = 0..n/2, interleaved (re,im), forward sign; lazily built */
};

static struct four_step *four_step_create(int h) {
  if (h < 65536) return NULL;
  int best = 0;
  for (int a = 2; (long)a * a <= h; a++) if (h % a == 0) best = a;     /* n2 = largest divisor <= sqrt(h) */
  if (best < 64) return NULL;
  struct four_step *fs = (struct four_step *)calloc(1, sizeof *fs);
  fs->h = h; fs->n2 = best; fs->n1 = h / best;
  fs->p1 = f32_plan_create(fs->n1); fs->p2 = f32_plan_create(fs->n2);
  fs->wstep = (double *)malloc(sizeof(double) * 2 * (size_t)fs->n2);
  for (int j = 0; j < fs->n2; j++) { double s, c; sincos(-2.0 * M_PI * j / h, &s, &c); fs->wstep[2 * j] = c; fs->wstep[2 * j + 1] = s; }
  return fs;
}
static void four_step_destroy(struct four_step *fs) {
  if (!fs) return;
  f32_plan_destroy(fs->p1); f32_plan_destroy(fs->p2); free(fs->wstep); free(fs);
}
/* ---- a few threads for the long float32 transform (round 6): FFTW runs a plan made after fftwf_plan_with_nthreads(n) on n threads
   (the reference's fft-internal-threads, src/filter.c:131-133); the shim hands that n to the plan (odft_set_threads) and the two passes of
   the four-step transform -- independent groups of columns, then of rows -- and the real-transform unpacking are split over that many. */
struct par_job { void (*fn)(void *ctx, long lo, long hi); void *ctx; long lo, hi; };
static void *par_thread(void *a) { struct par_job *j = (struct par_job *)a; j->fn(j->ctx, j->lo, j->hi); return NULL; }
static void par_for(int threads, long n, void (*fn)(void *ctx, long lo, long hi), void *ctx) {
  if (threads > 16) threads = 16;
  if (threads < 2 || n < 2 * threads) { fn(ctx, 0, n); return; }
  pthread_t th[16]; struct par_job jb[16];
  for (int t = 0; t < threads; t++) {
    jb[t].fn = fn; jb[t].ctx = ctx; jb[t].lo = n * t / threads; jb[t].hi = n * (t + 1) / threads;
    if (t + 1 < threads) pthread_create(&th[t], NULL, par_thread, &jb[t]);
  }
  par_thread(&jb[threads - 1]);
  for (int t = 0; t + 1 < threads; t++) pthread_join(th[t], NULL);
}

struct fs_ctx { const struct four_step *fs; const f32_cpx *in; f32_cpx *T; f32_cpx *out; int v8; };
#define FS_TB 8
/* columns: Y[k1][j2] = sum_j1 x[j1*n2 + j2] W_n1^(j1 k1), then times W_h^(j2 k1); groups [g0, g1) of 8 columns */
static void fs_cols_scalar(void *vc, long g0, long g1) {
  struct fs_ctx *c = (struct fs_ctx *)vc;
  const struct four_step *fs = c->fs;
  const int n1 = fs->n1, n2 = fs->n2, TB = FS_TB;
  f32_cpx *col = (f32_cpx *)malloc(sizeof(f32_cpx) * (size_t)TB * n1 * 2);
  f32_cpx *res = col + (size_t)TB * n1;
  for (long g = g0; g < g1; g++) {
    const int j0 = (int)g * TB, tb = n2 - j0 < TB ? n2 - j0 : TB;
    for (int j1 = 0; j1 < n1; j1++)
      for (int t = 0; t < tb; t++) col[(size_t)t * n1 + j1] = c->in[(size_t)j1 * n2 + j0 + t];
    for (int t = 0; t < tb; t++) {
      f32_execute(fs->p1, col + (size_t)t * n1, res + (size_t)t * n1, -1);
      /* twiddle by recurrence in double: w_{k1} = W_h^{(j0+t) k1} */
      const double sr = fs->wstep[2 * (j0 + t)], si = fs->wstep[2 * (j0 + t) + 1];
      double wr = 1.0, wi = 0.0;
      f32_cpx *r = res + (size_t)t * n1;
      for (int k1 = 0; k1 < n1; k1++) {
        const float a = r[k1].re, b = r[k1].im;
        r[k1].re = (float)(a * wr - b * wi); r[k1].im = (float)(a * wi + b * wr);
        const double nr = wr * sr - wi * si; wi = wr * si + wi * sr; wr = nr;
      }
    }
    for (int k1 = 0; k1 < n1; k1++)
      for (int t = 0; t < tb; t++) c->T[(size_t)k1 * n2 + j0 + t] = res[(size_t)t * n1 + k1];
  }
  free(col);
}
/* rows: Z[k1][k2] = sum_j2 T[k1][j2] W_n2^(j2 k2), out[k1 + n1*k2]; groups of 8 rows */
static void fs_rows_scalar(void *vc, long g0, long g1) {
  struct fs_ctx *c = (struct fs_ctx *)vc;
  const struct four_step *fs = c->fs;
  const int n1 = fs->n1, n2 = fs->n2, TB = FS_TB;
  f32_cpx *res = (f32_cpx *)malloc(sizeof(f32_cpx) * (size_t)TB * n2);
  for (long g = g0; g < g1; g++) {
    const int k0 = (int)g * TB, tb = n1 - k0 < TB ? n1 - k0 : TB;
    for (int t = 0; t < tb; t++) f32_execute(fs->p2, c->T + (size_t)(k0 + t) * n2, res + (size_t)t * n2, -1);
    for (int k2 = 0; k2 < n2; k2++)
      for (int t = 0; t < tb; t++) c->out[(size_t)k0 + t + (size_t)n1 * k2] = res[(size_t)t * n2 + k2];
  }
  free(res);
}
/* the same two passes with the 8 transforms of a group side by side in the lanes of AVX2 vectors (dft_batch.h) */
typedef float v4f __attribute__((vector_size(16)));
V8_TARGET static void fs_cols_v8(void *vc, long g0, long g1) {
  struct fs_ctx *c = (struct fs_ctx *)vc;
  const struct four_step *fs = c->fs;
  const int n1 = fs->n1, n2 = fs->n2;
  v8c *col = NULL;
  if (posix_memalign((void **)&col, 64, sizeof(v8c) * (size_t)n1 * 2) != 0) return;
  v8c *res = col + n1;
  for (long g = g0; g < g1; g++) {
    const int j0 = (int)g * 8, tb = n2 - j0 < 8 ? n2 - j0 : 8;
    for (int j1 = 0; j1 < n1; j1++) {
      const f32_cpx *src = c->in + (size_t)j1 * n2 + j0;
      v8f re = {0, 0, 0, 0, 0, 0, 0, 0}, im = re;
      for (int t = 0; t < tb; t++) { re[t] = src[t].re; im[t] = src[t].im; }
      col[j1].re = re; col[j1].im = im;
    }
    v8_rec(fs->p1, 0, n1, col, 1, res, -1);
    /* twiddles W_h^{(j0+t) k1} by recurrence in double, one column per lane (two 4-wide double vectors) */
    v4d sr[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, si[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, wr[2] = {{1, 1, 1, 1}, {1, 1, 1, 1}}, wi[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int q = 0; q < 2; q++)
      for (int t = 0; t < 4; t++) {
        const int j = j0 + 4 * q + t < n2 ? j0 + 4 * q + t : 0;
        sr[q][t] = fs->wstep[2 * j]; si[q][t] = fs->wstep[2 * j + 1]; wr[q][t] = 1.0; wi[q][t] = 0.0;
      }
    for (int k1 = 0; k1 < n1; k1++) {
      union { v8f v; v4f h[2]; } fr, fi;
      fr.h[0] = __builtin_convertvector(wr[0], v4f); fr.h[1] = __builtin_convertvector(wr[1], v4f);
      fi.h[0] = __builtin_convertvector(wi[0], v4f); fi.h[1] = __builtin_convertvector(wi[1], v4f);
      const v8f a = res[k1].re, b = res[k1].im;
      const v8f yr = a * fr.v - b * fi.v, yi = a * fi.v + b * fr.v;
      f32_cpx *dst = c->T + (size_t)k1 * n2 + j0;
      for (int t = 0; t < tb; t++) { dst[t].re = yr[t]; dst[t].im = yi[t]; }
      for (int q = 0; q < 2; q++) { const v4d nr = wr[q] * sr[q] - wi[q] * si[q]; wi[q] = wr[q] * si[q] + wi[q] * sr[q]; wr[q] = nr; }
    }
  }
  free(col);
}
V8_TARGET static void fs_rows_v8(void *vc, long g0, long g1) {
  struct fs_ctx *c = (struct fs_ctx *)vc;
  const struct four_step *fs = c->fs;
  const int n1 = fs->n1, n2 = fs->n2;
  v8c *col = NULL;
  if (posix_memalign((void **)&col, 64, sizeof(v8c) * (size_t)n2 * 2) != 0) return;
  v8c *res = col + n2;
  for (long g = g0; g < g1; g++) {
    const int k0 = (int)g * 8, tb = n1 - k0 < 8 ? n1 - k0 : 8;
    /* 8 rows read side by side (8 sequential streams), one element of each per vector */
    for (int j2 = 0; j2 < n2; j2++) {
      v8f re = {0, 0, 0, 0, 0, 0, 0, 0}, im = re;
      for (int t = 0; t < tb; t++) { const f32_cpx v = c->T[(size_t)(k0 + t) * n2 + j2]; re[t] = v.re; im[t] = v.im; }
      col[j2].re = re; col[j2].im = im;
    }
    v8_rec(fs->p2, 0, n2, col, 1, res, -1);
    for (int k2 = 0; k2 < n2; k2++) {
      f32_cpx *dst = c->out + (size_t)k0 + (size_t)n1 * k2;
      const v8f yr = res[k2].re, yi = res[k2].im;
      for (int t = 0; t < tb; t++) { dst[t].re = yr[t]; dst[t].im = yi[t]; }
    }
  }
  free(col);
}
/* forward transform of h complex points, out-of-place */
static void four_step_forward(const struct four_step *fs, const f32_cpx *in, f32_cpx *out, int threads) {
  struct fs_ctx c = {fs, in, NULL, out, v8_usable(fs->p1) && v8_usable(fs->p2)};
  c.T = (f32_cpx *)malloc(sizeof(f32_cpx) * (size_t)fs->h);
  par_for(threads, (fs->n2 + FS_TB - 1) / FS_TB, c.v8 ? fs_cols_v8 : fs_cols_scalar, &c);
  par_for(threads, (fs->n1 + FS_TB - 1) / FS_TB, c.v8 ? fs_rows_v8 : fs_rows_scalar, &c);
  free(c.T);
}

odft_plan *odft_create(int n, int precision) {
  if (n < 1) return NULL;
  odft_plan *p = (odft_plan *)calloc(1, sizeof *p);
  if (!p) return NULL;
  p->n = n;
  p->precision = precision;
  return p;
}

void odft_destroy(odft_plan *p) {
  if (!p) return;
  d64_plan_destroy(p->full64); d64_plan_destroy(p->half64);
  f32_plan_destroy(p->full32); f32_plan_destroy(p->half32);
  four_step_destroy(p->fs);
  free(p->rtw);
  free(p);
}

int odft_length(const odft_plan *p) { return p ? p->n : 0; }
void odft_set_threads(odft_plan *p, int threads) { if (p) p->threads = threads < 1 ? 1 : threads; }

/* Plans are built lazily; callers (the FFTW shim) create them under the
   reference's own planning mutex (src/filter.c:50-51), and the bench/test
   drivers touch each plan once from one thread before going parallel. */
static d64_plan *need_full64(odft_plan *p) { if (!p->full64) p->full64 = d64_plan_create(p->n); return p->full64; }
static d64_plan *need_half64(odft_plan *p) { if (!p->half64) p->half64 = d64_plan_create(p->n / 2); return p->half64; }
static f32_plan *need_full32(odft_plan *p) { if (!p->full32) p->full32 = f32_plan_create(p->n); return p->full32; }
static f32_plan *need_half32(odft_plan *p) { if (!p->half32) p->half32 = f32_plan_create(p->n / 2); return p->half32; }

static const double *need_rtw(odft_plan *p) {
  if (p->rtw) return p->rtw;
  int h = p->n / 2;
  double *t = (double *)malloc(sizeof(double) * 2 * (size_t)(h + 1));
  for (int k = 0; k <= h; k++) {
    double s, c;
    sincos(2.0 * M_PI * (double)k / (double)p->n, &s, &c);
    t[2 * k] = c; t[2 * k + 1] = -s;
  }
  p->rtw = t;
  return t;
}

void odft_warm(odft_plan *p, int real) {
  /* force table construction from a single thread */
  if (p->precision == ODFT_F64) { if (real && !(p->n & 1)) { need_half64(p); need_rtw(p); } else need_full64(p); }
  else {
    if (real && !(p->n & 1)) {
      if (!p->fs) p->fs = four_step_create(p->n / 2);
      if (!p->fs) need_half32(p);
      need_rtw(p);
    } else need_full32(p);
  }
}

void odft_c2c_f64(odft_plan *p, const double *in, double *out, int sign) {
  d64_plan *pl = need_full64(p);
  size_t n = (size_t)p->n;
  if (in == out) {
    d64_cpx *tmp = (d64_cpx *)malloc(sizeof(d64_cpx) * n);
    memcpy(tmp, in, sizeof(d64_cpx) * n);
    d64_execute(pl, tmp, (d64_cpx *)out, sign);
    free(tmp);
  } else {
    d64_execute(pl, (const d64_cpx *)in, (d64_cpx *)out, sign);
  }
}

void odft_c2c(odft_plan *p, const float *in, float *out, int sign) {
  size_t n = (size_t)p->n;
  if (p->precision == ODFT_F64) {
    d64_plan *pl = need_full64(p);
    d64_cpx *a = (d64_cpx *)malloc(sizeof(d64_cpx) * n * 2);
    d64_cpx *b = a + n;
    for (size_t i = 0; i < n; i++) { a[i].re = in[2 * i]; a[i].im = in[2 * i + 1]; }
    d64_execute(pl, a, b, sign);
    for (size_t i = 0; i < n; i++) { out[2 * i] = (float)b[i].re; out[2 * i + 1] = (float)b[i].im; }
    free(a);
  } else {
    f32_plan *pl = need_full32(p);
    if (in == out) {
      f32_cpx *tmp = (f32_cpx *)malloc(sizeof(f32_cpx) * n);
      memcpy(tmp, in, sizeof(f32_cpx) * n);
      f32_execute(pl, tmp, (f32_cpx *)out, sign);
      free(tmp);
    } else {
      f32_execute(pl, (const f32_cpx *)in, (f32_cpx *)out, sign);
    }
  }
}

/* Real-input transform of even length through the half-length complex
   transform of z[j] = x[2j] + i x[2j+1]:
     X[k] = (Z[k] + conj Z[h-k])/2  -  (i/2) W_n^k (Z[k] - conj Z[h-k]),  k = 0..h, Z[h] := Z[0]
   (the standard packing identity; odd n falls back to a full complex transform). */
void odft_r2c_f64(odft_plan *p, const double *in, double *out) {
  int n = p->n;
  if (n & 1) {
    d64_plan *pl = need_full64(p);
    d64_cpx *a = (d64_cpx *)malloc(sizeof(d64_cpx) * (size_t)n * 2);
    d64_cpx *b = a + n;
    for (int i = 0; i < n; i++) { a[i].re = in[i]; a[i].im = 0; }
    d64_execute(pl, a, b, -1);
    for (int k = 0; k <= n / 2; k++) { out[2 * k] = b[k].re; out[2 * k + 1] = b[k].im; }
    free(a);
    return;
  }
  int h = n / 2;
  d64_plan *pl = need_half64(p);
  const double *tw = need_rtw(p);
  d64_cpx *z = (d64_cpx *)malloc(sizeof(d64_cpx) * (size_t)h);
  d64_execute(pl, (const d64_cpx *)in, z, -1);
  for (int k = 0; k <= h; k++) {
    d64_cpx a = z[k == h ? 0 : k];
    d64_cpx b = z[k == 0 ? 0 : h - k];   /* conj applied below */
    double er = 0.5 * (a.re + b.re), ei = 0.5 * (a.im - b.im);   /* even part  */
    double orr = 0.5 * (a.re - b.re), oi = 0.5 * (a.im + b.im);  /* (Z - conj Z')/2 */
    /* -i * W * (orr + i oi) */
    double wr = tw[2 * k], wi = tw[2 * k + 1];
    double tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
    out[2 * k] = er + ti;
    out[2 * k + 1] = ei - tr;
  }
  free(z);
}

struct unpack_ctx { const f32_cpx *z; const double *tw; float *out; int h; };
static void unpack_range(void *vc, long lo, long hi) {
  const struct unpack_ctx *u = (const struct unpack_ctx *)vc;
  const f32_cpx *z = u->z; const double *tw = u->tw; float *out = u->out; const int h = u->h;
  for (long k = lo; k < hi; k++) {
    f32_cpx a = z[k == h ? 0 : k];
    f32_cpx b = z[k == 0 ? 0 : h - k];
    float er = 0.5f * (a.re + b.re), ei = 0.5f * (a.im - b.im);
    float orr = 0.5f * (a.re - b.re), oi = 0.5f * (a.im + b.im);
    float wr = (float)tw[2 * k], wi = (float)tw[2 * k + 1];
    float tr = wr * orr - wi * oi, ti = wr * oi + wi * orr;
    out[2 * k] = er + ti;
    out[2 * k + 1] = ei - tr;
  }
}
void odft_r2c(odft_plan *p, const float *in, float *out) {
  int n = p->n;
  int h = n / 2;
  if (p->precision == ODFT_F64) {
    double *a = (double *)malloc(sizeof(double) * ((size_t)n + 2 * (size_t)(h + 1)));
    double *b = a + n;
    for (int i = 0; i < n; i++) a[i] = in[i];
    odft_r2c_f64(p, a, b);
    for (int i = 0; i < 2 * (h + 1); i++) out[i] = (float)b[i];
    free(a);
    return;
  }
  if (n & 1) {
    f32_plan *pl = need_full32(p);
    f32_cpx *a = (f32_cpx *)malloc(sizeof(f32_cpx) * (size_t)n * 2);
    f32_cpx *b = a + n;
    for (int i = 0; i < n; i++) { a[i].re = in[i]; a[i].im = 0; }
    f32_execute(pl, a, b, -1);
    for (int k = 0; k <= h; k++) { out[2 * k] = b[k].re; out[2 * k + 1] = b[k].im; }
    free(a);
    return;
  }
  const double *tw = need_rtw(p);
  f32_cpx *z = (f32_cpx *)malloc(sizeof(f32_cpx) * (size_t)h);
  if (p->fs) four_step_forward(p->fs, (const f32_cpx *)in, z, p->threads);
  else f32_execute(need_half32(p), (const f32_cpx *)in, z, -1);
  struct unpack_ctx u = {z, tw, out, h};
  par_for(p->fs ? p->threads : 1, (long)h + 1, unpack_range, &u);
  free(z);
}

/* Hermitian-input backward transform: extend to the full spectrum and run the
   complex transform; only the small per-channel REAL-output case uses it
   (src/filter.c:387), so simplicity beats speed. */
void odft_c2r(odft_plan *p, const float *in, float *out) {
  int n = p->n;
  int h = n / 2;
  if (p->precision == ODFT_F64) {
    d64_plan *pl = need_full64(p);
    d64_cpx *a = (d64_cpx *)malloc(sizeof(d64_cpx) * (size_t)n * 2);
    d64_cpx *b = a + n;
    for (int k = 0; k <= h; k++) { a[k].re = in[2 * k]; a[k].im = in[2 * k + 1]; }
    a[0].im = 0;                                  /* c2r ignores the imaginary part of DC */
    if (!(n & 1)) a[h].im = 0;                    /* ... and of Nyquist */
    for (int k = h + 1; k < n; k++) { a[k].re = a[n - k].re; a[k].im = -a[n - k].im; }
    d64_execute(pl, a, b, +1);
    for (int i = 0; i < n; i++) out[i] = (float)b[i].re;
    free(a);
  } else {
    f32_plan *pl = need_full32(p);
    f32_cpx *a = (f32_cpx *)malloc(sizeof(f32_cpx) * (size_t)n * 2);
    f32_cpx *b = a + n;
    for (int k = 0; k <= h; k++) { a[k].re = in[2 * k]; a[k].im = in[2 * k + 1]; }
    a[0].im = 0;
    if (!(n & 1)) a[h].im = 0;
    for (int k = h + 1; k < n; k++) { a[k].re = a[n - k].re; a[k].im = -a[n - k].im; }
    f32_execute(pl, a, b, +1);
    for (int i = 0; i < n; i++) out[i] = b[i].re;
    free(a);
  }
}
