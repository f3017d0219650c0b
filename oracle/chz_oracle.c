/* oracle/chz_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 * Plain-C restatement of the reference algorithm; see chz_oracle.h for scope
 * and pinning.  Every function cites the reference lines it follows.
 */
#define _GNU_SOURCE 1
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <complex.h>
#include <stdint.h>
#include <pthread.h>
#include "chz_oracle.h"
#include "dft.h"

/* cabsf() as the reference's release build computes it: src/Makefile:105-109 compiles with -funsafe-math-optimizations
   -fcx-limited-range, under which gcc expands cabsf(z) inline to sqrtf(re*re + im*im) in float (two rounded products, a
   rounded sum, sqrtss) instead of calling libm's hypotf, which works in double.  The two differ in the last bit for about
   15 % of all arguments -- round 2's "+-1 LSB on < 2 % of the PCM samples" in the envelope modes was this restatement
   calling libm, not the device (bisected in round 3: every other mode was already bit-exact). */
static inline float chzo_cabsf(float re, float im) {
  volatile float a = re * re, b = im * im;      /* (volatile: no contraction into an fma, whatever flags this file is built with) */
  volatile float s = a + b;
  return sqrtf(s);
}

/* Plans are cached per (length, real?) so repeated blocks do not rebuild the
   twiddle tables; tables are built under the lock, execution is lock-free. */
#define PLAN_CACHE 16
static struct { int n, real; odft_plan *pl; } Cache[PLAN_CACHE];
static pthread_mutex_t Cache_lock = PTHREAD_MUTEX_INITIALIZER;
static odft_plan *cached_plan(int n, int real) {
  pthread_mutex_lock(&Cache_lock);
  int freeslot = -1;
  for (int i = 0; i < PLAN_CACHE; i++) {
    if (Cache[i].pl && Cache[i].n == n && Cache[i].real == real) {
      odft_plan *p = Cache[i].pl; pthread_mutex_unlock(&Cache_lock); return p;
    }
    if (!Cache[i].pl && freeslot < 0) freeslot = i;
  }
  if (freeslot < 0) { freeslot = 0; odft_destroy(Cache[0].pl); }
  odft_plan *p = odft_create(n, ODFT_F64);
  if (p) odft_warm(p, real);
  Cache[freeslot].n = n; Cache[freeslot].real = real; Cache[freeslot].pl = p;
  pthread_mutex_unlock(&Cache_lock);
  return p;
}

/* ------------------------------------------------------------------ */
/* small helpers                                                       */
/* ------------------------------------------------------------------ */

/* sin(pi x), cos(pi x) with the argument reduced in units of half-turns so the
   result does not lose accuracy for large |x| (role of src/sincospi.c:24-65). */
static void sincos_pi(double x, double *s, double *c) {
  if (!isfinite(x)) { *s = *c = NAN; return; }
  double y = fmod(x, 2.0);
  if (y < 0) y += 2.0;                 /* [0,2) */
  int q = (int)floor(2.0 * y);         /* quarter-turn index 0..3 */
  if (q > 3) q = 3;
  double r = y - 0.5 * q;              /* [0,0.5) */
  double sr, cr;
  if (r > 0.25) { sr = cos(M_PI * (0.5 - r)); cr = sin(M_PI * (0.5 - r)); }
  else          { sr = sin(M_PI * r);         cr = cos(M_PI * r); }
  switch (q) {
  case 0: *s =  sr; *c =  cr; break;
  case 1: *s =  cr; *c = -sr; break;
  case 2: *s = -sr; *c = -cr; break;
  default:*s = -cr; *c =  sr; break;
  }
}

static double sinc_pi(double x) {      /* src/misc.h:217-221 */
  return x == 0 ? 1.0 : sin(M_PI * x) / (M_PI * x);
}

/* ------------------------------------------------------------------ */
/* filter design                                                       */
/* ------------------------------------------------------------------ */

/* Modified Bessel function I0 by its power series, stopping when a term falls
   below 1e-12 of the running sum, at most 40 terms (src/misc.c:416-427). */
double chzo_i0(double z) {
  const double t = 0.25 * z * z;
  double term = t, sum = 1.0 + t;
  for (int k = 2; k < 40; k++) {
    term *= t / ((double)k * (double)k);
    sum += term;
    if (term < 1e-12 * sum) break;
  }
  return sum;
}

/* Kaiser window, float32 storage, computed symmetric from both ends; the middle
   tap of an odd-length window is exactly 1 (src/window.c:217-237). */
int chzo_make_kaiser(float *w, int M, double beta) {
  if (!w || M < 2 || !isfinite(beta)) return -1;
  const double inv = 1.0 / chzo_i0(beta);
  const double step = 2.0 / (M - 1);
  for (int n = 0; n < M / 2; n++) {
    double p = step * n - 1.0;
    float v = (float)(chzo_i0(beta * sqrt(1.0 - p * p)) * inv);
    w[n] = v; w[M - 1 - n] = v;
  }
  if (M & 1) w[(M - 1) / 2] = 1.0f;
  return 0;
}

/* scale so the taps sum to M (src/window.c:240-254) */
int chzo_normalize_window(float *w, int M) {
  if (!w || M == 0) return -1;
  double g = 0;
  for (int i = 0; i < M; i++) g += w[i];
  if (g == 0 || !isfinite(g)) return -1;
  g = M / g;
  for (int i = 0; i < M; i++) w[i] *= (float)g;
  return 0;
}

/* src/filter.c:968-1045.  Taps: h[i] = kaiser[i] * 2 bw2 sinc(2 bw2 n) * e^{j 2 pi center n},
   n = i - (M-1)/2, M = P - olen + 1; gain = (sqrt2 if master real) / (sum_i r_i * N_master);
   zero-pad to P, forward P-point transform. */
int chzo_set_filter(int P, int olen, int master_points, int master_real, int out_type,
                    double low, double high, double beta, float *response) {
  if (!response || isnan(low) || isnan(high) || isnan(beta)) return -1;
  if (out_type == CHZO_REAL) { low = fabs(low); high = fabs(high); }   /* :971-975 */
  if (low > high) { double t = low; low = high; high = t; }            /* :977-981 */
  low  = low  < -0.5 ? -0.5 : low  > 0.5 ? 0.5 : low;                  /* :983-984 */
  high = high < -0.5 ? -0.5 : high > 0.5 ? 0.5 : high;
  const int M = P - olen + 1;                                          /* :986-988 */
  if (M < 2) return -1;
  const double bw2 = (high == low) ? 0.0001 : fabs(high - low) / 2;    /* :992 */
  const double center = (high + low) / 2;                              /* :993 */

  float *win = (float *)malloc(sizeof(float) * (size_t)M);
  float *taps = (float *)calloc((size_t)P * 2, sizeof(float));
  if (!win || !taps) { free(win); free(taps); return -1; }
  chzo_make_kaiser(win, M, beta);
  chzo_normalize_window(win, M);

  double wsum = 0;
  for (int i = 0; i < M; i++) {                                        /* :1011-1019 */
    double n = i - (double)(M - 1) / 2;
    double r = win[i] * 2 * bw2 * sinc_pi(2 * bw2 * n);
    wsum += r;
    double s, c;
    sincos_pi(2 * center * n, &s, &c);
    taps[2 * i] = (float)(c * r);
    taps[2 * i + 1] = (float)(s * r);
  }
  const double gain = (master_real ? M_SQRT2 : 1.0) / (wsum * master_points);  /* :1024-1025 */
  for (int i = 0; i < M; i++) {                                        /* :1027-1028 */
    taps[2 * i] = (float)(taps[2 * i] * gain);
    taps[2 * i + 1] = (float)(taps[2 * i + 1] * gain);
  }
  odft_plan *pl = cached_plan(P, 0);
  if (!pl) { free(win); free(taps); return -1; }
  odft_c2c(pl, taps, response, -1);                                    /* :1030 */
  free(win); free(taps);
  return 0;
}

/* ------------------------------------------------------------------ */
/* forward transform                                                   */
/* ------------------------------------------------------------------ */

int chzo_forward_f64(const float *window, int N, int in_type, double *spectrum) {
  odft_plan *pl = cached_plan(N, in_type == CHZO_REAL);
  if (!pl) return -1;
  if (in_type == CHZO_REAL) {
    double *x = (double *)malloc(sizeof(double) * (size_t)N);
    for (int i = 0; i < N; i++) x[i] = window[i];
    odft_r2c_f64(pl, x, spectrum);                                     /* src/filter.c:508,582 */
    free(x);
  } else {
    double *x = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    for (long i = 0; i < 2L * N; i++) x[i] = window[i];
    odft_c2c_f64(pl, x, spectrum, -1);                                 /* src/filter.c:505,573 */
    free(x);
  }
  return 0;
}

int chzo_forward(const float *window, int N, int in_type, float *spectrum) {
  int bins = in_type == CHZO_REAL ? N / 2 + 1 : N;
  double *s = (double *)malloc(sizeof(double) * 2 * (size_t)bins);
  if (!s) return -1;
  int r = chzo_forward_f64(window, N, in_type, s);
  for (long i = 0; i < 2L * bins; i++) spectrum[i] = (float)s[i];
  free(s);
  return r;
}

/* src/filter.c:464-474: state += alpha (X[bin] - state); X[bin] -= state;
   list ends with the DC entry. */
void chzo_notch(double *state, const int *bins, int n, double alpha, float *spectrum) {
  for (int i = 0; i < n; i++) {
    int b = bins[i];
    state[2 * i]     += alpha * ((double)spectrum[2 * b]     - state[2 * i]);
    state[2 * i + 1] += alpha * ((double)spectrum[2 * b + 1] - state[2 * i + 1]);
    spectrum[2 * b]     = (float)((double)spectrum[2 * b]     - state[2 * i]);
    spectrum[2 * b + 1] = (float)((double)spectrum[2 * b + 1] - state[2 * i + 1]);
    if (b == 0) break;
  }
}

/* ------------------------------------------------------------------ */
/* per-channel gather                                                  */
/* ------------------------------------------------------------------ */

/* Source descriptor for one output bin: master index, or -1 for "zero";
   conj != 0 means the master bin is conjugated (inverted spectrum). */
struct src { int idx; int conj; };

/* Output bins are visited from the most negative frequency upward; the t-th
   visited bin lives at FFT-order index ((s_bins+1)/2 + t) mod s_bins
   (src/filter.c:730,818: "wp = (s_bins+1)/2"). */

/* REAL master, COMPLEX slave (src/filter.c:810-893) */
static struct src map_real_to_complex(int t, int m_bins, int s_bins, int shift) {
  struct src s = { -1, 0 };
  if (shift >= 0) {                       /* upright spectrum, :819-855 */
    long r = (long)shift - s_bins / 2 + t;
    if (r >= 0 && r < m_bins) s.idx = (int)r;
  } else {                                /* inverted spectrum, read downward, :856-892 */
    long r = -((long)shift - s_bins / 2) - t;
    if (r >= 0 && r < m_bins) { s.idx = (int)r; s.conj = 1; }
  }
  return s;
}

/* COMPLEX master, COMPLEX slave (src/filter.c:728-793, non-beam).  The walk is
   stateful: leading zeros while below -(m_bins+1)/2, one wrap of a negative
   start into the upper half of the array, then an upward copy that ends for
   good when the read index arrives at (m_bins+1)/2. */
static void map_complex_to_complex(struct src *map, int m_bins, int s_bins, int shift) {
  long r = (long)shift - s_bins / 2;
  int started = 0, ended = 0;
  long rp = 0;
  for (int t = 0; t < s_bins; t++, r++) {
    map[t].idx = -1; map[t].conj = 0;
    if (!started) {
      if (r < -(long)((m_bins + 1) / 2)) continue;        /* :733-741 */
      started = 1;
      rp = r < 0 ? r + m_bins : r;                         /* :742-743 */
      if (rp < 0 || rp >= m_bins) ended = 1;               /* :744-754 */
    }
    if (ended) continue;
    map[t].idx = (int)rp;                                  /* :780 */
    if (++rp == m_bins) rp = 0;                            /* :781-782 */
    if (rp == (m_bins + 1) / 2) ended = 1;                 /* :785 */
  }
}

static inline int imod(long x, int m) { long r = x % m; return (int)(r < 0 ? r + m : r); }

/* fd (float32, the reference's arithmetic) and/or fd64 (from a float64 spectrum) */
static int gather_core(const float *sp, const double *sp64, int m_bins, int in_type,
                       int s_bins, int out_type, int shift, int isb,
                       const float *H, float *fd, double *fd64) {
  if (m_bins <= 0 || s_bins <= 0 || !H) return -1;
#define X_RE(i) (sp ? (double)sp[2 * (i)] : sp64[2 * (i)])
#define X_IM(i) (sp ? (double)sp[2 * (i)+1] : sp64[2 * (i)+1])
#define PUT(w, re, im) do { if (fd) { fd[2*(w)] = (float)(re); fd[2*(w)+1] = (float)(im); } \
                            if (fd64) { fd64[2*(w)] = (re); fd64[2*(w)+1] = (im); } } while (0)
  /* float32 products when reproducing the reference, float64 otherwise */
#define MUL_PUT(w, xr, xi) do { \
    if (fd) { float a = (float)(xr), b = (float)(xi), hr = H[2*(w)], hi = H[2*(w)+1]; \
              fd[2*(w)] = a*hr - b*hi; fd[2*(w)+1] = a*hi + b*hr; } \
    if (fd64) { double hr = H[2*(w)], hi = H[2*(w)+1]; \
              fd64[2*(w)] = (xr)*hr - (xi)*hi; fd64[2*(w)+1] = (xr)*hi + (xi)*hr; } } while (0)

  if (out_type == CHZO_COMPLEX) {
    struct src *map = (struct src *)malloc(sizeof(struct src) * (size_t)s_bins);
    if (!map) return -1;
    if (in_type == CHZO_COMPLEX) map_complex_to_complex(map, m_bins, s_bins, shift);
    else for (int t = 0; t < s_bins; t++) map[t] = map_real_to_complex(t, m_bins, s_bins, shift);
    for (int t = 0; t < s_bins; t++) {
      int w = ((s_bins + 1) / 2 + t) % s_bins;
      if (map[t].idx < 0) { PUT(w, 0.0, 0.0); continue; }
      double xr = X_RE(map[t].idx), xi = X_IM(map[t].idx);
      if (map[t].conj) xi = -xi;
      MUL_PUT(w, xr, xi);
    }
    free(map);
  } else if (out_type == CHZO_REAL && in_type == CHZO_REAL) {
    for (int si = 0; si < s_bins; si++) {                  /* src/filter.c:803-809 */
      long mi = (long)si + shift;
      if (mi >= 0 && mi < m_bins) { double xr = X_RE(mi), xi = X_IM(mi); MUL_PUT(si, xr, xi); }
      else PUT(si, 0.0, 0.0);
    }
  } else if (out_type == CHZO_REAL && in_type == CHZO_COMPLEX) {
    for (int si = 0; si < s_bins; si++) {                  /* src/filter.c:794-802 */
      long mi = (long)si + shift;
      if (mi >= -(m_bins / 2) && mi < m_bins / 2) {
        int a = imod(mi, m_bins), b = imod((long)m_bins - mi, m_bins);
        double xr = X_RE(a) + X_RE(b), xi = X_IM(a) - X_IM(b);
        if (fd) { /* reference sums in float32 first, then multiplies */
          float fr = (float)X_RE(a) + (float)X_RE(b), fi = (float)X_IM(a) - (float)X_IM(b);
          float hr = H[2*si], hi = H[2*si+1];
          fd[2*si] = hr*fr - hi*fi; fd[2*si+1] = hr*fi + hi*fr;
        }
        if (fd64) { double hr = H[2*si], hi = H[2*si+1];
          fd64[2*si] = hr*xr - hi*xi; fd64[2*si+1] = hr*xi + hi*xr; }
      } else PUT(si, 0.0, 0.0);
    }
  } else return -1;

  if (isb && out_type == CHZO_COMPLEX) {                   /* src/filter.c:895-909 */
    for (int p = 1, dn = s_bins - 1; p < s_bins / 2; p++, dn--) {
      if (fd) {
        float pr = fd[2*p], pi = fd[2*p+1], nr = fd[2*dn], ni = fd[2*dn+1];
        fd[2*p] = pr + nr;  fd[2*p+1] = pi - ni;           /* pos + conj(neg) */
        fd[2*dn] = nr - pr; fd[2*dn+1] = ni + pi;          /* neg - conj(pos) */
      }
      if (fd64) {
        double pr = fd64[2*p], pi = fd64[2*p+1], nr = fd64[2*dn], ni = fd64[2*dn+1];
        fd64[2*p] = pr + nr;  fd64[2*p+1] = pi - ni;
        fd64[2*dn] = nr - pr; fd64[2*dn+1] = ni + pi;
      }
    }
    PUT(0, 0.0, 0.0);
  }
  PUT((s_bins + 1) / 2, 0.0, 0.0);                         /* src/filter.c:911 */
  return 0;
#undef X_RE
#undef X_IM
#undef PUT
#undef MUL_PUT
}

/* Beam variant of COMPLEX master -> COMPLEX slave (slave->beam, src/filter.c:756-775): two antennas ride on I and Q of
   the master; alpha and beta (set_filter_weights, :922-929) select or combine them,
       Y = (alpha X[rp] + beta conj(X[m_bins - rp])) H      (double complex arithmetic, rounded to float complex)
       Y = (Re X[rp] alpha + Im X[rp] beta) H               at rp = 0 and rp = m_bins/2.
   Same index walk as the plain variant.  The reference has no trailing zero fill here, so bins past the end of the
   walk keep whatever the buffer held (it is never initialised); this restatement defines them as zero. */
int chzo_gather_beam(const float *spectrum, int m_bins, int s_bins, int shift, const float *H,
                     double ar, double ai, double br, double bi, float *fd) {
  if (m_bins <= 0 || s_bins <= 0 || !H) return -1;
  struct src *map = (struct src *)malloc(sizeof(struct src) * (size_t)s_bins);
  if (!map) return -1;
  map_complex_to_complex(map, m_bins, s_bins, shift);
  for (int t = 0; t < s_bins; t++) {
    int w = ((s_bins + 1) / 2 + t) % s_bins;
    if (map[t].idx < 0) { fd[2 * w] = 0; fd[2 * w + 1] = 0; continue; }
    const int rp = map[t].idx;
    const double xr = spectrum[2 * rp], xi = spectrum[2 * rp + 1];
    const double hr = H[2 * w], hi = H[2 * w + 1];
    double sr, si;
    if (rp == 0 || rp == m_bins / 2) {                     /* :766-768 */
      /* xr*alpha*H + xi*beta*H, evaluated left to right as the C expression is */
      double t1r = xr * ar, t1i = xr * ai, t2r = xi * br, t2i = xi * bi;
      sr = (t1r * hr - t1i * hi) + (t2r * hr - t2i * hi);
      si = (t1r * hi + t1i * hr) + (t2r * hi + t2i * hr);
    } else {                                               /* :770-771 */
      const int mp = m_bins - rp;
      const double yr = spectrum[2 * mp], yi = -(double)spectrum[2 * mp + 1];   /* conjf */
      const double cr = (ar * xr - ai * xi) + (br * yr - bi * yi);
      const double ci = (ar * xi + ai * xr) + (br * yi + bi * yr);
      sr = cr * hr - ci * hi; si = cr * hi + ci * hr;
    }
    fd[2 * w] = (float)sr; fd[2 * w + 1] = (float)si;
  }
  free(map);
  fd[2 * ((s_bins + 1) / 2)] = 0; fd[2 * ((s_bins + 1) / 2) + 1] = 0;       /* :911 */
  return 0;
}

int chzo_channel_beam(const float *spectrum, int m_bins, int P, int olen, int shift, const float *response,
                      double ar, double ai, double br, double bi, float *out) {
  float *fd = (float *)malloc(sizeof(float) * 2 * (size_t)P);
  float *td = (float *)malloc(sizeof(float) * 2 * (size_t)P);
  if (!fd || !td) { free(fd); free(td); return -1; }
  int r = chzo_gather_beam(spectrum, m_bins, P, shift, response, ar, ai, br, bi, fd);
  if (r == 0) {
    odft_c2c(cached_plan(P, 0), fd, td, +1);
    memcpy(out, td + 2 * (size_t)(P - olen), sizeof(float) * 2 * (size_t)olen);
  }
  free(fd); free(td);
  return r;
}

int chzo_gather(const float *spectrum, int m_bins, int in_type, int s_bins, int out_type,
                int shift, int isb, const float *response, float *fdomain) {
  return gather_core(spectrum, NULL, m_bins, in_type, s_bins, out_type, shift, isb, response, fdomain, NULL);
}

int chzo_channel(const float *spectrum, int m_bins, int in_type, int P, int olen, int out_type,
                 int shift, int isb, const float *response, float *out) {
  int s_bins = out_type == CHZO_REAL ? P / 2 + 1 : P;      /* src/filter.c:346,374 */
  float *fd = (float *)malloc(sizeof(float) * 2 * (size_t)(s_bins + 1));
  float *td = (float *)malloc(sizeof(float) * 2 * (size_t)P);
  if (!fd || !td) { free(fd); free(td); return -1; }
  int r = gather_core(spectrum, NULL, m_bins, in_type, s_bins, out_type, shift, isb, response, fd, NULL);
  if (r == 0) {
    odft_plan *pl = cached_plan(P, 0);
    if (out_type == CHZO_COMPLEX) {
      odft_c2c(pl, fd, td, +1);                            /* src/filter.c:914 via :359 */
      memcpy(out, td + 2 * (size_t)(P - olen), sizeof(float) * 2 * (size_t)olen);   /* :357 */
    } else {
      odft_c2r(pl, fd, td);                                /* :387 */
      memcpy(out, td + (P - olen), sizeof(float) * (size_t)olen);                  /* :385 */
    }
  }
  free(fd); free(td);
  return r;
}

int chzo_channel_f64(const double *spectrum, int m_bins, int in_type, int P, int olen, int out_type,
                     int shift, int isb, const float *response, double *out) {
  if (out_type != CHZO_COMPLEX) return -1;                 /* demodulators only use COMPLEX out */
  double *fd = (double *)malloc(sizeof(double) * 2 * (size_t)P);
  double *td = (double *)malloc(sizeof(double) * 2 * (size_t)P);
  if (!fd || !td) { free(fd); free(td); return -1; }
  int r = gather_core(NULL, spectrum, m_bins, in_type, P, out_type, shift, isb, response, NULL, fd);
  if (r == 0) {
    odft_plan *pl = cached_plan(P, 0);
    odft_c2c_f64(pl, fd, td, +1);
    memcpy(out, td + 2 * (size_t)(P - olen), sizeof(double) * 2 * (size_t)olen);
  }
  free(fd); free(td);
  return r;
}

/* ------------------------------------------------------------------ */
/* sig_gen stream                                                      */
/* ------------------------------------------------------------------ */

struct chzo_siggen {
  /* rotator, src/osc.c:28-70 */
  double ph_re, ph_im, st_re, st_im;
  int steps;
  /* xoshiro256**, src/gauss.c:19-61 */
  uint64_t s[4];
  double amplitude, noise, scale;
  int isreal;
};

static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static uint64_t splitmix(uint64_t *x) {                    /* src/gauss.c:25-30 */
  uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

static uint64_t xo_next(uint64_t s[4]) {                   /* src/gauss.c:47-61 */
  uint64_t out = rotl(s[1] * 5, 7) * 9;
  uint64_t t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl(s[3], 45);
  return out;
}

/* popcount-based near-Gaussian, unit variance (src/gauss.c:102-111) */
static double gauss1(uint64_t s[4]) {
  uint64_t u = xo_next(s);
  double x = (double)(__builtin_popcountll(u * 0x2c1b3c6dULL) + __builtin_popcountll(u * 0x297a2d39ULL) - 64);
  x += (double)(int64_t)u * (1.0 / 9223372036854775808.0);
  return x * 0.1765469659009499;
}

chzo_siggen *chzo_siggen_create(double cycles_per_sample, double amplitude, double noise,
                                double scale, int isreal, uint64_t seed) {
  chzo_siggen *g = (chzo_siggen *)calloc(1, sizeof *g);
  if (!g) return NULL;
  g->ph_re = 1; g->ph_im = 0; g->steps = 16384;            /* src/osc.c:18,30-32 */
  double s, c;
  if (cycles_per_sample != 0) { sincos_pi(2 * cycles_per_sample, &s, &c); g->st_re = c; g->st_im = s; }
  else { g->st_re = 1; g->st_im = 0; }                      /* src/osc.c:37-40 */
  uint64_t x = seed;                                        /* src/gauss.c:33-45 */
  for (int i = 0; i < 4; i++) g->s[i] = splitmix(&x);
  if ((g->s[0] | g->s[1] | g->s[2] | g->s[3]) == 0) g->s[0] = 1;
  g->amplitude = amplitude; g->noise = noise; g->scale = scale; g->isreal = isreal;
  return g;
}
void chzo_siggen_delete(chzo_siggen *g) { free(g); }

/* one rotator step; returns the phasor BEFORE stepping (src/osc.c:60-70),
   renormalising every 16384 steps (src/osc.c:47-57) */
static void osc_step(chzo_siggen *g, double *re, double *im) {
  if (--g->steps <= 0) {
    g->steps = 16384;
    double a = hypot(g->ph_re, g->ph_im);
    g->ph_re /= a; g->ph_im /= a;
  }
  *re = g->ph_re; *im = g->ph_im;
  double nr = g->ph_re * g->st_re - g->ph_im * g->st_im;
  double ni = g->ph_re * g->st_im + g->ph_im * g->st_re;
  g->ph_re = nr; g->ph_im = ni;
}

void chzo_siggen_generate(chzo_siggen *g, float *out, long n) {
  if (g->isreal) {                                          /* src/sig_gen.c:291-296 */
    for (long i = 0; i < n; i++) {
      double cr, ci; osc_step(g, &cr, &ci);
      double samp = g->amplitude * cr + g->noise * gauss1(g->s);
      out[i] = (float)(samp * g->scale);
    }
  } else {                                                  /* src/sig_gen.c:321-326 */
    for (long i = 0; i < n; i++) {
      double cr, ci; osc_step(g, &cr, &ci);
      double nr = gauss1(g->s), ni = gauss1(g->s);          /* src/misc.h:399-403 */
      double sr = g->amplitude * cr + g->noise * nr, si = g->amplitude * ci + g->noise * ni;
      out[2 * i] = (float)(sr * g->scale);
      out[2 * i + 1] = (float)(si * g->scale);
    }
  }
}

/* src/radio.c:1630-1650: 10^(-gain/20) * 2^(1-bits), real front ends get -3 dB of gain */
double chzo_scale_ad(double rf_gain_db, double rf_atten_db, double level_cal_db, int isreal, int bitspersample) {
  double g = 0;
  if (isfinite(rf_gain_db)) g += rf_gain_db;
  if (isfinite(rf_atten_db)) g -= rf_atten_db;
  if (isfinite(level_cal_db)) g -= level_cal_db;
  if (isreal) g -= 3.0;
  return ldexp(pow(10.0, -g / 20.0), 1 - bitspersample);
}

/* src/radio.c:1175-1199 */
int chzo_compute_tuning(int N, double samprate, double freq, int *shift, double *remainder) {
  double hzperbin = samprate / N;
  long r = lrint(freq / hzperbin);
  if (shift) *shift = (int)r;
  if (remainder) *remainder = fma(-(double)r, hzperbin, freq);
  return labs(r) >= N / 2 ? -1 : 0;
}

/* ------------------------------------------------------------------ */
/* estimate_noise() (src/radio.c:1719-1866)          */
/* ------------------------------------------------------------------ */

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* Noise density estimate around one channel from the master spectrum of the block it just consumed:
   energies of nbins = max(slave bins, 1000) master bins centred on |shift| (:1794-1820), the
   p = 0.10 quantile with linear interpolation (:1760-1775), the mean of the energies not above
   1.5 x that quantile (:1846-1857), times the truncated-exponential correction (:1840-1844), divided
   by master bins x front-end sample rate (:1863-1865).  The k-th smallest value does not depend on
   how it is found, so the quickselect of :1722-1757 is replaced by a sort.
   Complex masters (:1821-1837): the reference stops filling energies[] at the +Nyquist seam and
   then reads the uninitialised remainder; here the remainder does not exist (n = entries filled). */
double chzo_estimate_noise(const float *spectrum, int m_bins, int in_type, int s_bins, int shift, double samprate) {
  if (s_bins <= 0) return 0;
  int nbins = s_bins < 1000 ? 1000 : s_bins;
  if (nbins > m_bins) return NAN;              /* the reference would index outside fdomain */
  double *en = (double *)malloc(sizeof(double) * (size_t)nbins);
  int n = 0;
  if (in_type == CHZO_REAL) {
    int mbin = abs(shift) - nbins / 2;
    if (mbin < 0) mbin = 0;
    else if (mbin + nbins > m_bins) mbin = m_bins - nbins;
    for (int i = 0; i < nbins; i++, mbin++) {
      float re = spectrum[2 * mbin], im = spectrum[2 * mbin + 1];
      en[n++] = (double)(re * re + im * im);   /* cnrmf, float arithmetic */
    }
  } else {
    int mbin = shift - nbins / 2;
    if (mbin < 0) mbin += m_bins; else if (mbin >= m_bins) mbin -= m_bins;
    if (mbin < 0 || mbin >= m_bins) { free(en); return 0; }
    for (int i = 0; i < nbins; i++) {
      float re = spectrum[2 * mbin], im = spectrum[2 * mbin + 1];
      en[n++] = (double)(re * re + im * im);
      if (++mbin == m_bins) mbin = 0;
      if (mbin == m_bins / 2) break;
    }
  }
  double *sorted = (double *)malloc(sizeof(double) * (size_t)n);
  memcpy(sorted, en, sizeof(double) * (size_t)n);
  qsort(sorted, (size_t)n, sizeof(double), cmp_double);
  const double NQ = 0.10, N_cutoff = 1.5;      /* src/radio.c:73-74 */
  double pos = NQ * (n - 1);
  int i = (int)floor(pos);
  double frac = pos - i;
  double q = sorted[i];
  if (frac != 0.0) q = sorted[i] + frac * (sorted[i + 1] - sorted[i]);
  const double cut = N_cutoff * q;
  double energy = 0; int noisebins = 0;
  for (int k = 0; k < n; k++) if (en[k] <= cut) { energy += en[k]; noisebins++; }
  free(en); free(sorted);
  if (noisebins == 0) return 0;
  energy /= noisebins;
  const double z = N_cutoff * (-log(1 - NQ));
  const double correction = 1 / (1 - z * exp(-z) / (1 - exp(-z)));
  return energy * correction / ((double)m_bins * samprate);
}

/* ------------------------------------------------------------------ */
/* RX888 sample conversion (src/rx888.c:697-767)     */
/* ------------------------------------------------------------------ */

/* int16 A/D samples -> scaled float32, with the energy sum and the clip count rx_callback keeps
   (src/rx888.c:753-767 portable, :697-751 AVX2).  rx888.c itself cannot be compiled here (libusb.h),
   so this is a restatement only.  The de-randomiser follows the AVX2 routine, which is the one an
   x86-64 host runs and which does what its comment says -- "if lsb == 1, flip all other bits", 16-bit
   shifts (:711-716).  (The portable fallback's `(x << 15) >> 14` is evaluated in int and does
   something else; it is not followed.) */
int chzo_convert_i16(const int16_t *samples, int n, float scale, int randomize, float *out, uint64_t *energy) {
  int clips = 0;
  uint64_t e = 0;
  for (int i = 0; i < n; i++) {
    int16_t x = samples[i];
    if (randomize) {
      int16_t mask = (int16_t)((int16_t)((uint16_t)x << 15) >> 14);   /* 0xFFFE if bit 0 set, else 0 */
      x ^= mask;
    }
    e += (uint64_t)((int32_t)x * x);
    if (x > 32766 || x < -32766) clips++;
    out[i] = (float)x * scale;
  }
  if (energy) *energy += e;
  return clips;
}

/* ------------------------------------------------------------------ */
/* downconvert() tail: fine tuning, block phase correction, baseband power
   (src/radio.c:1476-1520, src/osc.c:28-70, src/modes.c:265-266) */
/* ------------------------------------------------------------------ */

struct chzo_downconv {
  /* struct osc fine (src/osc.h:12-19) */
  double freq, rate;
  double ph_re, ph_im, st_re, st_im, ss_re, ss_im;
  int steps;
  /* chan->filter.{bin_shift,remainder,phase_adjust} (src/radio.h:179-181) */
  int bin_shift;
  double remainder;
  double pa_re, pa_im;
};

chzo_downconv *chzo_downconv_create(void) {
  chzo_downconv *d = (chzo_downconv *)calloc(1, sizeof *d);    /* struct channel starts zeroed: osc not initialised */
  if (!d) return NULL;
  d->remainder = NAN;          /* src/modes.c:265 */
  d->bin_shift = -1000999;     /* src/modes.c:266 */
  return d;
}
void chzo_downconv_delete(chzo_downconv *d) { free(d); }

static int dc_phasor_init(double re, double im) {               /* src/osc.c:20-24 */
  if (isnan(re) || isnan(im) || re * re + im * im < 0.9) return 0;
  return 1;
}
static void dc_cmul(double *ar, double *ai, double br, double bi) {
  double r = *ar * br - *ai * bi, i = *ar * bi + *ai * br; *ar = r; *ai = i;
}
static void dc_set_osc(chzo_downconv *d, double f, double r) {  /* src/osc.c:28-47 */
  if (!dc_phasor_init(d->ph_re, d->ph_im)) {
    d->ph_re = 1; d->ph_im = 0; d->steps = 16384; d->freq = 0; d->rate = 0;
    d->st_re = 1; d->st_im = 0; d->ss_re = 1; d->ss_im = 0;
  }
  if (f != d->freq) { d->freq = f; sincos_pi(2 * f, &d->st_im, &d->st_re); }
  if (r != d->rate) { d->rate = r; sincos_pi(2 * r, &d->ss_im, &d->ss_re); }
}
static void dc_step_osc(chzo_downconv *d, double *re, double *im) {   /* src/osc.c:49-70 */
  if (--d->steps <= 0) {
    if (!dc_phasor_init(d->ph_re, d->ph_im)) { d->ph_re = 1; d->ph_im = 0; }
    d->steps = 16384;
    double a = hypot(d->ph_re, d->ph_im); d->ph_re /= a; d->ph_im /= a;
    if (d->rate != 0) { double b = hypot(d->st_re, d->st_im); d->st_re /= b; d->st_im /= b; }
  }
  *re = d->ph_re; *im = d->ph_im;
  if (d->rate != 0) dc_cmul(&d->st_re, &d->st_im, d->ss_re, d->ss_im);
  dc_cmul(&d->ph_re, &d->ph_im, d->st_re, d->st_im);
}

/* One block of one channel after execute_filter_output(): buf holds olen complex samples and is
   rotated in place; returns bb_power.  L, M are the MASTER's block and impulse lengths
   (V = 1 + L/(M-1), src/radio.c:1492). */
double chzo_downconv_block(chzo_downconv *d, int shift, double remainder, double out_samprate,
                           double doppler_rate, int L, int M, float *buf, int olen) {
  if (shift != d->bin_shift || isnan(d->remainder) || remainder != d->remainder) {      /* :1479-1483 */
    dc_set_osc(d, -remainder / out_samprate, doppler_rate / (out_samprate * out_samprate));
    d->remainder = remainder;
  }
  if (shift != d->bin_shift) {                                                           /* :1491-1496 */
    const int V = 1 + L / (M - 1);
    sincos_pi(2.0 * (shift % V) / (double)V, &d->pa_im, &d->pa_re);
    double kr, ki; sincos_pi((shift - d->bin_shift) / (-2.0 * (V - 1)), &ki, &kr);
    dc_cmul(&d->ph_re, &d->ph_im, kr, ki);
    d->bin_shift = shift;
  }
  dc_cmul(&d->ph_re, &d->ph_im, d->pa_re, d->pa_im);                                     /* :1497 */
  double energy = 0;
  for (int n = 0; n < olen; n++) {                                                       /* :1500-1501 */
    double wr, wi; dc_step_osc(d, &wr, &wi);
    /* float complex * double complex, product rounded to float complex on store */
    const double xr = buf[2 * n], xi = buf[2 * n + 1];
    const float yr = (float)(xr * wr - xi * wi), yi = (float)(xr * wi + xi * wr);
    buf[2 * n] = yr; buf[2 * n + 1] = yi;
    energy += (double)(yr * yr + yi * yi);                                               /* cnrmf, :1516-1519 */
  }
  return olen ? energy / olen : 0.0;
}

/* ------------------------------------------------------------------ */
/* the linear demodulator's per-block work            */
/* (src/linear.c:56-375) and PCM packing                                */
/* (src/import.h:88-118, called from send_output, src/audio.c:117-133)  */
/* ------------------------------------------------------------------ */

/* ---- the PLL of the coherent modes (src/osc.h:21-32, src/osc.c:75-205) ---- */
/* nco(): phase accumulator -> (sin, cos).  Two quadrant bits, 10 table bits, 20 fraction bits; the table holds
   sin(pi/2 * i/1024), i = 0..1024; the cosine is read from the mirrored index; the fraction is applied as a second-order
   Taylor step with the table's own sine/cosine as derivatives (src/osc.c:91-126). */
static double nco_table[1025];
static int nco_ready;
void chzo_nco(unsigned accum, double *s, double *c) {
  if (!nco_ready) { for (int i = 0; i <= 1024; i++) nco_table[i] = sin(M_PI * 0.5 * (double)i / 1024); nco_ready = 1; }
  const unsigned fract = accum & ((1u << 20) - 1);
  unsigned tab = (accum >> 20) & 1023u;
  unsigned quad = accum >> 30;
  tab = (quad & 1) ? 1024 - tab : tab;
  const double sine = (quad & 2) ? -nco_table[tab] : nco_table[tab];
  tab = 1024 - tab;
  quad++;
  const double cosine = (quad & 2) ? -nco_table[tab] : nco_table[tab];
  const double diff = 2 * M_PI * ldexp((double)fract, -32);
  const double cdiff = cosine * diff, sdiff = sine * diff;
  if (s) *s = sine + cdiff - 0.5 * sdiff * diff;
  if (c) *c = cosine - sdiff - 0.5 * cdiff * diff;
}
typedef struct o_pll { uint32_t vco_phase; int32_t vco_step; double bw, damping, lower, upper, u, phi, K1, K2; int32_t wraps; } o_pll;
static void pll_set_limits(o_pll *q, double lo, double hi) { if (lo > hi) { double t = lo; lo = hi; hi = t; } q->lower = lo; q->upper = hi; }   /* src/osc.c:139-148 */
static void pll_set_params(o_pll *q, double bw, double damping) {                      /* src/osc.c:152-167 */
  if (bw == 0 || (bw == q->bw && damping == q->damping)) return;
  const double denom = damping + 1.0 / (4.0 * damping);
  const double wn = 4.0 * M_PI * fabs(bw) / denom;
  q->bw = bw; q->damping = damping;
  const double theta = wn;
  const double D = 1.0 + 2.0 * damping * theta + theta * theta;
  q->K1 = 4.0 * damping * theta / D;
  q->K2 = 4.0 * theta * theta / D;
}
static void pll_init(o_pll *q) { memset(q, 0, sizeof *q); pll_set_limits(q, -0.5, +0.5); pll_set_params(q, 0.01, M_SQRT1_2); }   /* src/osc.c:130-136 */
static double pll_run(o_pll *q, double phase) {                                        /* src/osc.c:174-205 */
  double u_new = q->u + q->K2 * phase;
  double dphi = u_new + q->K1 * phase;
  if (dphi > q->upper) { dphi = q->upper; if (phase > 0) u_new = q->u; }
  else if (dphi < q->lower) { dphi = q->lower; if (phase < 0) u_new = q->u; }
  q->u = u_new;
  q->phi += dphi;
  if (q->phi > 1) { q->phi -= 1; q->wraps++; }
  else if (q->phi < -1) { q->phi += 1; q->wraps--; }
  q->vco_step = (int32_t)ldexp(dphi, +32);
  q->vco_phase += (uint32_t)q->vco_step;
  return q->u;
}

struct chzo_lindemod {
  chzo_lindemod_params p;
  /* demod_linear's locals and the chan_t fields it updates */
  double gain;             /* chan->output.gain */
  int hangcount;           /* chan->linear.hangcount */
  double am_dc;            /* carrier removal filter state (src/linear.c:42) */
  double n0;               /* chan->sig.n0, NaN until the first estimate (src/radio.c:1467-1473) */
  int squelch_state;       /* src/linear.c:46 */
  int squelch_open;        /* src/linear.c:47 */
  chzo_downconv shift;     /* only the oscillator part is used: chan->shift (src/linear.c:168-172) */
  o_pll pll;               /* chan->pll.pll */
  int pll_lock, pll_lock_count, pll_rotations;   /* chan->pll.lock, .lock_count, .rotations */
  double pll_snr, pll_cphase, foffset;           /* chan->pll.snr, .cphase, chan->sig.foffset */
};

chzo_lindemod *chzo_lindemod_create(const chzo_lindemod_params *p) {
  chzo_lindemod *d = (chzo_lindemod *)calloc(1, sizeof *d);
  if (!d) return NULL;
  d->p = *p;
  d->gain = p->gain;
  d->n0 = NAN;
  d->squelch_state = (!p->pll_enable && !p->snr_squelch) ? p->squelch_tail + 4 : 0;   /* src/linear.c:46 */
  pll_init(&d->pll);                                                        /* :41 */
  d->squelch_open = 1;                                                      /* :47 */
  return d;
}
void chzo_lindemod_delete(chzo_lindemod *d) { free(d); }
void chzo_lindemod_set_params(chzo_lindemod *d, const chzo_lindemod_params *p) { double g = d->gain; d->p = *p; d->p.gain = g; }

static int pcm_bytes_per_sample(int enc) {
  return (enc == CHZO_PCM_MULAW || enc == CHZO_PCM_ALAW) ? 1 : (enc == CHZO_PCM_S16BE || enc == CHZO_PCM_S16LE || enc == CHZO_PCM_F16LE || enc == CHZO_PCM_F16BE) ? 2 : 4;
}

/* export_f16_noswap / _swap (src/import.h:140-157): `float16_t temp_float = in[i]` -- the C conversion float -> _Float16, round to
   nearest even, subnormals kept, overflow to infinity.  Restated in integer arithmetic (the image's gcc 11 has no _Float16 on x86-64;
   pinned against the reference's own import.h built with clang, tests/test_oracle_vs_reference.py). */
unsigned short chzo_f32_to_f16(float x) {
  uint32_t b; memcpy(&b, &x, 4);
  const uint32_t sign = (b >> 16) & 0x8000u, mag = b & 0x7fffffffu;
  if (mag >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (mag > 0x7f800000u ? 0x200u | ((mag >> 13) & 0x3ffu) : 0u));
  if (mag >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);            /* >= 65520: rounds to infinity */
  if (mag < 0x33000001u) return (unsigned short)sign;                          /* <= 2^-25: rounds to zero */
  const int e = (int)(mag >> 23) - 127;
  const uint32_t m = (mag & 0x7fffffu) | 0x800000u;
  const int shift = e < -14 ? 13 + (-14 - e) : 13;
  const uint32_t keep = m >> shift, rest = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  const uint32_t r = keep + ((rest > half || (rest == half && (keep & 1u))) ? 1u : 0u);
  const uint32_t hb = e < -14 ? r : ((uint32_t)(e + 15) << 10) + (r - 0x400u);
  return (unsigned short)(sign | hb);
}

/* src/rtp.c:459-483: clamp to +-1, to 16 bits, sign/magnitude, clip at 32635, bias 132, segment = position of the leading one,
   4 mantissa bits below it, everything inverted */
unsigned char chzo_float_to_mulaw(float x) {
  if (x > 1) x = 1; else if (x < -1) x = -1;
  int32_t sample = (int32_t)lrintf(ldexpf(x, 15));
  const int sign = sample < 0;
  int32_t pcm = sign ? -sample : sample;
  if (pcm > 32635) pcm = 32635;
  pcm += 0x84;
  int exponent = (31 - __builtin_clz((uint32_t)pcm)) - 7;
  exponent = exponent < 0 ? 0 : exponent > 7 ? 7 : exponent;
  const int mantissa = (pcm >> (exponent + 3)) & 0x0F;
  return (unsigned char)~((unsigned char)((exponent << 4) | mantissa) | (sign << 7));
}
/* src/rtp.c:500-533: the same without the bias; segment 0 is linear (mantissa = bits 4..7); XOR 0x55 / 0xD5 */
unsigned char chzo_float_to_alaw(float x) {
  if (x > 1.0f) x = 1.0f; else if (x < -1.0f) x = -1.0f;
  int32_t sample = (int32_t)lrintf(ldexpf(x, 15));
  const int sign = sample < 0;
  int32_t pcm = sign ? -sample : sample;
  if (pcm > 32635) pcm = 32635;
  int exponent = 0;
  if (pcm >= 256) exponent = (31 - __builtin_clz((uint32_t)pcm)) - 7;
  exponent = exponent < 0 ? 0 : exponent > 7 ? 7 : exponent;
  const int mantissa = exponent == 0 ? (pcm >> 4) & 0x0F : (pcm >> (exponent + 3)) & 0x0F;
  unsigned char a = (unsigned char)((exponent << 4) | mantissa);
  a ^= (sign ? 0xD5 : 0x55);
  return a;
}

/* export_s16_swap / _noswap and export_f32_* (src/import.h:88-118,176-183) on a little-endian host */
static void pcm_pack(int enc, const float *in, int count, unsigned char *out) {
  for (int i = 0; i < count; i++) {
    if (enc == CHZO_PCM_MULAW) out[i] = chzo_float_to_mulaw(in[i]);
    else if (enc == CHZO_PCM_ALAW) out[i] = chzo_float_to_alaw(in[i]);
    else if (enc == CHZO_PCM_S16BE || enc == CHZO_PCM_S16LE) {
      float t = ldexpf(in[i], 15);
      t = t > 32767.0f ? 32767.0f : t < -32767.0f ? -32767.0f : t;
      int16_t v = (int16_t)lrintf(t);
      uint16_t u = (uint16_t)v;
      if (enc == CHZO_PCM_S16BE) u = (uint16_t)((u >> 8) | (u << 8));
      memcpy(out + 2 * i, &u, 2);
    } else if (enc == CHZO_PCM_F16LE || enc == CHZO_PCM_F16BE) {
      uint16_t u = chzo_f32_to_f16(in[i]);
      if (enc == CHZO_PCM_F16BE) u = (uint16_t)((u >> 8) | (u << 8));
      memcpy(out + 2 * i, &u, 2);
    } else {
      uint32_t u; memcpy(&u, &in[i], 4);
      if (enc == CHZO_PCM_F32BE) u = __builtin_bswap32(u);
      memcpy(out + 4 * i, &u, 4);
    }
  }
}

/* One block: buf = N complex samples as downconvert() leaves them (rotated in place by the shift oscillator),
   bb_power = chan->sig.bb_power, n0_est = this block's estimate_noise().  pcm receives N*channels samples in the
   channel's encoding when st->frame == CHZO_FRAME_DATA.  Returns 0. */
int chzo_lindemod_block(chzo_lindemod *d, float *buf, int N, double bb_power, double n0_est, double blocktime,
                        unsigned char *pcm, chzo_lindemod_status *st) {
  const chzo_lindemod_params *c = &d->p;
  const double samprate = c->samprate;
  /* src/radio.c:1466-1473 (Power_alpha = 0.10, :72) */
  if (isnan(d->n0)) d->n0 = n0_est;
  else { double diff = n0_est - d->n0; d->n0 += 0.10 * diff; }
  /* src/linear.c:76-153: the PLL of the coherent modes runs first, on the block as downconvert() left it */
  {
    const int isamprate = (int)samprate;                          /* `int const samprate` (:27) */
    const int lock_limit = (int)lrint(0.5 * isamprate);           /* DEFAULT_PLL_LOCKTIME (:6,:38-40) */
    if (c->pll_enable) {
      double bw = c->pll_loop_bw / isamprate;
      if (d->pll_lock) bw *= 0.1;
      pll_set_params(&d->pll, bw, M_SQRT1_2);                     /* DEFAULT_PLL_DAMPING (:5) */
      double signal = 0, noise = 0;
      for (int n = 0; n < N; n++) {
        double sn, cs; chzo_nco(d->pll.vco_phase, &sn, &cs);
        const double br = buf[2 * n], bi = buf[2 * n + 1];
        const double sr = br * cs + bi * sn, si = bi * cs - br * sn;          /* buffer[n] * conj(vco) */
        buf[2 * n] = (float)sr; buf[2 * n + 1] = (float)si;
        double phase;
        if (d->pll_lock) {
          if (!c->pll_square) { const double mag = sqrt(sr * sr + si * si); phase = (mag > 0) ? si / mag : 0; }
          else phase = sr * si / (sr * sr - si * si);
        } else {
          if (!c->pll_square) phase = atan2(si, sr);
          else phase = 0.5 * atan2(sr * si + si * sr, sr * sr - si * si);     /* carg(s*s) */
        }
        phase /= (2 * M_PI);
        d->foffset = isamprate * pll_run(&d->pll, phase);
        signal += sr * sr; noise += si * si;
      }
      d->pll_cphase = ldexp(2 * M_PI * d->pll.vco_phase, -32);
      d->pll_rotations = d->pll.wraps;
      if (noise != 0) { d->pll_snr = (signal / noise) - 1; if (d->pll_snr < 0) d->pll_snr = 0; }
      else d->pll_snr = NAN;
      if (d->pll_snr < c->squelch_close) {
        d->pll_lock_count -= N;
        if (d->pll_lock_count <= -lock_limit) { d->pll_lock_count = -lock_limit; d->pll_lock = 0; }
      } else if (d->pll_snr > c->squelch_open) {
        d->pll_lock_count += N;
        if (d->pll_lock_count >= lock_limit) {
          d->pll_lock_count = lock_limit;
          if (!d->pll_lock) { d->pll_lock = 1; d->pll_rotations = 0; }
        }
      }
    } else { d->pll_rotations = 0; d->pll_lock_count = -lock_limit; d->pll_lock = 0; }
  }
  /* src/linear.c:168-172: post-downconversion shift */
  dc_set_osc(&d->shift, c->shift / samprate, 0);
  if (d->shift.freq != 0) {
    for (int n = 0; n < N; n++) {
      double wr, wi; dc_step_osc(&d->shift, &wr, &wi);
      const double xr = buf[2 * n], xi = buf[2 * n + 1];
      buf[2 * n] = (float)(xr * wr - xi * wi); buf[2 * n + 1] = (float)(xr * wi + xi * wr);
    }
  }
  /* src/linear.c:177-234: AGC */
  double gain_change = 1;
  if (c->agc) {
    const double bw = c->bandwidth;
    const double bn = sqrt(bw * d->n0);
    const double ampl = sqrt(bb_power);
    double peak_level = 0;
    {
      int sps = (int)lrint(N * .002 / blocktime);
      sps = sps < 1 ? 1 : sps;
      int n = 0;
      while (n + sps < N) {
        double energy = 0;
        for (int i = 0; i < sps; i++) { const float re = buf[2 * n], im = buf[2 * n + 1]; n++; energy += (double)(re * re + im * im); }
        if (energy > peak_level) peak_level = energy;
      }
      peak_level = sqrt(peak_level / sps);
    }
    if (peak_level * d->gain > M_SQRT2 * c->headroom) {
      d->gain = M_SQRT2 * c->headroom / peak_level;
      gain_change = 1;
      d->hangcount = (int)lrint(0.08 * samprate);
    } else if (ampl * d->gain > c->headroom) {
      const double newgain = c->headroom / ampl;
      if (newgain > 0) gain_change = pow(newgain / d->gain, 1.0 / N);
      d->hangcount = (int)lrint(c->hangtime * samprate);
    } else if (bn * d->gain > c->threshold * c->headroom) {
      const double newgain = c->threshold * c->headroom / bn;
      if (newgain > 0) gain_change = pow(newgain / d->gain, 1.0 / N);
    } else if (d->hangcount > 0) {
      d->hangcount -= N;
    } else {
      gain_change = pow(c->recovery_rate, 1.0 / samprate);
    }
  }
  /* src/linear.c:236-311: final pass */
  double output_power = 0;
  float *samples = buf;                    /* real output overlays the complex input, index low to high */
  if (c->channels == 1) {
    if (c->env) {
      double gain = d->gain;
      for (int n = 0; n < N; n++) {
        double s = gain * M_SQRT1_2 * (double)chzo_cabsf(buf[2 * n], buf[2 * n + 1]);
        gain *= gain_change;
        output_power += s * s;
        if (c->dc_alpha != 0) { d->am_dc += c->dc_alpha * (s - d->am_dc); s -= d->am_dc; }
        samples[n] = (float)s;
      }
      d->gain = gain;
    } else {
      double gain = d->gain;
      for (int n = 0; n < N; n++) {
        const double s = gain * buf[2 * n];
        gain *= gain_change;
        output_power += s * s;
        samples[n] = (float)s;
      }
      d->gain = gain;
    }
  } else {
    if (c->env) {
      double gain = d->gain;
      for (int n = 0; n < N; n++) {
        double sr = gain * M_SQRT1_2 * (double)buf[2 * n];
        double si = gain * M_SQRT1_2 * (double)chzo_cabsf(buf[2 * n], buf[2 * n + 1]);
        gain *= gain_change;
        output_power += sr * sr + si * si;
        if (c->dc_alpha != 0) { d->am_dc += c->dc_alpha * (si - d->am_dc); si -= d->am_dc; }
        buf[2 * n] = (float)sr; buf[2 * n + 1] = (float)si;
      }
      d->gain = gain;
    } else {
      double gain = d->gain;
      for (int n = 0; n < N; n++) {
        const double sr = gain * buf[2 * n], si = gain * buf[2 * n + 1];
        gain *= gain_change;
        output_power += sr * sr + si * si;
        buf[2 * n] = (float)sr; buf[2 * n + 1] = (float)si;
      }
      d->gain = gain;
    }
  }
  output_power /= N;
  if (c->channels == 1) output_power *= 2;
  st->output_power = output_power;
  /* src/linear.c:313-352: squelch */
  double snr = INFINITY;
  if (c->snr_squelch) snr = (bb_power / (d->n0 * c->bandwidth)) - 1.0;
  else if (c->pll_enable) snr = d->pll_snr;                                 /* :317-318 */
  const int squelch_state_max = c->squelch_tail + 4;
  if (!(c->snr_squelch || c->pll_enable) || snr >= c->squelch_open) d->squelch_state = squelch_state_max;
  else if (d->squelch_state > 0 && snr < c->squelch_close) d->squelch_state--;
  st->gain = d->gain; st->n0 = d->n0; st->snr = snr; st->squelch_state = d->squelch_state; st->foffset = d->foffset; st->pdeviation = 0;
  st->pll_lock = d->pll_lock; st->pll_snr = d->pll_snr; st->pll_cphase = d->pll_cphase; st->pll_rotations = d->pll_rotations;
  st->tone_deviation = 0; st->tone_mute = 0;
  switch (d->squelch_state) {
  case 3: st->output_power = 0; /* fallthrough */
  case 2: case 1:
    st->frame = CHZO_FRAME_SILENCE; st->mute = 0;           /* send_output(chan, NULL, N, false) */
    return 0;
  case 0:
    st->output_power = 0; st->frame = CHZO_FRAME_SILENCE; st->mute = 1;   /* send_output(chan, NULL, N, true) */
    return 0;
  default: break;
  }
  if (c->snr_squelch || c->pll_enable) {
    if (snr < c->squelch_close) d->squelch_open = 0;
    else if (!d->squelch_open && snr > c->squelch_open) { d->squelch_open = 1; d->am_dc = 0; }
  } else d->squelch_open = 1;
  st->mute = (output_power == 0 || !d->squelch_open || !c->tuned);     /* src/linear.c:366 */
  st->frame = CHZO_FRAME_DATA;
  pcm_pack(c->encoding, samples, N * c->channels, pcm);                  /* send_output -> export_* (src/audio.c:117-133) */
  return 0;
}
int chzo_pcm_bytes(int encoding, int nsamples) { return pcm_bytes_per_sample(encoding) * nsamples; }
void chzo_pcm_pack(int encoding, const float *in, int count, unsigned char *out) { pcm_pack(encoding, in, count, out); }

/* ------------------------------------------------------------------ */
/* SURVEY 8f rank 4, second half: the FM demodulator's per-block work   */
/* (demod_fm, src/fm.c:19-345)                                          */
/* ------------------------------------------------------------------ */

/* src/misc.c:414-468 */
static double fm_i0(double z) { double t = 0.25 * z * z, sum = 1 + t, term = t;
  for (int k = 2; k < 40; k++) { term *= t / (k * k); sum += term; if (term < 1e-12 * sum) break; } return sum; }
static double fm_i1(double z) { double t = 0.25 * z * z, term = 0.5 * t, sum = 1 + term;
  for (int k = 2; k < 40; k++) { term *= t / (k * (k + 1)); sum += term; if (term < 1e-12 * sum) break; } return 0.5 * z * sum; }
static double fm_xi(double thetasq) {
  double t = (2 + thetasq) * fm_i0(0.25 * thetasq) + thetasq * fm_i1(0.25 * thetasq);
  t *= t;
  return 2 + thetasq - (0.125 * M_PI) * exp(-0.5 * thetasq) * t;
}
double chzo_fm_snr(double r) {
  if (r <= M_PI / (4 - M_PI)) return 0;
  if (r > 100) return r;
  double thetasq = r;
  for (int i = 0; i < 10; i++) {
    double o = thetasq;
    thetasq = fm_xi(thetasq) * (1 + r) - 2;
    if (fabs(thetasq - o) <= 0.01) break;
  }
  return thetasq;
}

struct chzo_fmdemod {
  chzo_lindemod_params p;
  double n0;                  /* chan->sig.n0 */
  double pm_re, pm_im;        /* phase_memory (src/fm.c:36) */
  double deemph_state;        /* :57 */
  int squelch_state;          /* :58 */
  double foffset;             /* chan->sig.foffset */
  double pdeviation;          /* chan->fm.pdeviation */
  o_pll pll; int pll_was_on;  /* chan->pll.pll, chan->pll.was_on (:176-184) */
  /* PL tone squelch (:46-61): Goertzel detector (src/iir.h:24-44, src/iir.c:32-47), integration counter, phase memory */
  double g_coeff, g_cfr, g_cfi, g_s0, g_s1;
  int pl_sample_count, tone_mute;
  double old_pl_phase, tone_deviation;
};
chzo_fmdemod *chzo_fmdemod_create(const chzo_lindemod_params *p) {
  chzo_fmdemod *d = (chzo_fmdemod *)calloc(1, sizeof *d);
  if (!d) return NULL;
  d->p = *p; d->n0 = NAN;
  if (!isfinite(d->p.squelch_open) || d->p.squelch_open == 0) d->p.squelch_open = 6.3;      /* :38-41 */
  if (!isfinite(d->p.squelch_close) || d->p.squelch_close == 0) d->p.squelch_close = 4;
  d->tone_mute = 1;                                                          /* :61 muted until the tone is detected */
  if (p->tone_freq != 0) {                                                   /* :50-53 init_goertzel(tone_freq / samprate) */
    const double f = p->tone_freq / (int)p->samprate;
    d->g_coeff = 2 * cos(2 * M_PI * f); d->g_cfr = cos(2 * M_PI * f); d->g_cfi = -sin(2 * M_PI * f);
  }
  return d;
}
void chzo_fmdemod_delete(chzo_fmdemod *d) { free(d); }

int chzo_fmdemod_block(chzo_fmdemod *d, const float *buf, int N, double bb_power, double n0_est, double blocktime,
                       unsigned char *pcm, chzo_lindemod_status *st) {
  const chzo_lindemod_params *c = &d->p;
  const double samprate = c->samprate, devmax = 5000.;                       /* src/fm.c:43 */
  if (isnan(d->n0)) d->n0 = n0_est;                                          /* src/radio.c:1466-1473 */
  else { double diff = n0_est - d->n0; d->n0 += 0.10 * diff; }
  const double alpha = -expm1(-blocktime / 1.0);                             /* :55 */
  double fmsnr;
  const double noise = d->n0 * c->bandwidth;                                 /* :101 */
  const double beta = 0.5;
  const double snr = noise == 0 ? INFINITY : (bb_power / noise) - 1.0;       /* :105 */
  if (c->snr_squelch || (d->squelch_state <= 0 && snr < c->squelch_close)) {
    fmsnr = snr;
  } else {                                                                   /* :110-129 */
    double avg_amp = 0;
    double *amplitudes = (double *)malloc(sizeof(double) * (size_t)N);
    for (int n = 0; n < N; n++) avg_amp += amplitudes[n] = chzo_cabsf(buf[2 * n], buf[2 * n + 1]);
    avg_amp /= N;
    double fm_variance = 0;
    for (int n = 0; n < N; n++) fm_variance += (amplitudes[n] - avg_amp) * (amplitudes[n] - avg_amp);
    free(amplitudes);
    const double s2 = chzo_fm_snr(avg_amp * avg_amp * (N - 1) / fm_variance);
    fmsnr = s2 > 0.0 ? s2 : 0.0;
  }
  st->snr = fmsnr; st->n0 = d->n0; st->gain = 0; st->foffset = d->foffset; st->pdeviation = d->pdeviation;
  const int smax = c->squelch_tail + 5;                                      /* :149 */
  if (fmsnr >= c->squelch_open) d->squelch_state = smax;
  else if (d->squelch_state > 0 && (fmsnr < c->squelch_close || d->squelch_state < smax)) d->squelch_state--;
  st->squelch_state = d->squelch_state;
  st->pll_lock = 0; st->pll_snr = 0; st->pll_cphase = 0; st->pll_rotations = 0;
  st->tone_deviation = d->tone_deviation; st->tone_mute = c->tone_freq != 0 ? d->tone_mute : 0;
  if (d->squelch_state <= 4) {                                               /* :157-173 */
    if (d->squelch_state == 4) { d->g_s0 = d->g_s1 = 0; }                    /* reset_goertzel */
    if (d->squelch_state >= 1) { d->pm_re = 0; d->pm_im = 0; d->pl_sample_count = 0; st->output_power = 0; }
    else st->output_power = 0;          /* closed: chan->output.power keeps its last value, which was 0 */
    st->frame = CHZO_FRAME_SILENCE; st->mute = d->squelch_state == 0;
    return 0;
  }
  float *baseband = (float *)malloc(sizeof(float) * (size_t)N);
  if (c->pll_enable) {                                                       /* :176-203 PLL demodulator */
    const int isamprate = (int)samprate;
    const double pdev = devmax / isamprate;
    if (!d->pll_was_on) {
      d->pll_was_on = 1;
      pll_init(&d->pll);
      pll_set_params(&d->pll, 500.0 / isamprate, M_SQRT1_2);
      pll_set_limits(&d->pll, -pdev, +pdev);
    }
    for (int n = 0; n < N; n++) {
      double sn, cs; chzo_nco(d->pll.vco_phase, &sn, &cs);
      const double br = buf[2 * n], bi = buf[2 * n + 1];
      const double sr = br * cs + bi * sn, si = bi * cs - br * sn;            /* buffer[n] * conj(vco) */
      double phase = M_1_PI * atan2(si, sr);
      if (c->threshold_extend != 0) {
        if (fabs(phase) > devmax / isamprate) phase = copysign(devmax / isamprate, phase);
        double pw = (double)(buf[2 * n] * buf[2 * n] + buf[2 * n + 1] * buf[2 * n + 1]);   /* cnrmf */
        if (pw > 0) { pw /= (pw + beta * noise); phase *= pw; }
        else phase = 0;
      }
      baseband[n] = (float)(2 * pll_run(&d->pll, phase));
      d->pm_re = br; d->pm_im = bi;
    }
  } else {                                                                   /* :204-231 straight carg demodulation */
    d->pll_was_on = 0;
    double p0 = d->pm_re * d->pm_re + d->pm_im * d->pm_im;                   /* cnrm(phase_memory) */
    if (p0 > 0) p0 /= (p0 + beta * noise);
    for (int n = 0; n < N; n++) {
      const double br = buf[2 * n], bi = buf[2 * n + 1];
      const double sr = br * d->pm_re + bi * d->pm_im, si = bi * d->pm_re - br * d->pm_im;    /* buffer[n] * conj(phase_memory) */
      double phase = M_1_PI * atan2(si, sr);
      if (c->threshold_extend != 0) {
        if (fabs(phase) > devmax / samprate) phase = copysign(devmax / samprate, phase);
        double p1 = (double)(buf[2 * n] * buf[2 * n] + buf[2 * n + 1] * buf[2 * n + 1]);    /* cnrmf */
        if (p1 > 0) p1 /= (p1 + beta * noise);
        phase *= p0 * p1;
        p0 = p1;
      }
      baseband[n] = (float)phase;
      d->pm_re = br; d->pm_im = bi;
    }
  }
  if (d->squelch_state == smax) {                                            /* :232-256 */
    double peak_pos = 0, peak_neg = 0, foff = 0;
    for (int n = 0; n < N; n++) {
      foff += baseband[n];
      if (baseband[n] > peak_pos) peak_pos = baseband[n];
      else if (baseband[n] < peak_neg) peak_neg = baseband[n];
    }
    foff *= samprate * 0.5 / N;
    d->foffset += alpha * (foff - d->foffset);
    peak_pos *= samprate * 0.5; peak_neg *= samprate * 0.5;
    peak_pos -= d->foffset; peak_neg -= d->foffset;
    d->pdeviation = peak_pos > -peak_neg ? peak_pos : -peak_neg;
  }
  if (c->deemph_rate != 0) {                                                 /* :258-263 PM: remove DC */
    const float dc = (float)(2 * d->foffset / samprate);
    for (int n = 0; n < N; n++) baseband[n] -= dc;
  }
  if (c->tone_freq != 0) {                                                   /* :264-311 PL / CTCSS tone squelch */
    const int isamprate = (int)samprate;
    const int pl_integrate_samples = (int)lrint(isamprate * 0.24);           /* :59 */
    for (int n = 0; n < N; n++) {
      { const double s0save = d->g_s0; d->g_s0 = (double)baseband[n] + d->g_coeff * d->g_s0 - d->g_s1; d->g_s1 = s0save; }   /* update_goertzel */
      /* the 300 Hz low-pass (applyIIR, :269-270) only feeds lpf_energy, which only the branch behind
         `chan->options & (1LL<1)` reads -- and 1LL<1 is 0: it has no observable effect and is not restated */
      d->pl_sample_count++;
      if (d->pl_sample_count >= pl_integrate_samples) {
        { const double s0save = d->g_s0; d->g_s0 = 0 + d->g_coeff * d->g_s0 - d->g_s1; d->g_s1 = s0save; }   /* output_goertzel: one zero sample */
        const double cre = d->g_s0 - d->g_cfr * d->g_s1, cim = -d->g_cfi * d->g_s1;
        const double g = sqrt(cre * cre + cim * cim) / d->pl_sample_count;
        d->tone_deviation = isamprate * g;
        const double ph = atan2(cim, cre) / (2 * M_PI);
        double iptr = 0;
        d->old_pl_phase += c->tone_freq * d->pl_sample_count / isamprate;
        double np = 2 * modf(ph - d->old_pl_phase, &iptr);
        d->old_pl_phase = ph;
        np = np < -1 ? np + 2 : np > 1 ? np - 2 : np;
        d->tone_mute = d->tone_deviation < 250 || fabs(np) > .10;
        d->g_s0 = d->g_s1 = 0;
        d->pl_sample_count = 0;
      }
    }
    st->tone_deviation = d->tone_deviation; st->tone_mute = d->tone_mute;
    if (d->tone_mute) {                                                      /* :305-309 */
      st->output_power = 0; st->frame = CHZO_FRAME_SILENCE; st->mute = 1;
      st->foffset = d->foffset; st->pdeviation = d->pdeviation;
      free(baseband);
      return 0;
    }
  }
  if (c->deemph_rate != 0) {                                                 /* :312-320 */
    for (int n = 0; n < N; n++) {
      d->deemph_state += c->deemph_rate * (c->deemph_gain * baseband[n] - d->deemph_state);
      baseband[n] = (float)d->deemph_state;
    }
  }
  const double gain = (2 * c->headroom * samprate) / c->bandwidth;           /* :325 */
  double output_energy = 0;
  for (int n = 0; n < N; n++) {
    const double s = gain * baseband[n];
    output_energy += s * s;
    baseband[n] = (float)s;
  }
  st->gain = gain; st->output_power = output_energy / N; st->foffset = d->foffset; st->pdeviation = d->pdeviation;
  st->frame = CHZO_FRAME_DATA; st->mute = 0;
  pcm_pack(c->encoding, baseband, N, pcm);
  free(baseband);
  return 0;
}

/* ------------------------------------------------------------------ */
/* overlap-save stream                                                 */
/* ------------------------------------------------------------------ */

struct chzo_stream {
  int L, M, N, in_type, bins, per;
  float *win;       /* N samples: M-1 of history followed by the L newest */
};

chzo_stream *chzo_stream_create(int L, int M, int in_type) {
  if (L < 1 || M < 1 || (in_type != CHZO_REAL && in_type != CHZO_COMPLEX)) return NULL;
  chzo_stream *s = (chzo_stream *)calloc(1, sizeof *s);
  if (!s) return NULL;
  s->L = L; s->M = M; s->N = L + M - 1; s->in_type = in_type;         /* src/filter.c:196 */
  s->per = in_type == CHZO_REAL ? 1 : 2;
  s->bins = in_type == CHZO_REAL ? s->N / 2 + 1 : s->N;               /* src/filter.c:197 */
  if (s->bins < 2) { free(s); return NULL; }                          /* src/filter.c:198-199 */
  s->win = (float *)calloc((size_t)s->N * s->per, sizeof(float));     /* ring starts zeroed, :242,257 */
  if (!s->win) { free(s); return NULL; }
  return s;
}
void chzo_stream_delete(chzo_stream *s) { if (s) { free(s->win); free(s); } }
int chzo_stream_bins(const chzo_stream *s) { return s->bins; }
int chzo_stream_points(const chzo_stream *s) { return s->N; }

static void stream_advance(chzo_stream *s, const float *samples) {
  /* the read pointer advances by L per block while the write pointer leads it
     by M-1 (src/filter.c:244,259,628-635): window k = stream[kL-(M-1), kL+L) */
  size_t keep = (size_t)(s->M - 1) * s->per, fresh = (size_t)s->L * s->per;
  memmove(s->win, s->win + fresh, sizeof(float) * keep);
  memcpy(s->win + keep, samples, sizeof(float) * fresh);
}
int chzo_stream_push(chzo_stream *s, const float *samples, float *spectrum) {
  stream_advance(s, samples);
  return chzo_forward(s->win, s->N, s->in_type, spectrum);
}
int chzo_stream_push_f64(chzo_stream *s, const float *samples, double *spectrum) {
  stream_advance(s, samples);
  return chzo_forward_f64(s->win, s->N, s->in_type, spectrum);
}
