/* oracle/dft_core.h -- TEST INFRASTRUCTURE (CPU oracle), not a product path.
 *
 * Textbook mixed-radix decimation-in-time DFT, instantiated twice (double and
 * float) by oracle/dft.c.  It restates the *published definition* that the
 * reference delegates to FFTW3 (third-party, un-vendored, version unpinned:
 * docs/FFTW3.md:137-140 mention 3.3.8 / 3.3.10; linked at src/Makefile:290):
 *
 *     forward  (sign = -1):  X[k] = sum_n x[n] exp(-2 pi i k n / N)
 *     backward (sign = +1):  x[n] = sum_k X[k] exp(+2 pi i k n / N)
 *
 * both unnormalised, exactly the contract filter.c relies on
 * (src/filter.c:1020-1028 folds all scaling into the response).
 *
 * The double instantiation is the parity oracle (rounded once to float32 on
 * output, so it is strictly tighter than FFTW-float); the float instantiation
 * exists only so bench.py's cpu_baseline leg has a float32 CPU FFT to time.
 *
 * Include with  DFT_REAL  (double|float) and DFT_NAME(x) defined.
 */

typedef struct { DFT_REAL re, im; } DFT_NAME(cpx);

typedef struct DFT_NAME(plan_s) {
  int n;             /* transform length */
  int nfac;          /* number of radix factors */
  int fac[64];       /* radix sequence, product == n */
  DFT_NAME(cpx) *tw; /* tw[k] = exp(-2 pi i k / n), k = 0..n-1 (forward sign) */
} DFT_NAME(plan);

static void DFT_NAME(factorize)(int n, int *fac, int *nfac) {
  int k = 0;
  /* radix 4 first (cheapest), then 2, 3, 5, then any remaining odd primes */
  while (n % 4 == 0) { fac[k++] = 4; n /= 4; }
  while (n % 2 == 0) { fac[k++] = 2; n /= 2; }
  for (int p = 3; (long)p * p <= n; p += 2)
    while (n % p == 0) { fac[k++] = p; n /= p; }
  if (n > 1) fac[k++] = n;
  *nfac = k;
}

static DFT_NAME(plan) *DFT_NAME(plan_create)(int n) {
  DFT_NAME(plan) *p = (DFT_NAME(plan) *)calloc(1, sizeof *p);
  if (!p) return NULL;
  p->n = n;
  DFT_NAME(factorize)(n, p->fac, &p->nfac);
  p->tw = (DFT_NAME(cpx) *)malloc(sizeof(DFT_NAME(cpx)) * (size_t)n);
  if (!p->tw) { free(p); return NULL; }
  /* Octant-reduced evaluation keeps every table entry correctly rounded to
     ~1 ulp of double irrespective of n. */
  for (int k = 0; k < n; k++) {
    /* angle = 2 pi k / n ; reduce k/n to an eighth of a turn */
    long k8 = (long)k * 8;
    int oct = (int)(k8 / n);          /* 0..7 */
    long rem = k8 - (long)oct * n;    /* in [0,n) : fraction rem/(8n) of a turn */
    double a = (2.0 * M_PI / 8.0) * ((double)rem / (double)n); /* [0, pi/4) */
    double c, s;
    if (oct & 1) { a = M_PI / 4.0 - a; s = cos(a); c = sin(a); }  /* mirror inside the quadrant */
    else         { c = cos(a); s = sin(a); }
    double cc, ss;
    switch (oct >> 1) {
    case 0: cc =  c; ss =  s; break;
    case 1: cc = -s; ss =  c; break;
    case 2: cc = -c; ss = -s; break;
    default: cc =  s; ss = -c; break;
    }
    p->tw[k].re = (DFT_REAL)cc;
    p->tw[k].im = (DFT_REAL)(-ss);   /* forward sign */
  }
  return p;
}

static void DFT_NAME(plan_destroy)(DFT_NAME(plan) *p) {
  if (!p) return;
  free(p->tw); free(p);
}

#define DFT_CMUL(dr, di, ar, ai, br, bi) do { DFT_REAL _r = (ar)*(br) - (ai)*(bi); \
  DFT_REAL _i = (ar)*(bi) + (ai)*(br); (dr) = _r; (di) = _i; } while (0)

/* Recursive DIT:  out[0..n) <- DFT of in[0], in[is], in[2 is], ...
   tws = stride into the master twiddle table ( = N_master / n ).
   sgn = -1 forward, +1 backward (conjugated twiddles). */
static void DFT_NAME(rec)(const DFT_NAME(plan) *pl, int level, int n,
                          const DFT_NAME(cpx) *in, long is,
                          DFT_NAME(cpx) *out, long tws, int sgn) {
  if (n == 1) { out[0] = in[0]; return; }
  const int p = pl->fac[level];
  const int m = n / p;
  for (int r = 0; r < p; r++)
    DFT_NAME(rec)(pl, level + 1, m, in + (long)r * is, is * p, out + (long)r * m, tws * p, sgn);

  const DFT_NAME(cpx) *tw = pl->tw;
  const long N = pl->n; (void)N;
  const DFT_REAL s = (DFT_REAL)sgn;   /* multiplies the imaginary part of forward twiddles */
  if (p == 2) {
    for (int k = 0; k < m; k++) {
      DFT_NAME(cpx) a = out[k], b = out[k + m];
      long ti = tws * k;            /* < N/2 */
      DFT_REAL wr = tw[ti].re, wi = -s * tw[ti].im;
      DFT_REAL br, bi; DFT_CMUL(br, bi, b.re, b.im, wr, wi);
      out[k].re = a.re + br;     out[k].im = a.im + bi;
      out[k + m].re = a.re - br; out[k + m].im = a.im - bi;
    }
  } else if (p == 4) {
    for (int k = 0; k < m; k++) {
      DFT_NAME(cpx) a = out[k], b = out[k + m], c = out[k + 2 * m], d = out[k + 3 * m];
      long t1 = tws * k, t2 = 2 * t1, t3 = 3 * t1;   /* all < N */
      DFT_REAL br, bi, cr, ci, dr, di;
      DFT_CMUL(br, bi, b.re, b.im, tw[t1].re, -s * tw[t1].im);
      DFT_CMUL(cr, ci, c.re, c.im, tw[t2].re, -s * tw[t2].im);
      DFT_CMUL(dr, di, d.re, d.im, tw[t3].re, -s * tw[t3].im);
      DFT_REAL s0r = a.re + cr, s0i = a.im + ci, s1r = a.re - cr, s1i = a.im - ci;
      DFT_REAL s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
      /* forward: X1 = s1 - i s3 ; backward: X1 = s1 + i s3 */
      out[k].re = s0r + s2r;         out[k].im = s0i + s2i;
      out[k + 2 * m].re = s0r - s2r; out[k + 2 * m].im = s0i - s2i;
      /* -i*(x+iy) = y - ix ; +i*(x+iy) = -y + ix.  With s=-1 (fwd) use -i, s=+1 use +i. */
      out[k + m].re = s1r - s * s3i;      out[k + m].im = s1i + s * s3r;
      out[k + 3 * m].re = s1r + s * s3i;  out[k + 3 * m].im = s1i - s * s3r;
    }
  } else {
    /* generic radix p: p-point DFT of the twiddled column, O(p^2) */
    DFT_NAME(cpx) *tt = (DFT_NAME(cpx) *)alloca(sizeof(DFT_NAME(cpx)) * 2 * (size_t)p);
    DFT_NAME(cpx) *uu = tt + p;
    const long rs = N / p;  /* twiddle stride of the p-th roots of unity */
    for (int k = 0; k < m; k++) {
      for (int r = 0; r < p; r++) {
        long ti = tws * k * r;     /* < N because tws*k < N/p */
        DFT_CMUL(tt[r].re, tt[r].im, out[k + (long)r * m].re, out[k + (long)r * m].im,
                 tw[ti].re, -s * tw[ti].im);
      }
      for (int q = 0; q < p; q++) {
        DFT_REAL ar = 0, ai = 0;
        for (int r = 0; r < p; r++) {
          long ti = ((long)q * r % p) * rs;
          DFT_REAL wr = tw[ti].re, wi = -s * tw[ti].im;
          ar += tt[r].re * wr - tt[r].im * wi;
          ai += tt[r].re * wi + tt[r].im * wr;
        }
        uu[q].re = ar; uu[q].im = ai;   /* cannot write in place yet */
      }
      for (int q = 0; q < p; q++) out[k + (long)q * m] = uu[q];
    }
  }
}

/* out-of-place complex transform, in != out */
static void DFT_NAME(execute)(const DFT_NAME(plan) *pl, const DFT_NAME(cpx) *in,
                              DFT_NAME(cpx) *out, int sgn) {
  DFT_NAME(rec)(pl, 0, pl->n, in, 1, out, 1, sgn);
}

#undef DFT_CMUL
