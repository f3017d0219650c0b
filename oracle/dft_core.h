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
  /* per recursion level, contiguous combine twiddles: ltw[level][k*(p-1) + (r-1)] = W_{n_level}^{r k},
     so the combine loops stream through memory instead of striding through tw[] */
  DFT_NAME(cpx) *ltw[64];
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
  {
    long nn = n, stride = 1;
    for (int lv = 0; lv < p->nfac; lv++) {
      const int pr = p->fac[lv];
      const long m = nn / pr;
      p->ltw[lv] = (DFT_NAME(cpx) *)malloc(sizeof(DFT_NAME(cpx)) * (size_t)m * (size_t)(pr - 1) + 16);
      if (!p->ltw[lv]) return p;   /* rec() falls back to tw[] for a NULL level */
      for (long k = 0; k < m; k++)
        for (int r = 1; r < pr; r++) p->ltw[lv][k * (pr - 1) + (r - 1)] = p->tw[(stride * k * r) % n];
      nn = m; stride *= pr;
    }
  }
  return p;
}

static void DFT_NAME(plan_destroy)(DFT_NAME(plan) *p) {
  if (!p) return;
  for (int i = 0; i < 64; i++) free(p->ltw[i]);
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
  if (m == 1) { for (int r = 0; r < p; r++) out[r] = in[(long)r * is]; }   /* leaf: gather, then one butterfly */
  else
    for (int r = 0; r < p; r++)
      DFT_NAME(rec)(pl, level + 1, m, in + (long)r * is, is * p, out + (long)r * m, tws * p, sgn);

  const DFT_NAME(cpx) *tw = pl->tw;
  const long N = pl->n; (void)N;
  const DFT_REAL s = (DFT_REAL)sgn;   /* multiplies the imaginary part of forward twiddles */
  if (p == 2) {
    const DFT_NAME(cpx) *lt = pl->ltw[level];
    for (int k = 0; k < m; k++) {
      DFT_NAME(cpx) a = out[k], b = out[k + m];
      const DFT_NAME(cpx) w = lt ? lt[k] : tw[tws * k];
      DFT_REAL wr = w.re, wi = -s * w.im;
      DFT_REAL br, bi; DFT_CMUL(br, bi, b.re, b.im, wr, wi);
      out[k].re = a.re + br;     out[k].im = a.im + bi;
      out[k + m].re = a.re - br; out[k + m].im = a.im - bi;
    }
  } else if (p == 4) {
    const DFT_NAME(cpx) *lt = pl->ltw[level];
    for (int k = 0; k < m; k++) {
      DFT_NAME(cpx) a = out[k], b = out[k + m], c = out[k + 2 * m], d = out[k + 3 * m];
      long t1 = tws * k, t2 = 2 * t1, t3 = 3 * t1;   /* all < N */
      const DFT_NAME(cpx) w1 = lt ? lt[3 * k] : tw[t1], w2 = lt ? lt[3 * k + 1] : tw[t2], w3 = lt ? lt[3 * k + 2] : tw[t3];
      DFT_REAL br, bi, cr, ci, dr, di;
      DFT_CMUL(br, bi, b.re, b.im, w1.re, -s * w1.im);
      DFT_CMUL(cr, ci, c.re, c.im, w2.re, -s * w2.im);
      DFT_CMUL(dr, di, d.re, d.im, w3.re, -s * w3.im);
      DFT_REAL s0r = a.re + cr, s0i = a.im + ci, s1r = a.re - cr, s1i = a.im - ci;
      DFT_REAL s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
      /* forward: X1 = s1 - i s3 ; backward: X1 = s1 + i s3 */
      out[k].re = s0r + s2r;         out[k].im = s0i + s2i;
      out[k + 2 * m].re = s0r - s2r; out[k + 2 * m].im = s0i - s2i;
      /* -i*(x+iy) = y - ix ; +i*(x+iy) = -y + ix.  With s=-1 (fwd) use -i, s=+1 use +i. */
      out[k + m].re = s1r - s * s3i;      out[k + m].im = s1i + s * s3r;
      out[k + 3 * m].re = s1r + s * s3i;  out[k + 3 * m].im = s1i - s * s3r;
    }
  } else if (p == 3 && pl->ltw[level]) {
    const DFT_NAME(cpx) *lt = pl->ltw[level];
    const DFT_REAL h3 = (DFT_REAL)0.86602540378443864676 * s;   /* sgn * sin(2 pi/3) */
    for (int k = 0; k < m; k++) {
      DFT_NAME(cpx) a = out[k], b = out[k + m], c = out[k + 2 * m];
      const DFT_NAME(cpx) w1 = lt[2 * k], w2 = lt[2 * k + 1];
      DFT_REAL br, bi, cr, ci;
      DFT_CMUL(br, bi, b.re, b.im, w1.re, -s * w1.im);
      DFT_CMUL(cr, ci, c.re, c.im, w2.re, -s * w2.im);
      DFT_REAL tr = br + cr, ti = bi + ci, ur = (br - cr) * h3, ui = (bi - ci) * h3;
      DFT_REAL mr = a.re - (DFT_REAL)0.5 * tr, mi = a.im - (DFT_REAL)0.5 * ti;
      out[k].re = a.re + tr;          out[k].im = a.im + ti;
      /* X1 = m + i*sgn*sin*(b-c) : i*(ur + i ui) = -ui + i ur */
      out[k + m].re = mr - ui;        out[k + m].im = mi + ur;
      out[k + 2 * m].re = mr + ui;    out[k + 2 * m].im = mi - ur;
    }
  } else if (p == 5 && pl->ltw[level]) {
    const DFT_NAME(cpx) *lt = pl->ltw[level];
    const DFT_REAL c1 = (DFT_REAL)0.30901699437494742410, c2 = (DFT_REAL)-0.80901699437494742410;
    const DFT_REAL s1 = (DFT_REAL)0.95105651629515357212 * s, s2 = (DFT_REAL)0.58778525229247312917 * s;
    for (int k = 0; k < m; k++) {
      DFT_NAME(cpx) a = out[k];
      DFT_REAL xr[4], xi[4];
      for (int r = 0; r < 4; r++) {
        const DFT_NAME(cpx) w = lt[4 * k + r], v = out[k + (long)(r + 1) * m];
        DFT_CMUL(xr[r], xi[r], v.re, v.im, w.re, -s * w.im);
      }
      DFT_REAL p1r = xr[0] + xr[3], p1i = xi[0] + xi[3], q1r = xr[0] - xr[3], q1i = xi[0] - xi[3];
      DFT_REAL p2r = xr[1] + xr[2], p2i = xi[1] + xi[2], q2r = xr[1] - xr[2], q2i = xi[1] - xi[2];
      out[k].re = a.re + p1r + p2r; out[k].im = a.im + p1i + p2i;
      DFT_REAL m1r = a.re + c1 * p1r + c2 * p2r, m1i = a.im + c1 * p1i + c2 * p2i;
      DFT_REAL m2r = a.re + c2 * p1r + c1 * p2r, m2i = a.im + c2 * p1i + c1 * p2i;
      /* X1 = m1 + i*sgn*(sin1 q1 + sin2 q2), X2 = m2 + i*sgn*(sin2 q1 - sin1 q2) */
      DFT_REAL u1r = s1 * q1r + s2 * q2r, u1i = s1 * q1i + s2 * q2i;
      DFT_REAL u2r = s2 * q1r - s1 * q2r, u2i = s2 * q1i - s1 * q2i;
      out[k + m].re = m1r - u1i;            out[k + m].im = m1i + u1r;
      out[k + 4L * m].re = m1r + u1i;       out[k + 4L * m].im = m1i - u1r;
      out[k + 2L * m].re = m2r - u2i;       out[k + 2L * m].im = m2i + u2r;
      out[k + 3L * m].re = m2r + u2i;       out[k + 3L * m].im = m2i - u2r;
    }
  } else {
    /* generic radix p: p-point DFT of the twiddled column, O(p^2) */
    DFT_NAME(cpx) *tt = (DFT_NAME(cpx) *)alloca(sizeof(DFT_NAME(cpx)) * 2 * (size_t)p);
    DFT_NAME(cpx) *uu = tt + p;
    const long rs = N / p;  /* twiddle stride of the p-th roots of unity */
    const DFT_NAME(cpx) *lt = pl->ltw[level];
    for (int k = 0; k < m; k++) {
      tt[0] = out[k];
      for (int r = 1; r < p; r++) {
        const DFT_NAME(cpx) w = lt ? lt[(long)k * (p - 1) + (r - 1)] : tw[tws * k * r];   /* tws*k*r < N */
        DFT_CMUL(tt[r].re, tt[r].im, out[k + (long)r * m].re, out[k + (long)r * m].im, w.re, -s * w.im);
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
