/* oracle/dft_batch.h -- TEST INFRASTRUCTURE (CPU oracle / CPU timing baseline), not a product path.
 *
 * Eight independent complex transforms of one length at a time, one per SIMD lane: the same textbook mixed-radix decimation-in-time
 * recursion as dft_core.h (same radix sequence, same per-level twiddle tables of the float32 plan), with every data value an 8-wide
 * vector of floats (GCC vector extensions; AVX2 + FMA code through the function-level target attribute, checked at run time).
 * Only the float32 TIMING path of the four-step transform in dft.c uses it (round 6: the cpu_baseline of bench.py was 1.9 x slower
 * than pocketfft on the same host, which understated what the reference does on a real FFT library); the float64 parity oracle
 * stays on the plain scalar recursion.
 */
typedef float v8f __attribute__((vector_size(32)));
typedef double v4d __attribute__((vector_size(32)));
typedef struct { v8f re, im; } v8c;
#define V8_TARGET __attribute__((target("avx2,fma")))
#define V8_CMUL(dr, di, ar, ai, br, bi) do { v8f _r = (ar) * (br) - (ai) * (bi); v8f _i = (ar) * (bi) + (ai) * (br); (dr) = _r; (di) = _i; } while (0)

V8_TARGET static void v8_rec(const f32_plan *pl, int level, int n, const v8c *in, long is, v8c *out, int sgn) {
  if (n == 1) { out[0] = in[0]; return; }
  const int p = pl->fac[level];
  const int m = n / p;
  if (m == 1) { for (int r = 0; r < p; r++) out[r] = in[(long)r * is]; }
  else
    for (int r = 0; r < p; r++) v8_rec(pl, level + 1, m, in + (long)r * is, is * p, out + (long)r * m, sgn);
  const f32_cpx *lt = pl->ltw[level];
  const float s = (float)sgn;
  if (p == 2) {
    for (int k = 0; k < m; k++) {
      const v8c a = out[k], b = out[k + m];
      const float wr = lt[k].re, wi = -s * lt[k].im;
      v8f br, bi; V8_CMUL(br, bi, b.re, b.im, wr, wi);
      out[k].re = a.re + br; out[k].im = a.im + bi;
      out[k + m].re = a.re - br; out[k + m].im = a.im - bi;
    }
  } else if (p == 4) {
    for (int k = 0; k < m; k++) {
      const v8c a = out[k], b = out[k + m], c = out[k + 2 * m], d = out[k + 3 * m];
      const f32_cpx w1 = lt[3 * k], w2 = lt[3 * k + 1], w3 = lt[3 * k + 2];
      v8f br, bi, cr, ci, dr, di;
      V8_CMUL(br, bi, b.re, b.im, w1.re, -s * w1.im);
      V8_CMUL(cr, ci, c.re, c.im, w2.re, -s * w2.im);
      V8_CMUL(dr, di, d.re, d.im, w3.re, -s * w3.im);
      const v8f s0r = a.re + cr, s0i = a.im + ci, s1r = a.re - cr, s1i = a.im - ci;
      const v8f s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
      out[k].re = s0r + s2r; out[k].im = s0i + s2i;
      out[k + 2 * m].re = s0r - s2r; out[k + 2 * m].im = s0i - s2i;
      out[k + m].re = s1r - s * s3i; out[k + m].im = s1i + s * s3r;
      out[k + 3 * m].re = s1r + s * s3i; out[k + 3 * m].im = s1i - s * s3r;
    }
  } else if (p == 3) {
    const float h3 = 0.86602540378443864676f * s;
    for (int k = 0; k < m; k++) {
      const v8c a = out[k], b = out[k + m], c = out[k + 2 * m];
      const f32_cpx w1 = lt[2 * k], w2 = lt[2 * k + 1];
      v8f br, bi, cr, ci;
      V8_CMUL(br, bi, b.re, b.im, w1.re, -s * w1.im);
      V8_CMUL(cr, ci, c.re, c.im, w2.re, -s * w2.im);
      const v8f tr = br + cr, ti = bi + ci, ur = (br - cr) * h3, ui = (bi - ci) * h3;
      const v8f mr = a.re - 0.5f * tr, mi = a.im - 0.5f * ti;
      out[k].re = a.re + tr; out[k].im = a.im + ti;
      out[k + m].re = mr - ui; out[k + m].im = mi + ur;
      out[k + 2 * m].re = mr + ui; out[k + 2 * m].im = mi - ur;
    }
  } else if (p == 5) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f * s, s2 = 0.58778525229247312917f * s;
    for (int k = 0; k < m; k++) {
      const v8c a = out[k];
      v8f xr[4], xi[4];
      for (int r = 0; r < 4; r++) {
        const f32_cpx w = lt[4 * k + r]; const v8c v = out[k + (long)(r + 1) * m];
        V8_CMUL(xr[r], xi[r], v.re, v.im, w.re, -s * w.im);
      }
      const v8f p1r = xr[0] + xr[3], p1i = xi[0] + xi[3], q1r = xr[0] - xr[3], q1i = xi[0] - xi[3];
      const v8f p2r = xr[1] + xr[2], p2i = xi[1] + xi[2], q2r = xr[1] - xr[2], q2i = xi[1] - xi[2];
      out[k].re = a.re + p1r + p2r; out[k].im = a.im + p1i + p2i;
      const v8f m1r = a.re + c1 * p1r + c2 * p2r, m1i = a.im + c1 * p1i + c2 * p2i;
      const v8f m2r = a.re + c2 * p1r + c1 * p2r, m2i = a.im + c2 * p1i + c1 * p2i;
      const v8f u1r = s1 * q1r + s2 * q2r, u1i = s1 * q1i + s2 * q2i;
      const v8f u2r = s2 * q1r - s1 * q2r, u2i = s2 * q1i - s1 * q2i;
      out[k + m].re = m1r - u1i; out[k + m].im = m1i + u1r;
      out[k + 4L * m].re = m1r + u1i; out[k + 4L * m].im = m1i - u1r;
      out[k + 2L * m].re = m2r - u2i; out[k + 2L * m].im = m2i + u2r;
      out[k + 3L * m].re = m2r + u2i; out[k + 3L * m].im = m2i - u2r;
    }
  } else {
    /* generic radix p (a prime above 5): p-point DFT of the twiddled column, O(p^2) */
    v8c tt[64], uu[64];
    const long rs = pl->n / p;
    for (int k = 0; k < m; k++) {
      tt[0] = out[k];
      for (int r = 1; r < p; r++) {
        const f32_cpx w = lt[(long)k * (p - 1) + (r - 1)]; const v8c v = out[k + (long)r * m];
        V8_CMUL(tt[r].re, tt[r].im, v.re, v.im, w.re, -s * w.im);
      }
      for (int q = 0; q < p; q++) {
        v8f ar = tt[0].re * 0.f, ai = ar;
        for (int r = 0; r < p; r++) {
          const long ti = ((long)q * r % p) * rs;
          const float wr = pl->tw[ti].re, wi = -s * pl->tw[ti].im;
          ar += tt[r].re * wr - tt[r].im * wi; ai += tt[r].re * wi + tt[r].im * wr;
        }
        uu[q].re = ar; uu[q].im = ai;
      }
      for (int q = 0; q < p; q++) out[k + (long)q * m] = uu[q];
    }
  }
}
/* can this plan run batched?  (every level has its table; generic radices fit the scratch arrays) */
static int v8_usable(const f32_plan *pl) {
  if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("fma")) return 0;
  for (int lv = 0; lv < pl->nfac; lv++) if (!pl->ltw[lv] || pl->fac[lv] > 64) return 0;
  return 1;
}
