// regfft.h -- in-register mixed-radix DFTs for gfx950 wavefront lanes.
//
// Every lane of a wavefront owns R complex values in VGPRs and transforms them
// without touching memory.  R is a compile-time constant with prime factors in
// {2,3,5}; the recursion below is resolved entirely at compile time (every array
// index and every twiddle is a constant expression), so a RegDFT<R> is a straight
// line of v_add/v_fma/v_pk_* instructions with twiddles as literal operands.
//
// These are the "radix butterflies" of the channelizer kernels in chz_kernels.h:
// the forward transform the reference runs at src/filter.c:505-508 and the
// per-channel backward transform at src/filter.c:914 are both built out of them.
//
// SIGN = -1: X[k] = sum_n x[n] e^{-2 pi i n k / R}   (forward, unnormalised)
// SIGN = +1: x[n] = sum_k X[k] e^{+2 pi i n k / R}   (backward, unnormalised)
#pragma once
#include <utility>
#include <type_traits>

#define CHZ_DEV __device__ __forceinline__

namespace chz {

// ---- compile-time loop ------------------------------------------------------
template <int I> using ic = std::integral_constant<int, I>;
template <class F, int... Is>
CHZ_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ic<Is>{}), ...); }
template <int N, class F>
CHZ_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- compile-time roots of unity ---------------------------------------------
struct cis_t { double c, s; };

constexpr double cx_sin_small(double x) {   // |x| <= pi/4, Taylor to x^21 (err < 1e-19)
  double x2 = x * x, term = x, sum = x;
  for (int k = 1; k <= 10; k++) { term *= -x2 / ((2.0 * k) * (2.0 * k + 1.0)); sum += term; }
  return sum;
}
constexpr double cx_cos_small(double x) {   // |x| <= pi/4
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int k = 1; k <= 10; k++) { term *= -x2 / ((2.0 * k - 1.0) * (2.0 * k)); sum += term; }
  return sum;
}
// e^{+2 pi i num/den}, evaluated with the angle folded into one octant
constexpr cis_t cx_cis(long num, long den) {
  num %= den; if (num < 0) num += den;
  long k8 = 8 * num, oct = k8 / den, rem = k8 - oct * den;
  double a = 0.78539816339744830961566 * (static_cast<double>(rem) / static_cast<double>(den));
  double c = 0, s = 0;
  if (oct & 1) { a = 0.78539816339744830961566 - a; s = cx_cos_small(a); c = cx_sin_small(a); }
  else         { c = cx_cos_small(a); s = cx_sin_small(a); }
  switch (oct >> 1) {
    case 0: return cis_t{ c,  s};
    case 1: return cis_t{-s,  c};
    case 2: return cis_t{-c, -s};
    default: return cis_t{ s, -c};
  }
}
template <int NUM, int DEN> struct Root { static constexpr cis_t v = cx_cis(NUM, DEN); };

// ---- complex helpers on float2 -------------------------------------------------
CHZ_DEV float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
CHZ_DEV float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
CHZ_DEV float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
CHZ_DEV float2 cmul_conj(float2 a, float2 b) {   // a * conj(b)
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
CHZ_DEV float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
CHZ_DEV float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by (SIGN * i):  i*(x+iy) = -y+ix
template <int SIGN> CHZ_DEV float2 mul_si(float2 a) {
  return SIGN > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// multiply by e^{SIGN * 2 pi i NUM/DEN} with the trivial cases folded away
template <int NUM, int DEN, int SIGN> CHZ_DEV float2 twiddle_const(float2 a) {
  constexpr int n = ((NUM % DEN) + DEN) % DEN;
  if constexpr (n == 0) return a;
  else if constexpr (2 * n == DEN) return make_float2(-a.x, -a.y);
  else if constexpr (4 * n == DEN) return mul_si<SIGN>(a);
  else if constexpr (4 * n == 3 * DEN) return mul_si<-SIGN>(a);
  else {
    constexpr float c = static_cast<float>(Root<n, DEN>::v.c);
    constexpr float s = static_cast<float>(SIGN * Root<n, DEN>::v.s);
    return make_float2(fmaf(a.x, c, -a.y * s), fmaf(a.x, s, a.y * c));
  }
}

// ---- base butterflies -----------------------------------------------------------
template <int SIGN> CHZ_DEV void bfly2(float2& a, float2& b) {
  float2 t = csub(a, b); a = cadd(a, b); b = t;
}
template <int SIGN> CHZ_DEV void bfly3(float2& x0, float2& x1, float2& x2) {
  constexpr float s60 = static_cast<float>(0.86602540378443864676);
  float2 t = cadd(x1, x2);
  float2 m = make_float2(fmaf(-0.5f, t.x, x0.x), fmaf(-0.5f, t.y, x0.y));
  float2 d = cscale(csub(x1, x2), s60);
  float2 r = mul_si<SIGN>(d);               // SIGN*i*d
  x0 = cadd(x0, t);
  x1 = cadd(m, r);                          // forward: m - i d
  x2 = csub(m, r);
}
template <int SIGN> CHZ_DEV void bfly4(float2& x0, float2& x1, float2& x2, float2& x3) {
  float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = mul_si<SIGN>(csub(x1, x3));
  x0 = cadd(a, c); x2 = csub(a, c);
  x1 = cadd(b, d);                          // forward: b - i (x1 - x3)
  x3 = csub(b, d);
}
template <int SIGN> CHZ_DEV void bfly5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
  constexpr float c1 = static_cast<float>(Root<1, 5>::v.c), c2 = static_cast<float>(Root<2, 5>::v.c);
  constexpr float s1 = static_cast<float>(Root<1, 5>::v.s), s2 = static_cast<float>(Root<2, 5>::v.s);
  float2 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
  float2 m1 = make_float2(fmaf(c1, t1.x, fmaf(c2, t2.x, x0.x)), fmaf(c1, t1.y, fmaf(c2, t2.y, x0.y)));
  float2 m2 = make_float2(fmaf(c2, t1.x, fmaf(c1, t2.x, x0.x)), fmaf(c2, t1.y, fmaf(c1, t2.y, x0.y)));
  float2 u1 = make_float2(fmaf(s1, t3.x, s2 * t4.x), fmaf(s1, t3.y, s2 * t4.y));
  float2 u2 = make_float2(fmaf(s2, t3.x, -s1 * t4.x), fmaf(s2, t3.y, -s1 * t4.y));
  float2 r1 = mul_si<SIGN>(u1), r2 = mul_si<SIGN>(u2);
  x0 = cadd(x0, cadd(t1, t2));
  x1 = cadd(m1, r1); x4 = csub(m1, r1);     // forward: m1 -/+ i u1
  x2 = cadd(m2, r2); x3 = csub(m2, r2);
}

// Odd prime lengths above 5 (7, 11, 13, 19 ...: the Airspy HF+'s N = 36,480 = 2^7*3*5*19, docs/FFTW3.md:52) by
// definition, folded once: X[k] = x0 + sum_{n=1}^{(P-1)/2} [ (x_n + x_{P-n}) cos(2 pi n k / P) + SIGN i (x_n - x_{P-n}) sin(2 pi n k / P) ],
// with X[P-k] sharing both sums.  (P-1)^2/2 real multiply-adds per component, all coefficients literals.
template <int P, int SIGN, int ST> CHZ_DEV void bfly_prime(float2* v) {
  constexpr int H = (P - 1) / 2;
  float2 sp[H], sm[H];
  static_for<H>([&](auto n) {
    constexpr int N = decltype(n)::value + 1;
    sp[N - 1] = cadd(v[N * ST], v[(P - N) * ST]);
    sm[N - 1] = csub(v[N * ST], v[(P - N) * ST]);
  });
  const float2 x0 = v[0];
  float2 acc0 = x0;
  static_for<H>([&](auto n) { constexpr int N = decltype(n)::value; acc0 = cadd(acc0, sp[N]); });
  v[0] = acc0;
  static_for<H>([&](auto k) {
    constexpr int K = decltype(k)::value + 1;
    float2 a = x0, b = make_float2(0.f, 0.f);
    static_for<H>([&](auto n) {
      constexpr int N = decltype(n)::value + 1;
      constexpr float c = static_cast<float>(Root<(N * K) % P, P>::v.c);
      constexpr float sn = static_cast<float>(Root<(N * K) % P, P>::v.s);
      a = make_float2(fmaf(c, sp[N - 1].x, a.x), fmaf(c, sp[N - 1].y, a.y));
      b = make_float2(fmaf(sn, sm[N - 1].x, b.x), fmaf(sn, sm[N - 1].y, b.y));
    });
    const float2 r = mul_si<SIGN>(b);         // SIGN * i * b
    v[K * ST] = cadd(a, r);
    v[(P - K) * ST] = csub(a, r);
  });
}

constexpr bool is_small_prime(int r) { return r == 7 || r == 11 || r == 13 || r == 17 || r == 19; }
// smallest supported base radix of R (4 preferred over 2); an odd prime factor above 5 is taken whole
constexpr int base_radix(int r) {
  if (r % 4 == 0) return 4;
  if (r % 2 == 0) return 2;
  if (r % 3 == 0) return 3;
  if (r % 5 == 0) return 5;
  for (int p : {7, 11, 13, 17, 19}) if (r % p == 0) return p;
  return 0;
}

// ---- recursive Cooley-Tukey, natural order in and out ----------------------------
// v points at element 0; logical element i lives at v[i*ST].
template <int R, int SIGN> struct RegDFT {
  static constexpr int A = base_radix(R);
  static_assert(R == 1 || A != 0, "RegDFT: prime factors 2, 3, 5 and one of 7..19 only");
  static constexpr int B = (R == 1) ? 1 : R / (A == 0 ? 1 : A);

  template <int ST = 1> static CHZ_DEV void run(float2* v) {
    if constexpr (R == 1) { (void)v; }
    else if constexpr (R == 2) { bfly2<SIGN>(v[0], v[ST]); }
    else if constexpr (R == 3) { bfly3<SIGN>(v[0], v[ST], v[2 * ST]); }
    else if constexpr (R == 4) { bfly4<SIGN>(v[0], v[ST], v[2 * ST], v[3 * ST]); }
    else if constexpr (R == 5) { bfly5<SIGN>(v[0], v[ST], v[2 * ST], v[3 * ST], v[4 * ST]); }
    else if constexpr (is_small_prime(R)) { bfly_prime<R, SIGN, ST>(v); }
    else {
      // x[n], n = A*m + n1.  (1) B-point DFT over m for each residue n1.
      static_for<A>([&](auto n1) {
        constexpr int N1 = decltype(n1)::value;
        RegDFT<B, SIGN>::template run<ST * A>(v + N1 * ST);
      });
      // now v[(A*k2 + n1)*ST] = U_{n1}[k2].  (2) twiddle and A-point DFT over n1.
      float2 tmp[R];
      static_for<B>([&](auto k2) {
        constexpr int K2 = decltype(k2)::value;
        float2 t[A];
        static_for<A>([&](auto n1) {
          constexpr int N1 = decltype(n1)::value;
          t[N1] = twiddle_const<N1 * K2, R, SIGN>(v[(A * K2 + N1) * ST]);
        });
        RegDFT<A, SIGN>::template run<1>(t);
        static_for<A>([&](auto k1) {
          constexpr int K1 = decltype(k1)::value;
          tmp[K2 + B * K1] = t[K1];
        });
      });
      static_for<R>([&](auto i) {
        constexpr int I = decltype(i)::value;
        v[I * ST] = tmp[I];
      });
    }
  }
};

template <int R, int SIGN> CHZ_DEV void reg_dft(float2 (&v)[R]) { RegDFT<R, SIGN>::template run<1>(v); }

}  // namespace chz
