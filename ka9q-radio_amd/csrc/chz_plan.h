// chz_plan.h -- host-side planning for the forward transform and channel banks:
// choice of the axis lengths N = Na*Nb*Nc, tile widths, LDS geometry, and the
// float64-accurate twiddle tables the kernels in chz_kernels.h consume.
// Pure host C++ (no HIP runtime calls) so chz_engine.hip and the CPU test
// harness build the very same plan.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <complex>

namespace chz {

// (R1,R2) pairs with a compiled kernel instantiation.  Sub-transform length = R1*R2.
#define CHZ_FWD_MENU(X) \
  X(4, 4) X(5, 5) X(6, 6) X(5, 9) X(5, 10) X(5, 15) X(8, 8) X(8, 9) X(9, 9) X(10, 10) X(10, 12) X(5, 25) X(9, 15) \
  X(12, 12) X(10, 15) X(8, 19) X(10, 16) X(12, 15) X(12, 16) X(15, 15) X(15, 16) X(16, 16) X(16, 20) X(20, 20) X(9, 16) \
  X(10, 20) X(10, 13)
// per-channel backward transform lengths P = R1*R2 (reference sizes: docs/FFTW3.md:51-68)
#define CHZ_CHAN_MENU(X) \
  X(4, 5) X(5, 6) X(10, 15) X(10, 16) X(10, 20) X(15, 20) X(16, 20) X(20, 20) X(20, 24) \
  X(24, 25) X(25, 32) X(30, 32) X(30, 40) X(40, 48)

struct Radix2 { int r1, r2; };

inline bool fwd_menu_lookup(int np, Radix2* out) {
#define X(a, b) if ((a) * (b) == np) { if (out) { out->r1 = a; out->r2 = b; } return true; }
  CHZ_FWD_MENU(X)
#undef X
  return false;
}
// Axis c (fwd_rows) loads whole 128-byte lines straight into its first butterfly layer when R2 is a multiple of 16:
// prefer such a factorisation of the axis length where the menu has one.
inline bool fwd_menu_lookup_c(int np, Radix2* out) {
#define X(a, b) if ((a) * (b) == np && (b) % 16 == 0) { if (out) { out->r1 = a; out->r2 = b; } return true; }
  CHZ_FWD_MENU(X)
#undef X
  return fwd_menu_lookup(np, out);
}
inline bool chan_menu_lookup(int p, Radix2* out) {
#define X(a, b) if ((a) * (b) == p) { if (out) { out->r1 = a; out->r2 = b; } return true; }
  CHZ_CHAN_MENU(X)
#undef X
  return false;
}

struct f2 { float x, y; };   // layout-identical to HIP's float2

// e^{sign * 2 pi i num/den} rounded once from float64, angle folded into an octant
inline f2 root_of_unity(long long num, long long den, int sign, double scale_re = 1.0, double scale_im = 0.0) {
  num %= den; if (num < 0) num += den;
  long long k8 = 8 * num, oct = k8 / den, rem = k8 - oct * den;
  double a = (M_PI / 4.0) * ((double)rem / (double)den), c, s;
  if (oct & 1) { a = M_PI / 4.0 - a; s = std::cos(a); c = std::sin(a); }
  else { c = std::cos(a); s = std::sin(a); }
  double cc, ss;
  switch (oct >> 1) {
    case 0: cc = c; ss = s; break;
    case 1: cc = -s; ss = c; break;
    case 2: cc = -c; ss = -s; break;
    default: cc = s; ss = -c; break;
  }
  ss *= sign;
  // multiply by the optional complex scale (used to fold 1/2 and -i/2 into a table)
  double re = cc * scale_re - ss * scale_im, im = cc * scale_im + ss * scale_re;
  return f2{(float)re, (float)im};
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// W_NP^(sign*j*k1) laid out [R2][R1] as the first butterfly layer indexes it
inline std::vector<f2> make_tw_sub(int r1, int r2, int sign) {
  std::vector<f2> t((size_t)r1 * r2);
  for (int j = 0; j < r2; j++)
    for (int k1 = 0; k1 < r1; k1++) t[(size_t)j * r1 + k1] = root_of_unity((long long)j * k1, (long long)r1 * r2, sign);
  return t;
}

enum { CHZ_IN_COMPLEX = 1, CHZ_IN_REAL = 2 };   // enum filtertype values, src/filter.h:29-34

struct FwdPlan {
  int N = 0, in_type = 0, bins = 0;
  int Na = 0, Nb = 1, Nc = 0;         // Nb == 1: two-axis plan (no fwd_cols for axis b)
  Radix2 ra{}, rb{}, rc{};
  int Ra = 0;                          // rows of the intermediate buffer (Na/2+1 real, Na complex)
  int inner = 0;                       // Nb*Nc
  int T1 = 0, T2 = 0, Ta = 0;          // tile widths: first axis (packed cols if real), axis b, last axis
  int padk1 = 0, padk2 = 0, ld3 = 0, padg3 = 0;
  int spec_pitch = 0, spec_off = 0, ka_shift = 0;   // spectrum storage (see SpecLayout in chz_kernels.h)
  long spec_elems = 0;                              // float2 elements of one spectrum slot
  int grid1 = 0, block1 = 0, grid2 = 0, block2 = 0, grid3 = 0, block3 = 0;
  size_t lds1 = 0, lds2 = 0, lds3 = 0;
  std::vector<f2> tw_sub_a, tw_sub_b, tw_sub_c, tw1_tile, tw1_col, tw2_tile, tw2_col, tw2_full;
  std::string desc;
};

inline int padk_for(int r2, int T) { int v = (T - r2 * T) % 32; if (v < 0) v += 32; return v; }

// choose a divisor of `n` as tile width: closest to `want`, within [lo,hi], limited by thread count
inline int pick_tile(int n, int want, int lo, int hi, int lanes_per_col, int max_threads) {
  int best = 0; double bestscore = 1e30;
  for (int t = 1; t <= n && t <= hi; t++) {
    if (n % t) continue;
    if (t * lanes_per_col > max_threads) continue;
    // 16 float2 = one 128-byte line: aligned whole-line tiles are worth a lot
    double score = std::fabs(std::log((double)t / want)) + (t < lo ? 1.0 : 0.0) - (t % 16 == 0 ? 0.75 : 0.0);
    if (score < bestscore) { bestscore = score; best = t; }
  }
  return best;
}

inline bool finish_fwd_plan(FwdPlan& p, int T1_over, int T2_over, int Ta_over) {
  const bool real = p.in_type == CHZ_IN_REAL;
  p.inner = p.Nb * p.Nc;
  p.Ra = real ? p.Na / 2 + 1 : p.Na;
  p.bins = real ? p.N / 2 + 1 : p.N;
  const int la = p.ra.r1 > p.ra.r2 ? p.ra.r1 : p.ra.r2;
  // ---- first axis
  if (real) {
    if (p.inner & 1) return false;
    const int cols_p = p.inner / 2;
    p.T1 = T1_over > 0 ? T1_over : pick_tile(cols_p, 16, 8, 64, la, 1024);
    if (p.T1 <= 0 || cols_p % p.T1) return false;
    p.grid1 = cols_p / p.T1;
    p.tw1_tile.resize((size_t)p.grid1 * p.Ra);
    for (int tile = 0; tile < p.grid1; tile++)
      for (int k = 0; k < p.Ra; k++)
        p.tw1_tile[(size_t)tile * p.Ra + k] = root_of_unity((long long)k * 2 * tile * p.T1, p.N, -1);
    p.tw1_col.resize((size_t)p.Ra * 2 * p.T1);
    for (int k = 0; k < p.Ra; k++)
      for (int cc = 0; cc < 2 * p.T1; cc++)   // 1/2 for even columns, 1/(2i) = -i/2 for odd ones
        p.tw1_col[(size_t)k * 2 * p.T1 + cc] =
            (cc & 1) ? root_of_unity((long long)k * cc, p.N, -1, 0.0, -0.5) : root_of_unity((long long)k * cc, p.N, -1, 0.5, 0.0);
  } else {
    p.T1 = T1_over > 0 ? T1_over : pick_tile(p.inner, 16, 8, 64, la, 1024);
    if (p.T1 <= 0 || p.inner % p.T1) return false;
    p.grid1 = p.inner / p.T1;
    p.tw1_tile.resize((size_t)p.grid1 * p.Na);
    for (int tile = 0; tile < p.grid1; tile++)
      for (int k = 0; k < p.Na; k++)
        p.tw1_tile[(size_t)tile * p.Na + k] = root_of_unity((long long)k * tile * p.T1, p.N, -1);
    p.tw1_col.resize((size_t)p.Na * p.T1);
    for (int k = 0; k < p.Na; k++)
      for (int t = 0; t < p.T1; t++) p.tw1_col[(size_t)k * p.T1 + t] = root_of_unity((long long)k * t, p.N, -1);
  }
  p.padk1 = padk_for(p.ra.r2, p.T1);
  p.block1 = round_up(la * p.T1, 64);
  p.lds1 = sizeof(f2) * ((size_t)p.Na * p.T1 + (size_t)p.ra.r1 * p.padk1);
  if (real) p.lds1 += sizeof(f2) * (size_t)p.Na * p.T1;      // second region for the Hermitian split
  p.tw_sub_a = make_tw_sub(p.ra.r1, p.ra.r2, -1);
  // ---- axis b
  if (p.Nb > 1) {
    const int lb = p.rb.r1 > p.rb.r2 ? p.rb.r1 : p.rb.r2;
    p.T2 = T2_over > 0 ? T2_over : pick_tile(p.Nc, 16, 8, 64, lb, 1024);
    if (p.T2 <= 0 || p.Nc % p.T2) return false;
    const int tpr = p.Nc / p.T2;
    p.grid2 = p.Ra * tpr;
    p.block2 = round_up(lb * p.T2, 64);
    p.padk2 = padk_for(p.rb.r2, p.T2);
    p.lds2 = sizeof(f2) * ((size_t)p.Nb * p.T2 + (size_t)p.rb.r1 * p.padk2);
    const long long D = (long long)p.Nb * p.Nc;
    p.tw2_tile.resize((size_t)tpr * p.Nb);
    for (int ct = 0; ct < tpr; ct++)
      for (int k = 0; k < p.Nb; k++) p.tw2_tile[(size_t)ct * p.Nb + k] = root_of_unity((long long)k * ct * p.T2, D, -1);
    p.tw2_col.resize((size_t)p.Nb * p.T2);
    for (int k = 0; k < p.Nb; k++)
      for (int t = 0; t < p.T2; t++) p.tw2_col[(size_t)k * p.T2 + t] = root_of_unity((long long)k * t, D, -1);
    p.tw2_full.resize((size_t)p.Nb * p.Nc);                 // W_D^(k*col): tile and column factor in one, rounded once
    for (int k = 0; k < p.Nb; k++)
      for (int c = 0; c < p.Nc; c++) p.tw2_full[(size_t)k * p.Nc + c] = root_of_unity((long long)k * c, D, -1);
    p.tw_sub_b = make_tw_sub(p.rb.r1, p.rb.r2, -1);
  }
  // ---- last axis
  const int lc = p.rc.r1 > p.rc.r2 ? p.rc.r1 : p.rc.r2;
  p.Ta = Ta_over > 0 ? Ta_over : 16;
  if (p.Ta > p.Ra) p.Ta = p.Ra;
  while (lc * p.Ta > 1024) p.Ta--;
  // spectrum storage: natural order unless a real master with odd Na lets both the direct and the
  // conjugate-mirrored 16-bin store segments start on 128-byte lines:
  //   direct segment of tile [a0, a0+16) starts at off + a0, mirrored one at off + Na - a0 - 15
  //   => off = s with 2s = 15 - Na (mod 16), tiles start at a0 = -s (mod 16)
  p.spec_pitch = p.Na; p.spec_off = 0; p.ka_shift = 0;
  if (real && (p.Na & 1) && p.Ta == 16 && p.Nc % 16 == 0) {
    int s2 = ((15 - p.Na) % 16 + 16) % 16;          // even because Na is odd
    p.spec_off = s2 / 2;
    p.ka_shift = p.spec_off;
    p.spec_pitch = round_up(p.spec_off + p.Na, 16);
  }
  {
    const long xrows = real ? ((long)p.N / 2) / p.Na + 1 : (long)p.N / p.Na;
    p.spec_elems = xrows * p.spec_pitch + 16;
  }
  p.ld3 = p.Ta + 1 + (p.Ta & 1);                 // odd leading dimension
  if (!(p.ld3 & 1)) p.ld3++;
  p.padg3 = (16 - (p.rc.r2 * p.ld3) % 32 + 32) % 32;
  p.grid3 = p.Nb * ((p.Ra + p.ka_shift + p.Ta - 1) / p.Ta);
  p.block3 = round_up(lc * p.Ta, 64);
  p.lds3 = sizeof(f2) * ((size_t)p.Nc * p.ld3 + (size_t)p.rc.r1 * p.padg3 + 8);
  p.tw_sub_c = make_tw_sub(p.rc.r1, p.rc.r2, -1);
  char b[256];
  snprintf(b, sizeof b, "N=%d %s axes %dx%dx%d radices (%d,%d)(%d,%d)(%d,%d) tiles T1=%d T2=%d Ta=%d grids %d/%d/%d blocks %d/%d/%d spec pitch %d off %d",
           p.N, real ? "real" : "complex", p.Na, p.Nb, p.Nc, p.ra.r1, p.ra.r2, p.rb.r1, p.rb.r2, p.rc.r1, p.rc.r2,
           p.T1, p.T2, p.Ta, p.grid1, p.grid2, p.grid3, p.block1, p.block2, p.block3, p.spec_pitch, p.spec_off);
  p.desc = b;
  return true;
}

// spec: "" (automatic) or "NaxNbxNc[:T1,T2,Ta]" / "NaxNc[:T1,Ta]"
inline bool build_fwd_plan(int N, int in_type, const char* spec, FwdPlan& out, double* score_out = nullptr) {
  if (N < 4 || (in_type != CHZ_IN_REAL && in_type != CHZ_IN_COMPLEX)) return false;
  const bool real = in_type == CHZ_IN_REAL;
  int T1o = 0, T2o = 0, Tao = 0;
  if (spec && *spec) {
    int a = 0, b = 0, c = 0;
    int n = sscanf(spec, "%dx%dx%d", &a, &b, &c);
    if (n == 2) { c = b; b = 1; }
    else if (n != 3) return false;
    const char* colon = strchr(spec, ':');
    if (colon) {
      int t[3] = {0, 0, 0};
      int m = sscanf(colon + 1, "%d,%d,%d", &t[0], &t[1], &t[2]);
      if (b == 1) { T1o = t[0]; Tao = m >= 2 ? t[1] : 0; }
      else { T1o = t[0]; T2o = m >= 2 ? t[1] : 0; Tao = m >= 3 ? t[2] : 0; }
    }
    if ((long long)a * b * c != N) return false;
    FwdPlan p; p.N = N; p.in_type = in_type; p.Na = a; p.Nb = b; p.Nc = c;
    if (!fwd_menu_lookup(a, &p.ra) || !fwd_menu_lookup_c(c, &p.rc)) return false;
    if (b > 1 && !fwd_menu_lookup(b, &p.rb)) return false;
    if (!finish_fwd_plan(p, T1o, T2o, Tao)) return false;
    out = p;
    return true;
  }
  // automatic: enumerate menu triples whose product is N, prefer balanced axes with enough tiles
  static const int menu[] = {
#define X(a, b) (a) * (b),
      CHZ_FWD_MENU(X)
#undef X
  };
  const int nm = (int)(sizeof menu / sizeof menu[0]);
  double best = 1e300; FwdPlan bestp; bool found = false;
  for (int ia = 0; ia < nm; ia++) {
    const int a = menu[ia];
    if (N % a) continue;
    for (int ib = -1; ib < nm; ib++) {
      const int b = ib < 0 ? 1 : menu[ib];
      if ((N / a) % b) continue;
      const int c = N / a / b;
      if (!fwd_menu_lookup(c, nullptr)) continue;
      if (real && ((long long)b * c) % 2) continue;
      FwdPlan p; p.N = N; p.in_type = in_type; p.Na = a; p.Nb = b; p.Nc = c;
      fwd_menu_lookup(a, &p.ra); fwd_menu_lookup_c(c, &p.rc);
      if (b > 1) fwd_menu_lookup(b, &p.rb);
      if (!finish_fwd_plan(p, 0, 0, 0)) continue;
      // cost model (fitted to scripts/plan_sweep.py runs on MI355X): every pass moves the whole
      // data set once; 128-byte aligned 16-column tiles run at copy speed and misaligned ones at
      // about half of it; a pass wants >= ~600 workgroups; long or lopsided butterflies cost VALU time
      auto grid_pen = [](int g) { return g < 600 ? (600.0 - g) / 600.0 : 0.0; };
      auto shape_pen = [](Radix2 r) {
        const int hi = r.r1 > r.r2 ? r.r1 : r.r2, lo = r.r1 > r.r2 ? r.r2 : r.r1;
        return 0.08 * ((double)hi / lo - 1.0) + (hi * lo > 256 ? 0.3 : 0.0) + (hi > 16 ? 0.15 : 0.0);
      };
      const bool al1 = p.T1 % 16 == 0;
      const bool al2 = (b == 1) || (p.T2 % 16 == 0 && c % 16 == 0);
      const bool al3 = real ? (p.spec_off != 0 || (a % 16 == 0)) : (a % 16 == 0);
      double score = (b > 1 ? 3.0 : 2.0) + grid_pen(p.grid1) + grid_pen(p.grid3) + (b > 1 ? grid_pen(p.grid2) : 0.0) +
                     shape_pen(p.ra) + shape_pen(p.rc) + (b > 1 ? shape_pen(p.rb) : 0.0) +
                     (al1 ? 0.0 : 0.5) + (al2 ? 0.0 : 0.5) + (al3 ? 0.0 : 0.25) -
                     (p.rc.r2 % 16 == 0 ? 0.15 : 0.0);      // direct-load first layer of fwd_rows: that pass runs 15 % faster
      if (score < best) { best = score; bestp = p; found = true; }
    }
  }
  if (!found) return false;
  out = bestp;
  if (score_out) *score_out = best;
  return true;
}

// ---- channel banks -------------------------------------------------------------
#ifndef CHZ_MINI_MAX_STAGES
#define CHZ_MINI_MAX_STAGES 14
#endif
inline bool mini_factor(int N, int* radix, int* nstages);
#define CHZ_ANY_LDS_P 10240                     // two P-point complex buffers in the 160 KB of LDS
#define CHZ_ANY_MAX_P (1 << 20)                 // beyond CHZ_ANY_LDS_P the two buffers are global scratch (the L2 keeps them up to ~100 k points; beyond that they stream)
struct ChanGeom {
  int P = 0; Radix2 r{}; int lpc = 0, cpw = 0, wpb = 1;
  size_t lds = 0;
  std::vector<f2> tw_sub;   // backward sign
  // P without a register-tiled kernel in the menu (wfm's 384 kHz channel: P = 9600; odd sample rates): one workgroup per
  // channel, Stockham stages in LDS (chan_any)
  bool any = false, big = false;
  int any_threads = 0, nstages = 0, radix[CHZ_MINI_MAX_STAGES] = {0};
  std::vector<f2> tw_any;   // [P] e^{-2 pi i k / P}; Bluestein: [M] e^{-2 pi i k / M}, then the chirp [P], then F(b) [M]
  // P WITH a prime factor above 13 (FFTW plans any size, src/filter.c:101-163; so does this): Bluestein's chirp-z identity turns
  // the P-point backward transform into a circular convolution of M >= 2P - 1 points, M a power of two, run by the same Stockham
  // stages -- two M-point transforms and two chirp multiplications per channel and block (chan_any).
  int blue_M = 0;
  int lb = 0;               // points per transform buffer: P, or M for Bluestein
};
// host-side double-precision radix-2 transform for Bluestein's F(b) (M a power of two, forward sign)
inline void host_fft_pow2(std::vector<std::complex<double>>& x) {
  const size_t n = x.size();
  for (size_t i = 1, j = 0; i < n; i++) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(x[i], x[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; k++) {
        const double ang = -2.0 * M_PI * (double)k / (double)len;
        const std::complex<double> w(std::cos(ang), std::sin(ang));
        const std::complex<double> u = x[i + k], v = x[i + k + len / 2] * w;
        x[i + k] = u + v; x[i + k + len / 2] = u - v;
      }
  }
}
inline bool build_chan_geom(int P, ChanGeom& g) {
  if (!chan_menu_lookup(P, &g.r)) {
    if (P < 8 || P > CHZ_ANY_MAX_P) return false;
    g.P = P; g.any = true;
    if (!mini_factor(P, g.radix, &g.nstages)) {                 // a prime factor above 13: Bluestein over M = 2^k >= 2P - 1
      int M = 16;
      while (M < 2 * P - 1) M <<= 1;
      if (M > CHZ_ANY_MAX_P || !mini_factor(M, g.radix, &g.nstages)) return false;      // (chirp-z needs M >= 2P - 1 points per buffer)
      g.blue_M = M;
      g.tw_any.resize((size_t)M + (size_t)P + (size_t)M);
      for (int k = 0; k < M; k++) g.tw_any[(size_t)k] = root_of_unity(k, M, -1);
      // chirp w_n = e^{+i pi n^2 / P}, the angle reduced exactly: n^2 mod 2P over 2P
      std::vector<std::complex<double>> b((size_t)M, std::complex<double>(0.0, 0.0));
      for (long long n = 0; n < P; n++) {
        const long long q = (n * n) % (2LL * P);
        g.tw_any[(size_t)M + (size_t)n] = root_of_unity(q, 2LL * P, +1);
        const double ang = M_PI * (double)q / (double)P;
        const std::complex<double> cw(std::cos(ang), -std::sin(ang));       // conj(w_n)
        b[(size_t)n] = cw;
        if (n > 0) b[(size_t)(M - n)] = cw;
      }
      host_fft_pow2(b);
      for (int k = 0; k < M; k++) g.tw_any[(size_t)M + (size_t)P + (size_t)k] = f2{(float)b[(size_t)k].real(), (float)b[(size_t)k].imag()};
    } else {
      g.tw_any.resize((size_t)P);
      for (int k = 0; k < P; k++) g.tw_any[(size_t)k] = root_of_unity(k, P, -1);
    }
    g.lb = g.blue_M ? g.blue_M : P;
    g.any_threads = g.lb <= 1024 ? 128 : g.lb <= 2048 ? 256 : g.lb <= 4096 ? 512 : 1024;
    g.big = g.lb > CHZ_ANY_LDS_P;
    g.lds = g.big ? 0 : sizeof(f2) * 2 * (size_t)g.lb;
    return true;
  }
  g.P = P;
  g.lpc = g.r.r1 > g.r.r2 ? g.r.r1 : g.r.r2;
  g.cpw = 64 / g.lpc;
  g.wpb = 1;                                   // one wavefront per workgroup: most workgroups, no barrier partners
#ifdef CHZ_EXPERIMENTS
  if (const char* w = getenv("CHZ_CHAN_WPB")) { int v = atoi(w); if (v == 2 || v == 4) g.wpb = v; }   // experiment knob
#endif
  g.lds = sizeof(f2) * (size_t)g.wpb * g.cpw * g.r.r1 * (g.r.r2 + 1);
  g.tw_sub = make_tw_sub(g.r.r1, g.r.r2, +1);
  return true;
}

// Host restatement of the gather index walk (src/filter.c:728-911, COMPLEX
// output) as a closed-form descriptor; mirrors struct ChanDesc in chz_kernels.h.
struct ChanDescH { int t0, cnt, src0, dir, conj, wrap; };
inline ChanDescH make_chan_desc(int in_type, int m_bins, int P, int shift) {
  ChanDescH d{0, 0, 0, 1, 0, 0};
  const long long r0 = (long long)shift - P / 2;      // master index of the most negative output bin
  if (in_type == CHZ_IN_REAL) {
    if (shift >= 0) {                                 // src/filter.c:819-855
      long long t0 = r0 < 0 ? -r0 : 0;
      long long src0 = r0 + t0;
      long long cnt = (long long)P - t0; if (cnt > m_bins - src0) cnt = m_bins - src0;
      if (t0 >= P || cnt <= 0) return d;
      d.t0 = (int)t0; d.src0 = (int)src0; d.cnt = (int)cnt; d.dir = 1;
    } else {                                          // src/filter.c:856-892: read downward, conjugated
      long long top = -r0;                            // master index feeding t = 0
      long long t0 = top > m_bins - 1 ? top - (m_bins - 1) : 0;
      long long src0 = top - t0;
      long long cnt = (long long)P - t0; if (cnt > src0 + 1) cnt = src0 + 1;
      if (t0 >= P || src0 < 0 || cnt <= 0) return d;
      d.t0 = (int)t0; d.src0 = (int)src0; d.cnt = (int)cnt; d.dir = -1; d.conj = 1;
    }
  } else {                                            // src/filter.c:728-793
    const long long hb = (m_bins + 1) / 2;
    long long t0 = r0 < -hb ? -hb - r0 : 0;
    if (t0 >= P) return d;
    long long r = r0 + t0;
    long long rp = r < 0 ? r + m_bins : r;
    if (rp < 0 || rp >= m_bins) return d;
    long long cnt = rp < hb ? hb - rp : (long long)m_bins - rp + hb;   // until the read index arrives at hb
    if (cnt > P - t0) cnt = P - t0;
    d.t0 = (int)t0; d.src0 = (int)rp; d.cnt = (int)cnt; d.dir = 1; d.wrap = m_bins;
  }
  return d;
}

// ---- small inline masters (mini_ovs): Stockham stages of an N-point transform, radix 4 first ---------------
inline bool mini_factor(int N, int* radix, int* nstages) {
  int n = N, k = 0;
  while (n % 4 == 0 && k < CHZ_MINI_MAX_STAGES) { radix[k++] = 4; n /= 4; }
  const int small[6] = {2, 3, 5, 7, 11, 13};
  for (int r : small)
    while (n % r == 0 && k < CHZ_MINI_MAX_STAGES) { radix[k++] = r; n /= r; }
  *nstages = k;
  return n == 1;
}

}  // namespace chz
